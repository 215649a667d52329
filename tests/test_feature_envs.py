"""Feature-vector envs HarvestFeatures / CleanupFeatures (BASELINE config 0): the CPU oracle and the
HIP engine against fixtures produced by the reference itself (tests/golden/feat_*.npz)."""
import numpy as np
import pytest

import golden_check as gc

FEAT = gc.fixtures("feat_")


@pytest.mark.parametrize("name", FEAT)
def test_oracle_feature_golden(name):
    from oracle.pyoracle import Oracle
    g = gc.load(name)
    kind, n, kw = gc.feat_kwargs(g)
    orc = Oracle(kind, 2, n, **kw)
    gc.replay_feat(g, orc, env=1)
    orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FEAT)
def test_engine_feature_golden(name):
    from contracts_amd.engine import BatchedEnv
    g = gc.load(name)
    kind, n, kw = gc.feat_kwargs(g)
    env = BatchedEnv(kind, 3, n, **kw)
    gc.replay_feat(g, env, env=2)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract,horizon", [("harvest_features", 2, "harvest_local", 1000),
                                                     ("harvest_features", 8, None, 37),
                                                     ("cleanup_features", 5, "cleanup", 61),
                                                     ("cleanup_features", 9, None, 1000),
                                                     ("harvest_features", 1, None, 1),     # an episode per step, one agent
                                                     ("cleanup_features", 2, "cleanup", 2)])
def test_engine_feature_rollout_vs_oracle(kind, n, contract, horizon):
    """random rollouts with auto-reset: every persistent and output field after every step"""
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    E, T = 96, 150
    kw = dict(contract=contract, horizon=horizon, auto_reset=True)
    env, orc = BatchedEnv(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    seeds = (np.arange(E) * 7919 + 11).astype(np.uint64)
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    rs = np.random.RandomState(5)
    na = env.num_actions + 1  # the fire / second beam action of the code path is legal input too
    p = np.ones(na) / na
    if kind == "cleanup_features":
        p = np.array([.1, .1, .12, .1, .05, .1, .1, .28, .05])
    fields = ["grid", "agents", "rng", "timestep", "theta", "base_reward", "reward", "done", "info", "features", "int_metrics",
              "f64_metrics", "final_int_metrics", "final_f64_metrics"]
    for t in range(T):
        a = rs.choice(na, size=(E, n), p=p).astype(np.uint8)
        env.step(a)
        orc.step(a)
        for f in fields:
            x, y = env.download(f, raw=True) if f == "grid" else env.download(f), getattr(orc, f)
            if f == "rng":
                x, y = x.reshape(E, 2, 628)[:, :, :625], y.reshape(E, 2, 628)[:, :, :625]
            ok = np.allclose(x, y, rtol=0, atol=1e-9, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
            assert ok, "field %s differs at step %d (envs %s)" % (f, t, np.nonzero((x != y).reshape(E, -1).any(axis=1))[0][:6])
    env.close()
    orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["feat_harvest_n2", "feat_cleanup_n5", "feat_harvest_n8_nocontract"])
def test_feature_adapter_trace(name):
    """the drop-in classes through their reference-shaped dict API, process-global `random` / `np.random` fidelity"""
    import hashlib
    import random
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.feature_envs import CleanupFeatures, HarvestFeatures
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    g = gc.load(name)
    kind, n, kw = gc.feat_kwargs(g)
    seed = int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    if kind == "harvest_features":
        env, con = HarvestFeatures(num_agents=n, horizon=kw["horizon"]), contract_list.HarvestFeaturemodLocalContract(n)
    else:
        env, con = CleanupFeatures(num_agents=n, horizon=kw["horizon"]), contract_list.CleanupContract(n)
    top = SeparateContractSubgameStage(env, con, n, False) if kw["contract"] else env
    keys = ["a%d" % i for i in range(n)]
    nfeat = g["feature_obs"].shape[2]

    def fps():
        st, ps = np.random.get_state(), random.getstate()[1]
        return ((int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)),
                (int(ps[624]), int(hashlib.sha256(np.array(ps[:624], np.uint32).tobytes()).hexdigest()[:8], 16)))

    assert fps() == (tuple(int(x) for x in g["ctor_mt_np"]), tuple(int(x) for x in g["ctor_mt_py"]))
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    steps = 160
    for ep in range(len(g["ep_start"])):
        o = top.reset()
        assert fps() == (tuple(int(x) for x in g["reset_mt_np"][ep]), tuple(int(x) for x in g["reset_mt_py"][ep]))
        for i, k in enumerate(keys):
            assert np.array_equal(np.asarray(o[k])[:nfeat], g["reset_obs"][ep][i])
            if kw["contract"]:
                assert np.array_equal(np.asarray(o[k])[nfeat:], [g["theta"][ep], 0.0])
        truncated = False
        for t in range(ep_start[ep], min(ep_start[ep + 1], ep_start[ep] + steps)):
            acts = {k: int(g["actions"][t][i]) for i, k in enumerate(keys)}
            o, r, d, info = top.step(acts)
            assert set(d.keys()) == {"__all__", "a0", "a1"} and d["__all__"] == bool(g["done"][t])
            for i, k in enumerate(keys):
                assert np.array_equal(np.asarray(o[k])[:nfeat], g["feature_obs"][t][i]), (t, k)
                np.testing.assert_allclose(r[k], g["rew"][t][i], rtol=0, atol=1e-9)
                if kind == "harvest_features":
                    assert info[k]["eaten_apples"] == g["info0"][t][i] and info[k]["eaten_close_apples"] == g["info1"][t][i]
                else:
                    assert info[k]["cleaned_squares"] == g["info1"][t][i]
            assert fps() == (tuple(int(x) for x in g["mt_np"][t]), tuple(int(x) for x in g["mt_py"][t])), t
            truncated = t + 1 < ep_start[ep + 1]
        if truncated:
            break
    env.close()
