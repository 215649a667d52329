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
                                                     ("cleanup_features", 9, None, 1000)])
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
