"""RLlib BaseEnv-shaped vector hook: poll / send_actions / try_reset over one engine handle,
checked against the CPU oracle stepping the same seeds, including done -> try_reset cycles."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract,horizon", [("cleanup", 4, "cleanup", 25), ("harvest", 3, None, 30),
                                                     ("harvest_features", 2, "harvest_local", 20)])
def test_base_env_protocol_matches_oracle(kind, n, contract, horizon):
    from contracts_amd.vector_env import BatchedBaseEnv
    from oracle.pyoracle import Oracle
    E, T = 6, 70
    venv = BatchedBaseEnv(kind, E, n, contract=contract, seed0=500, horizon=horizon)
    orc = Oracle(kind, E, n, contract=contract, horizon=horizon)
    orc.seed(seed0=500)
    orc.reset()
    keys = ["a%d" % i for i in range(n)]
    grid = kind in ("cleanup", "harvest")

    def check_obs(ob, e):
        for i, k in enumerate(keys):
            if grid:
                assert np.array_equal(ob[k]["image"], orc.obs[e][i] / 255)
                if contract:
                    assert np.array_equal(ob[k]["contract"], [orc.theta[e], 0.0])
            else:
                f = orc.features[e][i].astype(np.float64)
                assert np.array_equal(ob[k], np.concatenate((f, [orc.theta[e], 0.0])) if contract else f)

    obs, rew, dones, infos, _ = venv.poll()
    assert sorted(obs.keys()) == list(range(E))
    for e in range(E):
        check_obs(obs[e], e)
    rs = np.random.RandomState(9)
    na = venv.engine.num_actions
    resets = 0
    for t in range(T):
        a = rs.randint(na, size=(E, n))
        venv.send_actions({e: {k: int(a[e, i]) for i, k in enumerate(keys)} for e in range(E)})
        orc.step(a.astype(np.uint8))
        obs, rew, dones, infos, _ = venv.poll()
        for e in range(E):
            check_obs(obs[e], e)
            for i, k in enumerate(keys):
                np.testing.assert_allclose(rew[e][k], orc.reward[e][i] if contract else orc.base_reward[e][i], rtol=0, atol=1e-9)
            assert dones[e]["__all__"] == bool(orc.done[e])
            if kind != "cleanup_features":
                assert infos[e]["a0"]["eaten_apples"] == orc.info[e][0][0]
        done_ids = [e for e in range(E) if dones[e]["__all__"]]
        if done_ids:
            mask = np.zeros((E,), np.uint8)
            mask[done_ids] = 1
            orc.reset(mask)
            for e in done_ids:
                ob = venv.try_reset(e)
                check_obs(ob[e], e)
                resets += 1
    assert resets >= E  # every sub-env went through at least one done -> try_reset cycle
    venv.stop()
    orc.close()
