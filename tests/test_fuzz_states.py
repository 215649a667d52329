"""One reference step from crafted dense-cluster states (tests/golden/fuzz_*.npz, produced by the
reference itself): contested cells, swaps, chains, cycles, shared cells, beams through crowds.
All scenarios of a fixture run as ONE batch (env i = scenario i)."""
import hashlib

import numpy as np
import pytest

import golden_check as gc

FUZZ = gc.fixtures("fuzz_")


def run_fuzz(g, impl, get, put):
    kind, n = str(g["kind"]), int(g["n"])
    S = len(g["seed"])
    impl.seed(g["seed"].astype(np.uint64), replay_constructor=False)
    put("grid", g["in_grid"])
    agents = np.zeros((S, n, 4), np.uint8)
    agents[:, :, :3] = g["in_agents"]
    put("agents", agents)
    put("timestep", np.full((S,), 5, np.int32))
    if kind == "cleanup":
        put("waste_perm", g["in_waste_perm"])
    impl.step(g["actions"])
    assert np.array_equal(get("agents")[:, :, :3], g["out_agents"]), "agents: scenarios %s" % np.nonzero(
        (get("agents")[:, :, :3] != g["out_agents"]).any(axis=(1, 2)))[0][:10]
    assert np.array_equal(get("grid"), g["out_grid"]), "grid"
    assert np.array_equal(get("base_reward"), g["base_rew"]), "rewards"
    info = get("info")
    assert np.array_equal(info[:, :, 0], g["eaten"]), "eaten"
    assert np.array_equal(info[:, :, 1], g["second"]), "cleaned/eaten_close"
    assert np.array_equal(get("features").astype(np.float64), g["feature_obs"]), "feature_obs"
    assert np.array_equal(get("rng")[:, 624], g["mt_pos"]), "MT position"
    if kind == "cleanup":
        assert np.array_equal(get("waste_perm"), g["out_waste_perm"]), "waste perm"
    obs = get("obs")
    assert np.array_equal(obs[: len(g["obs"])], g["obs"]), "obs (stored)"
    for s in range(S):
        sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(obs[s]).tobytes()).digest(), np.uint8)
        assert np.array_equal(sha, g["obs_sha"][s]), "obs sha scenario %d" % s


@pytest.mark.parametrize("name", FUZZ)
def test_oracle_fuzz(name):
    from oracle.pyoracle import Oracle
    g = gc.load(name)
    orc = Oracle(str(g["kind"]), len(g["seed"]), int(g["n"]), firing=bool(int(g["firing"])))

    def put(field, arr):
        getattr(orc, field)[...] = arr
        orc.import_state()

    run_fuzz(g, orc, lambda f: getattr(orc, f), put)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FUZZ)
def test_engine_fuzz(name):
    from contracts_amd.engine import BatchedEnv
    g = gc.load(name)
    env = BatchedEnv(str(g["kind"]), len(g["seed"]), int(g["n"]), firing=bool(int(g["firing"])))
    run_fuzz(g, env, env.download, env.upload)
    env.check_faults()
    env.close()
