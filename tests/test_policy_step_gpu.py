"""ce_step_policy: the action selection fused into the step kernel's action load (include/contracts_engine.h CE_POLICY_*).
The engine takes the policy's device output — a byte per agent, or |A| float scores per agent — derives the action inside the
step launch, records it in `actions_taken`, and steps.  Checked against a host restatement of the selection rule + the oracle
stepped with those actions: every field, every step, both RNG streams, slices on their own launches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["grid", "agents", "obs", "base_reward", "reward", "done", "info", "features", "int_metrics", "f64_metrics", "timestep", "theta"]


def _pair(kind, E, n, **kw):
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    env, orc = BatchedEnv(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 31 + 5
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    return env, orc


def _same(env, orc, tag):
    for f in FIELDS:
        a, b = env.download(f), getattr(orc, f)
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, err_msg="%s %s" % (f, tag))
        else:
            assert np.array_equal(a, b), "%s %s" % (f, tag)


@pytest.mark.parametrize("kind,n,firing,rng", [("cleanup", 8, False, "mt19937"), ("cleanup", 4, True, "mt19937"), ("harvest", 8, False, "mt19937"),
                                               ("harvest", 3, True, "mt19937"), ("cleanup", 8, False, "counter"), ("harvest", 8, True, "counter")])
def test_policy_bytes_mod(kind, n, firing, rng):
    import torch
    E, T = 97, 40
    contract = "cleanup" if kind == "cleanup" else "harvest_local"
    env, orc = _pair(kind, E, n, contract=contract, firing=firing, horizon=17, auto_reset=True, rng=rng)
    A = env.num_actions
    rs = np.random.RandomState(3)
    for t in range(T):
        raw = rs.randint(0, 256, size=(E, n)).astype(np.uint8)
        dev = torch.from_numpy(raw).cuda()
        if t % 2 == 0:
            env.step_policy_device(dev.data_ptr(), "bytes")
        else:  # two slices, each its own launch
            env.step_policy_device(dev.data_ptr(), "bytes", 0, 40)
            env.step_policy_device(dev.data_ptr(), "bytes", 40, E - 40)
        env.synchronize()
        want = (raw % A).astype(np.uint8)
        assert np.array_equal(env.download("actions_taken"), want), "actions step %d" % t
        orc.step(want)
        _same(env, orc, "step %d" % t)
    env.check_faults()
    env.close()


@pytest.mark.parametrize("kind,n,firing", [("cleanup", 8, False), ("cleanup", 5, True), ("harvest", 8, False), ("harvest", 2, True)])
def test_policy_argmax_scores(kind, n, firing):
    import torch
    E, T = 64, 30
    env, orc = _pair(kind, E, n, contract=None, firing=firing, horizon=1000)
    A = env.num_actions
    rs = np.random.RandomState(11)
    for t in range(T):
        sc = rs.standard_normal((E, n, A)).astype(np.float32)
        if t % 3 == 0:  # ties: the first maximum wins; a NaN never does
            sc = np.round(sc * 2) / 2
        if t % 5 == 0:
            sc[rs.rand(E, n, A) < 0.1] = np.nan
            sc[..., A - 1][np.isnan(sc).all(axis=2)] = 0.0
        dev = torch.from_numpy(sc).cuda()
        env.step_policy_device(dev.data_ptr(), "argmax")
        env.synchronize()
        filled = np.where(np.isnan(sc), -np.inf, sc)
        want = filled.argmax(axis=2).astype(np.uint8)  # numpy: first occurrence of the maximum
        assert np.array_equal(env.download("actions_taken"), want), "actions step %d" % t
        orc.step(want)
        _same(env, orc, "step %d" % t)
    env.close()


def test_policy_step_misuse():
    from contracts_amd._lib import EngineError
    from contracts_amd.engine import BatchedEnv
    import torch
    sd = BatchedEnv("selfdrive", 8, 4)
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(EngineError):
        sd.step_policy_device(buf.data_ptr(), "bytes")
    sd.close()
    env = BatchedEnv("cleanup", 8, 4)
    env.seed(seed0=1)
    env.reset()
    with pytest.raises(EngineError):
        env.step_policy_device(buf.data_ptr(), "bytes", 4, 9)  # range past the last env
    with pytest.raises(EngineError):
        env.step_policy_device(0, "bytes")
    env.close()
