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


@pytest.mark.parametrize("kind,n,firing,rng,sliced", [("cleanup", 8, False, "mt19937", True), ("cleanup", 3, True, "mt19937", False),
                                                      ("harvest", 8, False, "mt19937", True), ("harvest", 5, True, "counter", False),
                                                      ("cleanup", 8, False, "counter", True)])
def test_policy_ahead_noise_closes_the_loop_in_the_kernel(kind, n, firing, rng, sliced):
    """CE_POLICY_AHEAD_NOISE: the benchmark's closed-loop policy inside the step kernel — the env's noise byte moves on by the
    green channel of view pixel (6, 7) of the PREVIOUS observation (the one a reset or the last step left in `obs`), the action is
    the new byte mod |A|.  A host restatement of that rule over the oracle's observations + the oracle stepped with the resulting
    actions must reproduce every field, the noise plane and `actions_taken`, across in-launch auto-resets (the reset observation
    is the next tick's input), as single launches, explicit slices and ce_step_policy_sliced on three streams."""
    import torch
    E, T = 101, 60
    contract = "cleanup" if kind == "cleanup" else "harvest_local"
    env, orc = _pair(kind, E, n, contract=contract, firing=firing, horizon=13, auto_reset=True, rng=rng)
    A = env.num_actions
    rs = np.random.RandomState(5)
    noise = rs.randint(0, 256, size=(E, n)).astype(np.uint8)
    dev = torch.from_numpy(noise.copy()).cuda()
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    for t in range(T):
        ahead = np.asarray(orc.obs)[:, :, 6, 7, 1]
        noise = (noise.astype(np.uint32) + ahead).astype(np.uint8)  # modulo 256
        want = (noise % A).astype(np.uint8)
        if sliced and t % 3 == 0:
            env.step_policy_sliced(dev.data_ptr(), "ahead_noise", [s.cuda_stream for s in streams])
        elif t % 3 == 1:
            env.step_policy_device(dev.data_ptr(), "ahead_noise", 0, 37)
            env.step_policy_device(dev.data_ptr(), "ahead_noise", 37, E - 37)
        else:
            env.step_policy_device(dev.data_ptr(), "ahead_noise")
        torch.cuda.synchronize()
        assert np.array_equal(env.download("actions_taken"), want), "actions step %d" % t
        assert np.array_equal(dev.cpu().numpy(), noise), "noise plane step %d" % t
        orc.step(want)
        _same(env, orc, "step %d" % t)
    assert len(np.unique(env.download("actions_taken"))) == A  # the loop wanders over the whole action space
    env.check_faults()
    env.close()
