"""The counter-RNG mode (CE_FLAG_RNG_COUNTER) of the grid kernels on the MI355X: the HIP path against the CPU oracle's
restatement of the same stream on every field, every step; fused rollouts against per-step launches; the state round trip.
The mode is the engine's own (16 bytes of generator state per env instead of 2 512) — reference parity is the MT19937 mode's
claim (test_gpu_parity.py); tests/test_counter_rng.py pins the generator itself to its published vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_parity import FIELDS_GRID, _compare, _engine  # noqa: E402
from test_fused_rollout_gpu import OUT_GRID, STATE_GRID, _actions, _same  # noqa: E402


@pytest.mark.parametrize("kind,n,contract,firing,horizon,E,T", [
    ("cleanup", 8, "cleanup", False, 60, 512, 150),
    ("cleanup", 4, "cleanup", True, 1000, 256, 120),
    ("harvest", 8, "harvest_local", False, 50, 512, 120),
    ("harvest", 3, None, True, 1000, 128, 100),
    ("cleanup", 5, None, False, 40, 130, 90),
    ("cleanup", 8, "cleanup", True, 1, 65, 40),   # an episode per step: every step resets inside the launch
    ("harvest", 9, None, True, 2, 3, 60),
    ("cleanup", 1, None, False, 3, 1, 80),
    ("cleanup", 9, None, True, 1000, 1, 60),
])
def test_counter_rollout_vs_oracle(kind, n, contract, firing, horizon, E, T):
    from oracle.pyoracle import Oracle
    kw = dict(contract=contract, firing=firing, horizon=horizon, auto_reset=True, rng="counter")
    env, orc = _engine(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    assert env.b.rng_words == 4
    seeds = np.arange(E, dtype=np.uint64) * 7919 + 12345
    env.seed(seeds)
    orc.seed(seeds)
    fields = FIELDS_GRID + (["waste_perm"] if kind == "cleanup" else [])
    _compare(env, orc, ["agents", "spawn_perm", "rng"], "after construct")
    env.reset()
    orc.reset()
    _compare(env, orc, fields, "after reset")
    rs = np.random.RandomState(99)
    na = env.num_actions
    p = None
    if kind == "cleanup":  # CLEAN-heavy so that the spawn model is exercised
        p = np.full(na, 0.6 / (na - 1))
        p[7] = 0.4
    for t in range(T):
        a = rs.choice(na, size=(E, n), p=p).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _compare(env, orc, fields, "step %d" % t)
    env.check_faults()
    env.close()


@pytest.mark.parametrize("kind,n,contract,firing,horizon,T,per", [
    ("cleanup", 8, "cleanup", False, 25, 64, 0),       # auto-reset inside the launch, twice
    ("cleanup", 8, "cleanup", False, 1000, 40, 16),
    ("cleanup", 4, "cleanup", True, 20, 45, 7),
    ("harvest", 8, "harvest_local", False, 20, 64, 0),
    ("harvest", 5, None, True, 1000, 33, 8),
    ("cleanup", 8, "cleanup", True, 1, 20, 0),         # every step resets inside the launch
])
def test_counter_fused_equals_per_step(kind, n, contract, firing, horizon, T, per):
    from contracts_amd.engine import BatchedEnv
    E = 193
    kw = dict(contract=contract, firing=firing, horizon=horizon, auto_reset=True, rng="counter")
    fused, ref = BatchedEnv(kind, E, n, **kw), BatchedEnv(kind, E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 31 + 73907
    for e in (fused, ref):
        e.seed(seeds)
        e.reset()
    acts = _actions(fused, T)
    traj = fused.alloc_trajectory(T)
    fused.rollout_fused(acts.data_ptr(), T, per, traj)
    fused.synchronize()
    host = {f: traj.tensors[f].cpu().numpy() for f in traj.tensors}
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * E * n)
        for f in OUT_GRID:
            want = ref.download(f, raw=True)
            assert host[f][t].reshape(want.shape).tobytes() == want.tobytes(), "%s plane %d" % (f, t)
    _same(fused, ref, STATE_GRID + (["waste_perm"] if kind == "cleanup" else []), "state after %d steps" % T)
    fused.check_faults()
    fused.close()
    ref.close()


def test_counter_headline_size_slices_and_oracle_sample():
    """BASELINE's headline shape in counter mode: 16 384 envs x 8 agents, 40 steps launched as three slices per step, then 32 fused;
    envs 0..47 and the last 48 against the oracle, every field"""
    from oracle.pyoracle import Oracle
    E, n, S = 16384, 8, 48
    kw = dict(contract="cleanup", horizon=30, auto_reset=True, rng="counter")
    env = _engine("cleanup", E, n, **kw)
    env.seed(seed0=555)
    env.reset()
    acts = _actions(env, 72, key=11)
    host_acts = acts.cpu().numpy()
    bounds = [0, 5461, 10922, E]
    for t in range(40):
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            env.step_range_device(acts.data_ptr() + t * E * n, lo, hi - lo)
    env.rollout_fused(acts.data_ptr() + 40 * E * n, 32, 16)
    env.synchronize()
    fields = FIELDS_GRID + ["waste_perm"]
    got = {f: env.download(f) for f in fields}
    for base in (0, E - S):
        orc = Oracle("cleanup", S, n, env_index_base=base, **kw)
        orc.seed(seed0=555)
        orc.reset()
        for t in range(72):
            orc.step(np.ascontiguousarray(host_acts[t, base:base + S]))
        for f in fields:
            a, b = got[f][base:base + S], getattr(orc, f)
            if a.dtype.kind == "f":
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, err_msg=f)
            else:
                assert np.array_equal(a, b), "%s, envs %d.." % (f, base)
    env.check_faults()
    env.close()


def test_counter_state_roundtrip_and_mode_is_part_of_the_state():
    from contracts_amd.engine import BatchedEnv
    E, n = 40, 4
    kw = dict(contract="cleanup", horizon=15, auto_reset=True)
    a, b = BatchedEnv("cleanup", E, n, rng="counter", **kw), BatchedEnv("cleanup", E, n, rng="counter", **kw)
    a.seed(seed0=9)
    a.reset()
    acts = _actions(a, 30)
    for t in range(12):
        a.step_device(acts.data_ptr() + t * E * n)
    state = a.state_dict()
    assert state["rng"].shape == (E, 4)
    b.load_state_dict(state)
    for t in range(12, 30):
        a.step_device(acts.data_ptr() + t * E * n)
        b.step_device(acts.data_ptr() + t * E * n)
    _same(a, b, STATE_GRID + ["waste_perm"] + OUT_GRID, "restored")
    mt = BatchedEnv("cleanup", E, n, **kw)
    with pytest.raises(Exception):
        mt.load_state_dict(state)  # other flags, other rng layout
    for e in (a, b, mt):
        e.close()


def test_counter_seeds_are_64_bit_keys():
    """ce_seed takes whole 64-bit seeds in counter mode (key = (low, high) words); the MT19937 mode keeps np.random.seed's range"""
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    E, n = 16, 3
    seeds = (np.arange(E, dtype=np.uint64) << np.uint64(33)) + np.uint64(0x1234567)
    kw = dict(contract="harvest_local", horizon=9, auto_reset=True, rng="counter")
    env, orc = BatchedEnv("harvest", E, n, **kw), Oracle("harvest", E, n, **kw)
    env.seed(seeds)
    orc.seed(seeds)
    env.reset()
    orc.reset()
    assert np.array_equal(env.download("rng")[:, 1], (seeds >> np.uint64(32)).astype(np.uint32))
    rs = np.random.RandomState(2)
    for t in range(30):
        a = rs.randint(0, 8, size=(E, n)).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _compare(env, orc, FIELDS_GRID, "step %d" % t)
    mt = BatchedEnv("harvest", E, n, contract="harvest_local")
    with pytest.raises(_lib.EngineError):
        mt.seed(seeds)
    env.close()
    mt.close()


def test_counter_mode_belongs_to_the_grid_kinds():
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    for kind, n in (("selfdrive", 4), ("harvest_features", 2)):
        with pytest.raises(_lib.EngineError):
            BatchedEnv(kind, 8, n, rng="counter")
