"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP engine, called through the C-ABI,
against (a) the golden fixtures produced by the reference itself and (b) the CPU oracle on
seeded random rollouts.  Integer / byte state is compared bit-exact; float64 rewards within 1e-9
(north-star tolerance is 1e-6)."""
import numpy as np
import pytest

import golden_check as gc

pytestmark = pytest.mark.gpu


def _engine(kind, E, n, **kw):
    from contracts_amd.engine import BatchedEnv
    return BatchedEnv(kind, E, n, **kw)


def test_selftest_wave_primitives():
    import ctypes as C
    from contracts_amd import _lib
    L = _lib.load()
    assert L.ce_device_count() >= 1
    mask = C.c_uint32(0xffffffff)
    rc = L.ce_selftest(0, C.byref(mask))
    assert rc == 0 and mask.value == 0, "selftest failed mask=%#x" % mask.value


GRID_FIXTURES = [f for f in gc.fixtures("g") if "selfdrive" not in f] + gc.fixtures("m")  # m_*: the reference on hand-made layouts


@pytest.mark.parametrize("name", GRID_FIXTURES)
def test_golden_grid(name):
    g = gc.load(name)
    kind, n, kw = gc.grid_kwargs(g)
    env = _engine(kind, 3, n, **kw)
    gc.replay_grid(g, env, env=2)
    env.check_faults()
    env.close()


@pytest.mark.parametrize("name", gc.fixtures("p"))
def test_golden_grid_counter_rng(name):
    """p_* fixtures: the reference itself, run with its np.random draw sites routed to the counter stream
    (tests/golden/make_counter_golden.py) — the HIP engine in rng="counter" mode against them, every field of every step plus
    the (key0, key1, generation) row after each construct / reset / step"""
    g = gc.load(name)
    assert str(g["rng_mode"]) == "counter"
    kind, n, kw = gc.grid_kwargs(g)
    env = _engine(kind, 3, n, rng="counter", **kw)
    gc.replay_grid(g, env, env=2)
    env.check_faults()
    env.close()


@pytest.mark.parametrize("name", gc.fixtures("g5_selfdrive"))
def test_golden_selfdrive(name):
    g = gc.load(name)
    env = _engine("selfdrive", 3, int(g["n"]), contract="selfdrive_distprop", collision_on=bool(int(g["collision_on"])))
    gc.replay_selfdrive(g, env, env=2)
    env.close()


FIELDS_GRID = ["grid", "agents", "spawn_perm", "rng", "timestep", "theta", "obs", "base_reward", "reward", "done", "info",
               "features", "int_metrics", "f64_metrics", "final_int_metrics", "final_f64_metrics"]


def _compare(env, orc, fields, tag):
    for f in fields:
        a, b = env.download(f), getattr(orc, f)
        if f == "rng":
            a, b = a[:, :625], b[:, :625]
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, err_msg="%s %s" % (f, tag))
        else:
            if not np.array_equal(a, b):
                bad = np.nonzero((a != b).reshape(a.shape[0], -1).any(axis=1))[0]
                raise AssertionError("%s mismatch %s in envs %s" % (f, tag, bad[:8]))


@pytest.mark.parametrize("kind,n,contract,firing,horizon,E,T", [
    ("cleanup", 8, "cleanup", False, 60, 512, 150),
    ("cleanup", 4, "cleanup", True, 1000, 256, 120),
    ("harvest", 8, "harvest_local", False, 50, 512, 120),
    ("harvest", 3, None, True, 1000, 128, 100),
    ("cleanup", 5, None, False, 40, 130, 90),
    # degenerate shapes: an episode per step (every step ends in an in-launch reset), one env, one agent, nine agents
    ("cleanup", 8, "cleanup", True, 1, 65, 40),
    ("harvest", 9, None, True, 2, 3, 60),
    ("cleanup", 1, None, False, 3, 1, 80),
    ("harvest", 1, None, True, 1000, 2, 80),
    ("cleanup", 9, None, True, 1000, 1, 60),
])
def test_random_rollout_vs_oracle(kind, n, contract, firing, horizon, E, T):
    """many envs, distinct seeds, auto-reset across episode boundaries; everything compared every step"""
    from oracle.pyoracle import Oracle
    kw = dict(contract=contract, firing=firing, horizon=horizon, auto_reset=True)
    env, orc = _engine(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 7919 + 12345
    env.seed(seeds)
    orc.seed(seeds)
    env.reset()
    orc.reset()
    fields = FIELDS_GRID + (["waste_perm"] if kind == "cleanup" else [])
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


@pytest.mark.parametrize("n,collision_on,null_prob", [(4, False, 0.0), (4, True, 0.0), (3, False, 0.0), (6, True, 0.0), (1, False, 0.0),
                                                      (2, False, 0.0), (8, False, 0.0), (10, True, 0.0),
                                                      # null_prob > 0: theta draws take 2 or 4 numpy words, so the window of a
                                                      # reset drifts off the 4-word grid and eventually straddles a generation
                                                      # end (the serial fallback); n = 5, 7: the same for the `random` stream
                                                      (4, False, 0.5), (5, True, 0.3), (7, False, 0.5)])
def test_selfdrive_random_vs_oracle(n, collision_on, null_prob):
    from oracle.pyoracle import Oracle
    E = 1030 if n == 4 else 203  # not a multiple of the envs per wave: the last wave is partly out of range
    kw = dict(contract="selfdrive_distprop", auto_reset=True, collision_on=collision_on, null_prob=null_prob)
    env, orc = _engine("selfdrive", E, n, **kw), Oracle("selfdrive", E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) + 5
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    rs = np.random.RandomState(3)
    for t in range(200 if null_prob == 0.0 else 900):  # ~15 resets per env at 900 steps: window crossings happen
        a = rs.uniform(-0.15, 0.15, size=(E, n)).astype(np.float32)
        env.step(a)
        orc.step(a)
        if null_prob != 0.0 and t % 50 != 49:
            continue
        for f in ("obs_f64", "reward", "sd_state", "theta", "sd_info", "f64_metrics"):
            np.testing.assert_allclose(env.download(f), getattr(orc, f), rtol=0, atol=1e-9, equal_nan=True,
                                       err_msg="%s step %d" % (f, t))
        for f in ("done", "done_agents", "info", "base_reward"):
            assert np.array_equal(env.download(f), getattr(orc, f)), "%s step %d" % (f, t)
    keep = np.r_[0:625, 628:1253]  # both MT19937 streams (np.random: theta draws, `random`: start positions) after the resets
    assert np.array_equal(env.download("rng")[:, keep], orc.rng[:, keep])
    env.close()


def test_bad_action_sets_fault():
    env = _engine("cleanup", 4, 2)
    env.seed(seed0=1)
    env.reset()
    a = np.zeros((4, 2), np.uint8)
    a[2, 1] = 9
    env.step(a)
    f = env.download("error_flags")
    assert f[2] & 1 and not f[[0, 1, 3]].any()
    env.close()


@pytest.mark.parametrize("kind,n,contract", [("cleanup", 4, "cleanup"), ("selfdrive", 3, "selfdrive_distprop")])
def test_state_checkpoint_roundtrip(tmp_path, kind, n, contract):
    """save -> keep stepping -> load into a fresh engine -> the same continuation, bit for bit"""
    E = 96
    a = _engine(kind, E, n, contract=contract, horizon=40, auto_reset=True)
    a.seed(seed0=7)
    a.reset()
    rs = np.random.RandomState(5)

    def acts():
        return rs.uniform(-0.15, 0.15, (E, n)).astype(np.float32) if kind == "selfdrive" else rs.randint(8, size=(E, n))

    for _ in range(30):
        a.step(acts())
    path = str(tmp_path / "ckpt.npz")
    a.save(path)
    st = rs.get_state()
    ref = []
    for _ in range(25):
        a.step(acts())
        ref.append((a.download("reward").copy(), a.download("obs_f64" if kind == "selfdrive" else "obs").copy()))
    b = _engine(kind, E, n, contract=contract, horizon=40, auto_reset=True)
    b.load(path)
    rs.set_state(st)
    for t in range(25):
        b.step(acts())
        assert np.array_equal(b.download("reward"), ref[t][0], equal_nan=True), t
        assert np.array_equal(b.download("obs_f64" if kind == "selfdrive" else "obs"), ref[t][1], equal_nan=True), t
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cleanup", "harvest"])
def test_packed_grid_roundtrip_and_validation(kind):
    """the device keeps 32 B of presence bits per env; ce_download / ce_upload("grid") expand to / pack from the map
    image: round trip is the identity, a blank (constructed, never reset) map reads as zeros, and an image that differs
    from the static map anywhere but on apple / waste cells raises CE_FAULT_BAD_GRID"""
    from contracts_amd.engine import BatchedEnv
    env = BatchedEnv(kind, 4, 3)
    env.seed(np.arange(4).astype(np.uint64) + 3)
    assert not env.download("grid").any()  # world_map is blank until the first reset
    env.reset()
    g = env.download("grid")
    assert g.any() and env.b.grid_env_stride == 32
    env.upload("grid", g)
    assert np.array_equal(env.download("grid"), g) and not env.download("error_flags").any()
    bits = env.download("grid")  # flip one legal cell and one illegal one
    apple = np.argwhere(np.isin(g[0], (0, 2)) & (np.arange(g.shape[2])[None, :] > 0))
    h = g.copy()
    if kind == "cleanup":
        waste = np.argwhere(g[1] == 3)[0]
        h[1, waste[0], waste[1]] = 4  # waste -> river on a waste cell: legal
    wall = np.argwhere(g[2] == 1)[5]
    h[2, wall[0], wall[1]] = 0        # a wall removed: illegal
    env.upload("grid", h)
    flags = env.download("error_flags")
    assert flags[2] & 8 and not flags[0] and not flags[1] and not flags[3]
    back = env.download("grid")
    assert np.array_equal(back[1], h[1]) and np.array_equal(back[2], g[2])  # the static map wins on env 2
    env.close()


FULL_SIZE = {  # BASELINE.json configs at their full single-GPU size
    "C4": dict(kind="cleanup", n=8, E=16384, contract="cleanup", horizon=12, T=30),
    "C2": dict(kind="cleanup", n=4, E=4096, contract="cleanup", horizon=12, T=30),
    "C3": dict(kind="harvest", n=8, E=16384, contract="harvest_local", horizon=12, T=30),
    "C5": dict(kind="selfdrive", n=4, E=32768, contract="selfdrive_distprop", horizon=1000, T=40),
    "C1": dict(kind="harvest_features", n=2, E=16384, contract="harvest_local", horizon=12, T=30),
}
FULL_FIELDS = {
    "cleanup": ("obs", "reward", "grid", "agents", "rng", "waste_perm", "features", "int_metrics", "f64_metrics", "theta"),
    "harvest": ("obs", "reward", "grid", "agents", "rng", "features", "int_metrics", "f64_metrics", "theta"),
    "selfdrive": ("obs_f64", "reward", "base_reward", "sd_state", "sd_info", "rng", "theta", "done", "done_agents", "info", "f64_metrics"),
    "harvest_features": ("reward", "apple_stamp", "next_stamp", "agents", "rng", "features", "int_metrics", "f64_metrics", "theta"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", list(FULL_SIZE))
def test_full_size_partition_invariance_and_oracle_sample(cfg):
    """Every BASELINE config at its full single-GPU size: the result of a rollout does not depend on how the env axis
    is cut or launched — one launch per step, three slices on three streams (bench.py's mode), fused multi-step launches
    (where built), or two engines that each own half of the global index range (the 8-GPU shard rule) — and a random
    sample of envs agrees with the CPU oracle stepping the same seeds and actions (tail slices, grid sizes and index
    arithmetic at the real E)."""
    import hashlib
    import torch
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    c = FULL_SIZE[cfg]
    kind, n, E, T, seed0 = c["kind"], c["n"], c["E"], c["T"], 73907
    kw = dict(contract=c["contract"], auto_reset=True)
    if kind != "selfdrive":
        kw["horizon"] = c["horizon"]  # several in-launch auto-resets inside the window
    fields = FULL_FIELDS[kind]
    fused_ok = True  # every family has a fused rollout kernel

    def digest(envs):
        h = hashlib.sha256()
        for f in fields:
            for env in envs:
                a = getattr(env, f) if f.endswith("_stamp") else env.download(f, raw=True)
                h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()

    whole = BatchedEnv(kind, E, n, **kw)
    acts = torch.empty((T, E, n), dtype=torch.float32 if kind == "selfdrive" else torch.uint8, device="cuda")
    whole.synth_actions(seed0 + 1, 0, T, acts.data_ptr())
    whole.synchronize()
    plane = E * n * acts.element_size()

    def run(mode):
        if mode == "halves":
            envs = [BatchedEnv(kind, E // 2, n, env_index_base=k * (E // 2), **kw) for k in range(2)]
        else:
            envs = [BatchedEnv(kind, E, n, **kw)]
        for env in envs:
            env.seed(seed0=seed0)
            env.reset()
        if mode == "single":
            for t in range(T):
                envs[0].step_device(acts.data_ptr() + t * plane)
        elif mode == "streams":
            streams = [torch.cuda.Stream() for _ in range(3)]
            envs[0].rollout_device(acts.data_ptr(), T, [s.cuda_stream for s in streams])
        elif mode == "fused":
            streams = [torch.cuda.Stream() for _ in range(3)]
            envs[0].rollout_fused(acts.data_ptr(), T, 7, None, [s.cuda_stream for s in streams])
        else:
            half = acts.reshape(T, 2, E // 2, n)
            for k, env in enumerate(envs):
                a_k = half[:, k].contiguous()
                env.rollout_device(a_k.data_ptr(), T, None)
        torch.cuda.synchronize()
        for env in envs:
            env.check_faults()
        return envs

    ref_envs = run("single")
    d0 = digest(ref_envs)
    for mode in ("streams", "halves") + (("fused",) if fused_ok else ()):
        envs = run(mode)
        assert digest(envs) == d0, mode
        for env in envs:
            env.close()
    # oracle on a sample of the global index range (first / last envs of the batch and of the stream slices included)
    rs = np.random.RandomState(4)
    edge = [0, 1, E // 3 - 1, E // 3, E // 3 + 1, E // 2 - 1, E // 2, 2 * E // 3, E - 2, E - 1]
    pick = np.unique(np.concatenate([rs.choice(E, size=38, replace=False), edge]))
    orc = Oracle(kind, len(pick), n, **kw)
    orc.seed((pick + seed0).astype(np.uint64))
    orc.reset()
    a_host = acts.cpu().numpy()
    for t in range(T):
        orc.step(a_host[t][pick])
    ref = ref_envs[0]
    for f in fields:
        a, b = getattr(ref, f)[pick], getattr(orc, f)
        if f == "rng":  # key words + position of the numpy stream (and of the CPython `random` stream where there is one)
            keep = np.r_[0:625, 628:1253] if a.shape[1] > 628 else np.r_[0:625]
            a, b = a[:, keep], b[:, keep]
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, equal_nan=True, err_msg=f)
        else:
            assert np.array_equal(a, b), f
    ref.close()
    whole.close()
    orc.close()


@pytest.mark.gpu
def test_eight_shard_global_range_on_one_gpu():
    """BASELINE config 3 at its GLOBAL size — cleanup_new n = 8 + CleanupContract, 131 072 envs sharded over 8 GPUs as
    16 384 per rank with env_index_base = g * 16 384 (runner.py:51-82: one env per worker, seeds keyed by the worker
    index) — on one GPU: one handle that owns all 131 072 envs against eight handles that own one shard each, stepped
    the way bench.py steps a rank (three slices on three streams), a horizon short enough for several in-launch resets.
    Every persistent field and every output must be byte-identical, and a sample of envs from shards 0, 3 and 7 (global
    indices >= 114 688 included) must agree with the CPU oracle seeded with the same global indices."""
    import torch
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    kind, n, G, Eg, T, seed0 = "cleanup", 8, 8, 16384, 26, 73907
    E = G * Eg
    kw = dict(contract="cleanup", auto_reset=True, horizon=11)
    fields = FULL_FIELDS[kind] + ("timestep", "done", "info", "base_reward", "spawn_perm")
    whole = BatchedEnv(kind, E, n, **kw)
    acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
    whole.synth_actions(seed0 + 1, 0, T, acts.data_ptr())  # keyed by the global env index
    whole.synchronize()  # (null stream; the slices' streams below are non-blocking)
    whole.seed(seed0=seed0)
    whole.reset()
    streams = [torch.cuda.Stream() for _ in range(3)]
    handles = [s.cuda_stream for s in streams]
    whole.rollout_device(acts.data_ptr(), T, handles)
    torch.cuda.synchronize()
    whole.check_faults()
    shards = []
    for g in range(G):
        env = BatchedEnv(kind, Eg, n, env_index_base=g * Eg, **kw)
        a_g = acts[:, g * Eg:(g + 1) * Eg].contiguous()
        # the shard's own generator must give the planes the whole-batch call gave for its index range
        chk = torch.empty_like(a_g)
        env.synth_actions(seed0 + 1, 0, T, chk.data_ptr())
        env.synchronize()
        assert torch.equal(chk, a_g), "synthetic actions of shard %d are not keyed by the global index" % g
        env.seed(seed0=seed0)
        env.reset()
        env.rollout_device(a_g.data_ptr(), T, handles)
        torch.cuda.synchronize()
        env.check_faults()
        shards.append(env)
    # every field compared ON THE DEVICE, byte for byte (the whole batch is ~1.3 GB of state and outputs: host copies of
    # both sides would dominate the test): raw views of the engine's buffers through __cuda_array_interface__
    from contracts_amd.engine import _DevArray, _FIELD_DTYPES

    def dev_bytes(env, f):
        per_env = int(np.prod(env._env_shape(f), dtype=np.int64)) * np.dtype(_FIELD_DTYPES[f]).itemsize
        if f == "grid":
            per_env = env.b.grid_env_stride  # the packed presence bits as they sit in HBM
        return torch.as_tensor(_DevArray(getattr(env.b, f), (env.E, per_env), np.uint8, None, env), device="cuda")

    for f in fields:
        w = dev_bytes(whole, f)
        for g, env in enumerate(shards):
            assert torch.equal(w[g * Eg:(g + 1) * Eg], dev_bytes(env, f)), "%s differs in shard %d" % (f, g)
    assert int(whole.download("int_metrics")[:, 0].sum()) >= 0 and whole.download("timestep").max() <= 11
    # oracle sample: first / last envs of shards 0, 3, 7 + random picks inside them
    rs = np.random.RandomState(8)
    pick = []
    for g in (0, 3, 7):
        pick += [g * Eg, g * Eg + 1, (g + 1) * Eg - 1] + list(g * Eg + rs.choice(Eg, size=13, replace=False))
    pick = np.unique(np.array(pick))
    assert pick.max() >= 114688
    orc = Oracle(kind, len(pick), n, **kw)
    orc.seed((pick + seed0).astype(np.uint64))
    orc.reset()
    a_host = acts[:, torch.from_numpy(pick).cuda()].cpu().numpy()
    for t in range(T):
        orc.step(a_host[t])
    for f in FULL_FIELDS[kind] + ("timestep", "done", "info", "base_reward"):
        a, b = np.concatenate([whole.download(f, int(e), 1) for e in pick]), getattr(orc, f)
        if f == "rng":
            a, b = a[:, :625], b[:, :625]
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9, err_msg=f)
        else:
            assert np.array_equal(a, b), f
    for env in shards:
        env.close()
    whole.close()
    orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract", [("cleanup", 4, "cleanup"), ("harvest", 5, "harvest_local"), ("cleanup_features", 3, "cleanup")])
def test_external_theta_one_contract_per_env(kind, n, contract):
    """CE_FLAG_EXTERNAL_THETA: the caller writes one contract parameter per env (batched evaluation of many sampled
    contracts at once); resets draw nothing from np.random and leave the parameters alone"""
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    E, T = 128, 70
    kw = dict(contract=contract, horizon=23, auto_reset=True, external_theta=True)
    env, orc = BatchedEnv(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    rs = np.random.RandomState(12)
    thetas = rs.uniform(0.0, 0.2 if contract == "cleanup" else 10.0, size=E)
    for o in (env, orc):
        o.seed(seed0=4242)
    env.upload("theta", thetas)
    orc.theta[...] = thetas
    orc.import_state()
    for o in (env, orc):
        o.reset()
    assert np.array_equal(env.download("theta"), thetas) and np.array_equal(orc.theta, thetas)
    na = env.num_actions
    p = np.array([.1, .1, .15, .1, .05, .1, .1, .3]) if contract == "cleanup" else None
    for t in range(T):
        a = rs.choice(na, size=(E, n), p=p).astype(np.uint8)
        env.step(a)
        orc.step(a)
        np.testing.assert_allclose(env.download("reward"), orc.reward, rtol=0, atol=1e-9)
        assert np.array_equal(env.download("rng").reshape(E, -1, 628)[:, :, :625], orc.rng.reshape(E, -1, 628)[:, :, :625])
        assert np.array_equal(env.download("theta"), thetas)
        np.testing.assert_allclose(env.download("f64_metrics"), orc.f64_metrics, rtol=0, atol=1e-9)
    assert np.abs(env.download("final_f64_metrics")[:, 0]).max() > 0  # transfers did happen
    env.close()
    orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", gc.fixtures("render_"))
def test_golden_render(name):
    """CE_FLAG_BEAM_TRACE: the step's beam cells (MapEnv.beam_pos) and the composed full_map_to_colors frame"""
    g = gc.load(name)
    env = _engine(str(g["kind"]), 3, int(g["n"]), horizon=int(g["horizon"]), firing=True, beam_trace=True)
    gc.replay_render(g, env, env=2)
    env.check_faults()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,firing,horizon", [("cleanup", 8, True, 35), ("cleanup", 3, False, 1000), ("harvest", 7, True, 30)])
def test_beam_trace_random_vs_oracle(kind, n, firing, horizon):
    """beam-heavy random rollouts with auto-reset (a done step's beams are cleared by the in-launch reset, as reset()
    clears beam_pos); the trace changes nothing else, and switching it off leaves the map untouched"""
    from oracle.pyoracle import Oracle
    E, T = 192, 80
    kw = dict(firing=firing, horizon=horizon, auto_reset=True)
    env, orc = _engine(kind, E, n, beam_trace=True, **kw), Oracle(kind, E, n, beam_trace=True, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 104729 + 5
    for impl in (env, orc):
        impl.seed(seeds)
        impl.reset()
    fields = FIELDS_GRID + ["beam_map"] + (["waste_perm"] if kind == "cleanup" else [])
    rs = np.random.RandomState(5)
    na = env.num_actions
    p = np.full(na, 0.5 / 7)
    p[7:] = 0.5 / (na - 7)
    seen = 0
    for t in range(T):
        a = rs.choice(na, size=(E, n), p=p).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _compare(env, orc, fields, "step %d" % t)
        seen += int(orc.beam_map.any())
    assert seen > T // 2
    last = env.download("beam_map")
    env.set_flags(beam_trace=False)
    env.step(a)
    assert np.array_equal(env.download("beam_map"), last)  # not written (and not cleared) while the trace is off
    env.check_faults()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 4, 8])
def test_selfdrive_reset_windows_across_generation_ends(n):
    """Resets whose stream windows straddle the end of an MT19937 generation, every split: env b starts with the CPython
    stream at position 624 - (b mod (2n + 2)) (a reset draws 2n words: 0 .. 2n + 1 of them from the old generation) and the
    numpy stream at 624 - (b // (2n + 2) mod 6) (theta draws 2 or 4 words depending on u0 > null_prob: the regeneration must
    happen exactly when a consumed word lies past the end).  Explicit ce_reset, then in-launch auto-resets, per-step and
    fused, against the oracle's word-by-word genrand."""
    import torch
    from oracle.pyoracle import Oracle
    E = 12 * (2 * n + 2) * 2
    kw = dict(contract="selfdrive_distprop", auto_reset=True, null_prob=0.5)
    env, orc = _engine("selfdrive", E, n, **kw), Oracle("selfdrive", E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 31 + 7
    for o in (env, orc):
        o.seed(seeds)
        o.reset()  # generations now filled; positions are overwritten below
    b = np.arange(E)
    rng = env.download("rng").copy()
    assert np.array_equal(rng[:, :625], orc.rng[:, :625]) and np.array_equal(rng[:, 628:1253], orc.rng[:, 628:1253])
    rng[:, 628 + 624] = 624 - (b % (2 * n + 2))
    rng[:, 624] = 624 - ((b // (2 * n + 2)) % 6)
    env.upload("rng", rng)
    orc.rng[...] = rng
    orc.import_state()
    keep = np.r_[0:625, 628:1253]

    def same(tag):
        for f in ("obs_f64", "reward", "sd_state", "theta", "sd_info"):
            np.testing.assert_allclose(env.download(f), getattr(orc, f), rtol=0, atol=1e-9, equal_nan=True, err_msg="%s %s" % (f, tag))
        assert np.array_equal(env.download("rng")[:, keep], orc.rng[:, keep]), "streams " + tag
        assert np.array_equal(env.download("done"), orc.done), tag

    for o in (env, orc):
        o.reset()  # explicit reset: every split of both windows at once
    same("after explicit reset")
    # in-launch auto-resets: park the positions near the end again and step until every env has reset at least once
    rng = env.download("rng").copy()
    rng[:, 628 + 624] = 624 - ((b + 3) % (2 * n + 2))
    rng[:, 624] = 624 - ((b // (2 * n + 2) + 1) % 6)
    env.upload("rng", rng)
    orc.rng[...] = rng
    orc.import_state()
    rs = np.random.RandomState(n)
    resets = np.zeros(E, bool)
    for t in range(160):
        a = rs.uniform(0.05, 0.1, size=(E, n)).astype(np.float32)  # everybody accelerates: episodes end quickly
        env.step(a)
        orc.step(a)
        resets |= orc.done.astype(bool)
        if t % 20 == 19:
            same("step %d" % t)
    assert resets.all()
    same("after per-step auto-resets")
    # the fused kernel over the same situation
    rng = env.download("rng").copy()
    rng[:, 628 + 624] = 624 - ((b + 1) % (2 * n + 2))
    rng[:, 624] = 624 - ((b // (2 * n + 2) + 2) % 6)
    env.upload("rng", rng)
    orc.rng[...] = rng
    orc.import_state()
    acts = torch.from_numpy(rs.uniform(0.05, 0.1, size=(150, E, n)).astype(np.float32)).cuda()
    env.rollout_fused(acts.data_ptr(), 150, 37)
    env.synchronize()
    a_host = acts.cpu().numpy()
    for t in range(150):
        orc.step(a_host[t])
    same("after fused auto-resets")
    env.close()
    orc.close()


@pytest.mark.parametrize("kind,rng,fused", [("cleanup", "mt19937", False), ("cleanup", "mt19937", True), ("cleanup", "counter", False),
                                            ("cleanup", "counter", True), ("harvest", "mt19937", True), ("harvest", "counter", False)])
def test_custom_layout_rollouts_vs_oracle(kind, rng, fused):
    """hand-made layouts (ce_config.ascii_map) through the CM kernel instances: many envs, distinct seeds, in-launch resets,
    per-step and fused launches, both streams, beams traced — everything compared after every launch.  The small cleanup layout
    has 24 waste cells (the one-register list walk), the other 90 (the two-register path the shipped layout takes)."""
    import sys
    import os
    import torch
    sys.path.insert(0, os.path.join(gc.GOLDEN_DIR))
    from make_golden import CLEANUP_MID, CLEANUP_SMALL, HARVEST_SMALL
    from oracle.pyoracle import Oracle
    for rows, n in (((CLEANUP_SMALL, 3), (CLEANUP_MID, 4)) if kind == "cleanup" else ((HARVEST_SMALL, 5),)):
        E, T = 150, 90
        kw = dict(contract="cleanup" if kind == "cleanup" else "harvest_local", firing=True, horizon=25, auto_reset=True,
                  beam_trace=True, rng=rng, ascii_map=rows)
        env, orc = _engine(kind, E, n, **kw), Oracle(kind, E, n, **kw)
        assert env.download("grid").shape == (E, len(rows), len(rows[0])) == orc.grid.shape
        seeds = np.arange(E, dtype=np.uint64) * 131 + 9
        for o in (env, orc):
            o.seed(seeds)
            o.reset()
        fields = [f for f in FIELDS_GRID if not (rng == "counter" and f == "rng")] + (["waste_perm"] if kind == "cleanup" else []) + ["beam_map"]
        _compare(env, orc, fields, "after reset")
        rs = np.random.RandomState(5)
        na = env.num_actions
        p = np.full(na, 0.6 / (na - 1))
        p[7 if kind == "cleanup" else 2] = 0.4
        t = 0
        while t < T:
            c = int(rs.randint(1, 8)) if fused else 1
            a = rs.choice(na, size=(c, E, n), p=p).astype(np.uint8)
            if fused:
                dev = torch.from_numpy(a).cuda()
                env.rollout_fused(dev.data_ptr(), c, int(rs.choice([0, 3])))
                env.synchronize()
            else:
                env.step(a[0])
            for k in range(c):
                orc.step(a[k])
            t += c
            _compare(env, orc, fields, "step %d" % t)
            if rng == "counter":
                assert np.array_equal(env.download("rng")[:, :3], orc.rng[:, :3])
        # ce_global_view on a caller's layout (round 6): the colour map of every env, agents painted (every env has stepped)
        from contracts_amd.environments.map_env import AGENT_RGB, CELL_RGB
        gv = torch.empty((E, len(rows), len(rows[0]), 3), dtype=torch.uint8, device="cuda")
        env.global_view(gv.data_ptr())
        env.synchronize()
        want = CELL_RGB[orc.grid].copy()
        steps_taken = orc.timestep
        for e in range(E):
            if steps_taken[e] > 0:  # (an env whose last launch ended in an in-launch reset shows its map without agents)
                for i, ag in enumerate(orc.agents[e]):
                    want[e, ag[0], ag[1]] = AGENT_RGB[i]
        assert np.array_equal(gv.cpu().numpy(), want)
        env.check_faults()
        env.close()
