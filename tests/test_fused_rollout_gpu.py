"""Fused multi-step rollouts (ce_rollout_fused) against per-step launches and the CPU oracle.

The fused kernels keep an env's state on chip for T consecutive steps; the contract is that state, per-step outputs
(every plane of the trajectory), metrics and RNG streams are those of T single-step launches, bit for bit — which are in
turn pinned to the oracle / the reference's golden traces by test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATE_GRID = ["grid", "agents", "spawn_perm", "rng", "timestep", "theta", "int_metrics", "f64_metrics", "final_int_metrics",
              "final_f64_metrics", "error_flags"]
OUT_GRID = ["obs", "base_reward", "reward", "done", "info", "features"]


def _pair(kind, E, n, **kw):
    from contracts_amd.engine import BatchedEnv
    a, b = BatchedEnv(kind, E, n, **kw), BatchedEnv(kind, E, n, **kw)
    seeds = np.arange(E, dtype=np.uint64) * 31 + 73907
    for e in (a, b):
        e.seed(seeds)
        e.reset()
    return a, b


def _actions(env, T, key=5, bad_at=None):
    import torch
    acts = torch.empty((T, env.E, env.n), dtype=torch.float32 if env.kind == "selfdrive" else torch.uint8, device="cuda")
    env.synth_actions(key, 0, T, acts.data_ptr())
    env.synchronize()
    if bad_at is not None:  # an action outside the table: that env skips the step and raises its fault bit
        t, e = bad_at
        acts[t, e, 0] = 99
    return acts


def _same(a, b, fields, tag):
    for f in fields:
        x, y = a.download(f, raw=True), b.download(f, raw=True)
        assert x.tobytes() == y.tobytes(), "%s differs (%s)" % (f, tag)


@pytest.mark.parametrize("kind,n,contract,firing,horizon,T,per,extra", [
    ("cleanup", 8, "cleanup", False, 1000, 1, 0, None),
    ("cleanup", 8, "cleanup", False, 1000, 7, 0, None),
    ("cleanup", 8, "cleanup", False, 25, 64, 0, None),       # auto-reset inside the launch, twice
    ("cleanup", 8, "cleanup", False, 25, 64, 5, None),       # 13 launches of <= 5 steps
    ("cleanup", 4, "cleanup", True, 1000, 40, 16, None),     # FIRE + CLEAN beams
    ("cleanup", 9, None, True, 30, 50, 0, None),             # n = 9: odd feature pitch, serial paint path
    ("harvest", 8, "harvest_local", False, 20, 64, 0, None),
    ("harvest", 5, None, True, 1000, 33, 8, None),
    ("cleanup", 4, "cleanup", True, 20, 45, 0, dict(inequity=True, alpha=5.0, beta=0.05)),  # float reward accumulators
    ("harvest", 3, "harvest_local", True, 1000, 30, 4, dict(collective=True)),
    ("cleanup", 8, "cleanup", True, 1, 20, 0, None),         # an episode per step: every step resets inside the launch
    ("harvest", 1, None, False, 2, 30, 7, None),             # one agent
])
def test_fused_equals_per_step(kind, n, contract, firing, horizon, T, per, extra):
    E = 193
    kw = dict(contract=contract, firing=firing, horizon=horizon, auto_reset=True, **(extra or {}))
    fused, ref = _pair(kind, E, n, **kw)
    acts = _actions(fused, T)
    traj = fused.alloc_trajectory(T)
    fused.rollout_fused(acts.data_ptr(), T, per, traj)
    fused.synchronize()
    state = STATE_GRID + (["waste_perm"] if kind == "cleanup" else [])
    plane = E * n
    host = {f: traj.tensors[f].cpu().numpy() for f in traj.tensors}
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * plane)
        for f in OUT_GRID:
            want = ref.download(f, raw=True)
            got = host[f][t].reshape(want.shape)
            assert got.tobytes() == want.tobytes(), "%s plane %d" % (f, t)
    _same(fused, ref, state, "state after %d steps" % T)
    fused.check_faults()
    fused.close()
    ref.close()


def test_fused_without_trajectory_leaves_last_step_outputs():
    E, n, T = 130, 8, 12
    fused, ref = _pair("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
    acts = _actions(fused, T)
    fused.rollout_fused(acts.data_ptr(), T)  # outputs go to the per-step buffers, every step
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * E * n)
    _same(fused, ref, STATE_GRID + ["waste_perm"] + OUT_GRID, "no trajectory")
    fused.close()
    ref.close()


def test_fused_ring_and_partial_fields():
    """a ring of 4 planes over 10 steps keeps the last 4 steps; only obs + reward are kept as trajectories"""
    E, n, T, P = 70, 4, 10, 4
    fused, ref = _pair("harvest", E, n, contract="harvest_local", horizon=1000, auto_reset=True)
    acts = _actions(fused, T)
    traj = fused.alloc_trajectory(P, fields=("obs", "reward"))
    fused.rollout_fused(acts.data_ptr(), 6, 0, traj)
    fused.rollout_fused(acts.data_ptr() + 6 * E * n, 4, 3, traj)
    fused.synchronize()
    obs, rew = traj.tensors["obs"].cpu().numpy(), traj.tensors["reward"].cpu().numpy()
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * E * n)
        if t >= T - P:
            assert obs[t % P].tobytes() == ref.download("obs", raw=True).tobytes(), t
            assert rew[t % P].tobytes() == ref.download("reward").tobytes(), t
    _same(fused, ref, STATE_GRID + ["info", "features", "done", "base_reward"], "ring")
    fused.close()
    ref.close()


def test_fused_bad_action_skips_the_step_and_raises_the_fault():
    from contracts_amd import _lib
    E, n, T = 66, 8, 9
    fused, ref = _pair("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
    acts = _actions(fused, T, bad_at=(4, 17))
    fused.rollout_fused(acts.data_ptr(), T)
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * E * n)
    _same(fused, ref, STATE_GRID + ["waste_perm"], "bad action")
    flags = fused.download("error_flags")
    assert flags[17] == _lib.FAULT_BAD_ACTION and flags.sum() == _lib.FAULT_BAD_ACTION
    assert fused.download("timestep")[17] == T - 1
    fused.close()
    ref.close()


def test_fused_full_episode_vs_oracle_sample():
    """BASELINE headline size: 16 384 envs x 8 agents, one whole 1000-step episode + auto-reset in fused launches of 64;
    a sample of envs is replayed on the CPU oracle (same seeds, same synthetic actions) and compared at the end, and the
    whole batch must equal the per-step engine's digest."""
    import hashlib
    from oracle.pyoracle import Oracle
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    E, n, T = 16384, 8, 1003
    kw = dict(contract="cleanup", horizon=1000, auto_reset=True)
    fused, ref = BatchedEnv("cleanup", E, n, **kw), BatchedEnv("cleanup", E, n, **kw)
    for e in (fused, ref):
        e.seed(seed0=73907)
        e.reset()
    acts = _actions(fused, T, key=73908)
    traj = fused.alloc_trajectory(8, fields=("reward", "done", "info"))
    fused.rollout_fused(acts.data_ptr(), T, 64, traj)
    ref.rollout_device(acts.data_ptr(), T)
    fused.synchronize()
    fields = STATE_GRID + ["waste_perm"] + [f for f in OUT_GRID if f not in traj.tensors]
    for f in fields:
        x, y = fused.download(f, raw=True), ref.download(f, raw=True)
        assert hashlib.sha256(x.tobytes()).digest() == hashlib.sha256(y.tobytes()).digest(), f
    last = (T - 1) % traj.P  # the outputs kept as trajectories: the last step's plane is what the per-step engine holds
    for f in traj.tensors:
        assert traj.tensors[f][last].cpu().numpy().tobytes() == ref.download(f, raw=True).tobytes(), f
    # oracle sample: envs spread over the batch, identical actions (the counter hash is reproducible on the host)
    sample = np.array([0, 1, 63, 64, 4095, 8191, 8192, 12345, 16383])
    L = _lib.load()
    orc = Oracle("cleanup", len(sample), n, **kw)
    orc.seed(sample.astype(np.uint64) + 73907)
    orc.reset()
    a = acts.cpu().numpy()[:, sample, :]
    for t in range(T):
        orc.step(a[t])
    for f in ["grid", "agents", "timestep", "obs", "base_reward", "features", "int_metrics", "final_int_metrics",
              "waste_perm", "spawn_perm"]:
        assert np.array_equal(fused.download(f)[sample], getattr(orc, f)), f
    assert np.array_equal(fused.download("rng")[sample, :625], orc.rng[:, :625])
    assert np.array_equal(traj.tensors["info"][last].cpu().numpy()[sample], orc.info)
    np.testing.assert_allclose(traj.tensors["reward"][last].cpu().numpy()[sample], orc.reward, rtol=0, atol=1e-9)
    np.testing.assert_allclose(fused.download("final_f64_metrics")[sample], orc.final_f64_metrics, rtol=0, atol=1e-9)
    assert (fused.download("timestep") == 3).all()
    fused.check_faults()
    fused.close()
    ref.close()


@pytest.mark.parametrize("n,collision_on,T,per", [(4, False, 150, 0), (4, True, 90, 16), (3, False, 120, 7), (6, False, 80, 0),
                                                  (1, False, 70, 0), (10, True, 70, 16), (2, True, 60, 5)])
def test_fused_selfdrive_equals_per_step(n, collision_on, T, per):
    """selfdrive: cars resident in registers across the steps of a launch; several episodes (in-launch auto-resets,
    both MT19937 streams advanced by them) inside the window"""
    E = 517
    fused, ref = _pair("selfdrive", E, n, contract="selfdrive_distprop", auto_reset=True, collision_on=collision_on)
    acts = _actions(fused, T)
    traj = fused.alloc_trajectory(T)
    fused.rollout_fused(acts.data_ptr(), T, per, traj)
    fused.synchronize()
    host = {f: traj.tensors[f].cpu().numpy() for f in traj.tensors}
    plane = E * n * 4
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * plane)
        for f in ("obs_f64", "base_reward", "reward", "done", "done_agents", "info", "sd_info"):
            want = ref.download(f, raw=True)
            assert host[f][t].reshape(want.shape).tobytes() == want.tobytes(), "%s plane %d" % (f, t)
    _same(fused, ref, ["sd_state", "rng", "theta", "f64_metrics", "final_f64_metrics", "error_flags"], "selfdrive state")
    fused.check_faults()
    fused.close()
    ref.close()


@pytest.mark.parametrize("kind,n,contract,horizon,T,per", [("harvest_features", 2, "harvest_local", 17, 60, 0),
                                                          ("harvest_features", 5, None, 1000, 40, 9),
                                                          ("cleanup_features", 2, "cleanup", 13, 60, 16),
                                                          ("cleanup_features", 8, None, 1000, 35, 0)])
def test_fused_feature_envs_equal_per_step(kind, n, contract, horizon, T, per):
    """HarvestFeatures / CleanupFeatures: list stamps, agents and the CPython `random` stream resident across the steps
    of a launch; in-launch resets draw from both generators"""
    E = 150
    fused, ref = _pair(kind, E, n, contract=contract, horizon=horizon, auto_reset=True)
    acts = _actions(fused, T)
    traj = fused.alloc_trajectory(T)
    fused.rollout_fused(acts.data_ptr(), T, per, traj)
    fused.synchronize()
    host = {f: traj.tensors[f].cpu().numpy() for f in traj.tensors}
    for t in range(T):
        ref.step_device(acts.data_ptr() + t * E * n)
        for f in ("features", "base_reward", "reward", "done", "info"):
            want = ref.download(f, raw=True)
            assert host[f][t].reshape(want.shape).tobytes() == want.tobytes(), "%s plane %d" % (f, t)
    _same(fused, ref, ["grid", "agents", "rng", "timestep", "theta", "int_metrics", "f64_metrics", "final_int_metrics",
                       "final_f64_metrics", "error_flags"], "feature env state")
    fused.check_faults()
    fused.close()
    ref.close()
