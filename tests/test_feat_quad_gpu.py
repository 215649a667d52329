"""k_feat_step_quad (HarvestFeatures, two agents, four envs per wave — BASELINE config 0's single-step kernel) against the
oracle: every persistent and output field after every step, with batch sizes that leave a partly filled wave, short horizons
(the episode end + reset take the one-env path inside the same launch), both contract settings, slices that start off a
multiple of four, and bad action ids in some envs (fault flag set, the env left untouched)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["grid", "agents", "rng", "timestep", "theta", "base_reward", "reward", "done", "info", "features", "int_metrics",
          "f64_metrics", "final_int_metrics", "final_f64_metrics", "error_flags"]


STATE = ["grid", "agents", "rng", "timestep", "theta", "int_metrics", "f64_metrics", "final_int_metrics", "final_f64_metrics", "error_flags"]


def _same(env, orc, E, tag, fields=FIELDS):
    for f in fields:
        x, y = env.download(f, raw=True) if f == "grid" else env.download(f), getattr(orc, f)
        if f == "rng":
            x, y = x.reshape(E, 2, 628)[:, :, :625], y.reshape(E, 2, 628)[:, :, :625]
        ok = np.allclose(x, y, rtol=0, atol=1e-9, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        assert ok, "field %s differs %s (envs %s)" % (f, tag, np.nonzero((np.asarray(x) != np.asarray(y)).reshape(E, -1).any(axis=1))[0][:8])


def _pair(E, **kw):
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    env, orc = BatchedEnv("harvest_features", E, 2, **kw), Oracle("harvest_features", E, 2, **kw)
    seeds = (np.arange(E) * 104729 + 17).astype(np.uint64)
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    return env, orc


@pytest.mark.parametrize("E,contract,horizon,T", [(203, "harvest_local", 23, 120), (64, None, 1000, 400), (1, "harvest_local", 7, 40),
                                                  (6, None, 1, 12), (1031, "harvest_local", 1000, 90)])
def test_quad_step_vs_oracle(E, contract, horizon, T):
    env, orc = _pair(E, contract=contract, horizon=horizon, auto_reset=True)
    rs = np.random.RandomState(E + horizon)
    for t in range(T):
        a = rs.randint(0, 8, size=(E, 2)).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _same(env, orc, E, "at step %d" % t)
    env.close()
    orc.close()


def test_quad_step_eats_a_lot():
    """agents that mostly walk (few turns): more apples eaten, more cells to draw for, longer spawn passes"""
    E, T = 257, 300
    env, orc = _pair(E, contract="harvest_local", horizon=1000, auto_reset=True)
    rs = np.random.RandomState(9)
    for t in range(T):
        a = rs.choice(8, size=(E, 2), p=[.22, .22, .22, .22, .03, .03, .03, .03]).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _same(env, orc, E, "at step %d" % t)
    env.close()
    orc.close()


def test_quad_step_slices_and_bad_actions():
    import torch
    E, T = 150, 60
    env, orc = _pair(E, contract="harvest_local", horizon=31, auto_reset=True)
    rs = np.random.RandomState(3)
    cuts = [0, 5, 6, 71, 150]  # slices starting off a multiple of four, a one-env slice
    for t in range(T):
        a = rs.randint(0, 8, size=(E, 2)).astype(np.uint8)
        bad = rs.rand(E) < 0.05
        a[bad, rs.randint(0, 2)] = 8 + rs.randint(0, 200)
        dev = torch.from_numpy(a).cuda()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            env.step_range_device(dev.data_ptr(), lo, hi - lo)
        torch.cuda.synchronize()
        orc.step(a)
        _same(env, orc, E, "at step %d" % t)
        assert (env.download("error_flags")[bad] & 1).all()
    env.close()
    orc.close()


@pytest.mark.parametrize("E,contract,horizon,T,per", [(203, "harvest_local", 23, 120, 16), (65, None, 1000, 300, 0), (7, "harvest_local", 5, 50, 7)])
def test_quad_fused_rollout_vs_oracle(E, contract, horizon, T, per):
    """k_feat_rollout_quad: the packed rows resident across the steps of a launch (episode ends and bad action ids inside a
    launch park the state in HBM for the one-env code and load it back) — every output plane and the final state"""
    import torch
    env, orc = _pair(E, contract=contract, horizon=horizon, auto_reset=True)
    rs = np.random.RandomState(E * 7 + T)
    a = rs.randint(0, 8, size=(T, E, 2)).astype(np.uint8)
    bad = rs.rand(T, E) < 0.01
    a[bad, 0] = 200
    acts = torch.from_numpy(a).cuda()
    traj = env.alloc_trajectory(T)
    env.rollout_fused(acts.data_ptr(), T, per, traj)
    env.synchronize()
    host = {f: traj.tensors[f].cpu().numpy() for f in traj.tensors}
    for t in range(T):
        orc.step(a[t])
        for f in ("features", "base_reward", "reward", "done", "info"):
            want = getattr(orc, f)
            took = ~bad[t]  # an env that refuses the step writes nothing into the plane
            got, want = host[f][t].reshape(want.shape)[took], want[took]
            ok = np.allclose(got, want, rtol=0, atol=1e-9) if want.dtype.kind == "f" else np.array_equal(got, want)
            assert ok, "%s plane %d (envs %s)" % (f, t, np.nonzero(took)[0][(got != want).reshape(len(got), -1).any(axis=1)][:8])
    _same(env, orc, E, "after the rollout", STATE)  # (the outputs went to the planes, not to the handle's per-step buffers)
    env.close()
    orc.close()


def test_one_env_kernels_with_the_packing_switched_off():
    """CE_FEAT_QUAD=0 (read once per process) routes HarvestFeatures n = 2 back through k_feat_step<harvest, 2> /
    k_feat_rollout<harvest, 2> — the kernels the packed ones fall back to row by row, and the A/B baseline: a child process
    steps and rolls out against the oracle with the switch off"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
from test_feat_quad_gpu import _pair, _same, STATE
E, T = 70, 90
env, orc = _pair(E, contract="harvest_local", horizon=29, auto_reset=True)
rs = np.random.RandomState(4)
for t in range(T):
    a = rs.randint(0, 8, size=(E, 2)).astype(np.uint8)
    env.step(a)
    orc.step(a)
    _same(env, orc, E, "at step %d" % t)
a = rs.randint(0, 8, size=(40, E, 2)).astype(np.uint8)
env.rollout_fused(torch.from_numpy(a).cuda().data_ptr(), 40)
for t in range(40):
    orc.step(a[t])
_same(env, orc, E, "after the rollout")
print("one-env kernels ok")
'''
    env = dict(os.environ, CE_FEAT_QUAD="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "one-env kernels ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_default_threshold_mixes_both_kernel_families():
    """without CE_FEAT_QUAD_MIN_ENVS a launch of fewer than 2 048 envs takes the one-env kernels and a larger one the packed
    ones: a child process steps one batch through slices on either side of the threshold, then as a whole, against the oracle"""
    import os
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
from test_feat_quad_gpu import _pair, _same
E, T = 2600, 25
env, orc = _pair(E, contract="harvest_local", horizon=11, auto_reset=True)
rs = np.random.RandomState(8)
for t in range(T):
    a = rs.randint(0, 8, size=(E, 2)).astype(np.uint8)
    if t % 2:
        dev = torch.from_numpy(a).cuda()
        for lo, hi in ((0, 301), (301, 2600)):  # 301 envs: one per wave; 2 299: four per wave
            env.step_range_device(dev.data_ptr(), lo, hi - lo)
        torch.cuda.synchronize()
    else:
        env.step(a)  # 2 600 envs in one launch: four per wave
    orc.step(a)
    _same(env, orc, E, "at step %d" % t)
print("both families ok")
'''
    env = {k: v for k, v in os.environ.items() if k not in ("CE_FEAT_QUAD_MIN_ENVS", "CE_FEAT_QUAD")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "both families ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _untemper(y):
    """inverse of MT19937's output tempering"""
    y ^= y >> 18
    y ^= (y << 15) & 0xefc60000
    t = y
    for _ in range(5):
        t = y ^ ((t << 7) & 0x9d2c5680)
    y = t & 0xffffffff
    t = y
    for _ in range(3):
        t = y ^ (t >> 11)
    return t & 0xffffffff


def test_quad_threshold_ties_take_the_second_word():
    """The packed kernels decide `random.random() < p` on the upper 27 bits of the 53-bit integer form and look at the second
    stream word only when those tie with the threshold's (once in 2^27 draws: never in a random test).  Here the CPython
    generator rows of engine and oracle are rewritten before every step so that the coming doubles sit exactly on, just below
    and just above the four thresholds of SPAWN_PROB (harvest_new.py:34)."""
    import math
    assert _untemper(0x12345678) != 0x12345678
    E, T = 96, 40
    env, orc = _pair(E, contract="harvest_local", horizon=1000, auto_reset=True)
    rs = np.random.RandomState(21)
    thr = [int(math.ceil(math.ldexp(p, 53))) for p in (0.0, 0.005, 0.02, 0.05)]
    walk = [.2, .2, .2, .2, .05, .05, .05, .05]
    for t in range(30):  # some apples eaten first, so that there are cells to draw for
        a = rs.choice(8, size=(E, 2), p=walk).astype(np.uint8)
        env.step(a)
        orc.step(a)
    for t in range(T):
        rng = env.download("rng").reshape(E, 2, 628).copy()
        for e in range(E):
            pos = int(rng[e, 1, 624])
            for q in range((624 - pos) // 2):  # every double left in this generation
                th = thr[rs.randint(1, 4)] if rs.rand() < 0.8 else thr[0]
                x = max(0, min((1 << 53) - 1, th + int(rs.choice([-(1 << 26), -1, 0, 1, 1 << 26, 0, -1]))))
                if rs.rand() < 0.3:
                    x = (th >> 26 << 26) | int(rs.randint(0, 1 << 26))  # same upper half, any lower half
                a_t = ((x >> 26) << 5) | int(rs.randint(0, 32))
                b_t = ((x & ((1 << 26) - 1)) << 6) | int(rs.randint(0, 64))
                rng[e, 1, pos + 2 * q] = _untemper(a_t)
                rng[e, 1, pos + 2 * q + 1] = _untemper(b_t)
        env.upload("rng", rng.reshape(E, -1))
        orc.rng.reshape(E, 2, 628)[...] = rng
        orc.import_state()
        a = rs.choice(8, size=(E, 2), p=walk).astype(np.uint8)
        env.step(a)
        orc.step(a)
        _same(env, orc, E, "at crafted step %d" % t)
    # ... and the same through the fused kernel: one crafted window per launch
    import torch
    for t in range(6):
        rng = env.download("rng").reshape(E, 2, 628).copy()
        for e in range(E):
            pos = int(rng[e, 1, 624])
            for q in range((624 - pos) // 2):
                th = thr[rs.randint(1, 4)]
                x = (th >> 26 << 26) | int(rs.randint(0, 1 << 26)) if rs.rand() < 0.5 else th + int(rs.choice([-1, 0, 1]))
                rng[e, 1, pos + 2 * q] = _untemper(((x >> 26) << 5) | int(rs.randint(0, 32)))
                rng[e, 1, pos + 2 * q + 1] = _untemper(((x & ((1 << 26) - 1)) << 6) | int(rs.randint(0, 64)))
        env.upload("rng", rng.reshape(E, -1))
        orc.rng.reshape(E, 2, 628)[...] = rng
        orc.import_state()
        a = rs.choice(8, size=(5, E, 2), p=walk).astype(np.uint8)
        env.rollout_fused(torch.from_numpy(a).cuda().data_ptr(), 5)
        for k in range(5):
            orc.step(a[k])
        _same(env, orc, E, "after crafted launch %d" % t)
    env.close()
    orc.close()
