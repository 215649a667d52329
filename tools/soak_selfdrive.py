"""Randomised soak of the selfdrive kernels vs the CPU oracle: random n, collision rule, null_prob, batch size and policy,
per-step launches and fused rollouts, and — what the fixed tests cannot enumerate — both MT19937 streams parked at random
positions near (and at) the end of a generation before every run, so that resets straddle generation ends in every split
(the wave-cooperative regeneration of sd_reset_group).  Usage: python tools/soak_selfdrive.py [seconds] [seed]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FLOATS = ("obs_f64", "reward", "sd_state", "theta", "sd_info", "f64_metrics")
INTS = ("done", "done_agents", "info", "base_reward")
KEEP = np.r_[0:625, 628:1253]
t_end = time.time() + budget
runs = steps_total = 0
while time.time() < t_end:
    n = int(rs.randint(1, 11))
    E = int(rs.choice([37, 200, 1030]))
    kw = dict(contract=None if rs.rand() < 0.2 else "selfdrive_distprop", auto_reset=True, collision_on=bool(rs.randint(2)),
              null_prob=float(rs.choice([0.0, 0.3, 0.7])))
    env, orc = BatchedEnv("selfdrive", E, n, **kw), Oracle("selfdrive", E, n, **kw)
    seeds = rs.randint(0, 2 ** 31 - 1, size=E).astype(np.uint64)
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    rng = env.download("rng").copy()
    rng[:, 628 + 624] = 624 - rs.randint(0, 2 * n + 6, size=E)  # CPython stream: a reset draws 2n words
    rng[:, 624] = 624 - rs.randint(0, 8, size=E)                # numpy stream: 2 or 4
    env.upload("rng", rng)
    orc.rng[...] = rng
    orc.import_state()
    if rs.rand() < 0.5:
        for o in (env, orc):
            o.reset()
    T = int(rs.choice([60, 200]))
    lo, hi = (0.0, 0.1) if rs.rand() < 0.5 else (-0.15, 0.15)
    fused = rs.rand() < 0.5
    t, ok = 0, True
    while t < T and ok:
        c = int(min(T - t, rs.randint(1, 40))) if fused else 1
        a = rs.uniform(lo, hi, size=(c, E, n)).astype(np.float32)
        if fused:
            dev = torch.from_numpy(a).cuda()
            env.rollout_fused(dev.data_ptr(), c, int(rs.choice([0, 3, 16])))
            env.synchronize()
        else:
            env.step(a[0])
        for k in range(c):
            orc.step(a[k])
        t += c
        for f in FLOATS + INTS + ("rng",):
            x, y = env.download(f), getattr(orc, f)
            if f == "rng":
                x, y = x[:, KEEP], y[:, KEEP]
            same = np.allclose(x, y, rtol=0, atol=1e-9, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
            if not same:
                bad = np.nonzero((x != y).reshape(E, -1).any(axis=1))[0]
                print("MISMATCH selfdrive n=%d" % n, kw, "fused" if fused else "per-step", "field", f, "step", t, "envs", bad[:6])
                ok = False
                break
    runs += 1
    steps_total += T * E
    env.close()
    orc.close()
    if not ok:
        sys.exit(1)
print("selfdrive soak ok: %d random configs, %d env-steps compared field by field" % (runs, steps_total))
