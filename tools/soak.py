"""Randomised soak: engine vs CPU oracle over random configurations (family, n, flags, horizon, policy mix),
every persistent and output field compared after every step — or, for half of the configurations, after every fused
multi-step launch (ce_rollout_fused, chunks of 1..9 steps).  Usage: python tools/soak.py [seconds] [seed] [counter|quad]
(`counter`: the grid kinds run in the counter-RNG mode, engine and oracle alike; `quad`: HarvestFeatures with two agents only —
the four-envs-per-wave kernels — with batch sizes off multiples of four, longer runs and walk-heavy policies)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 3 and sys.argv[3] == "quad":  # the packed kernels at every batch size (the library reads this once)
    os.environ.setdefault("CE_FEAT_QUAD_MIN_ENVS", "1")
from contracts_amd.engine import BatchedEnv
from oracle.pyoracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
COUNTER = len(sys.argv) > 3 and sys.argv[3] == "counter"
QUAD = len(sys.argv) > 3 and sys.argv[3] == "quad"
FIELDS = ["grid", "agents", "spawn_perm", "rng", "timestep", "theta", "obs", "base_reward", "reward", "done", "info", "features",
          "int_metrics", "f64_metrics", "final_int_metrics", "final_f64_metrics"]
t_end = time.time() + budget
runs = steps_total = 0
while time.time() < t_end:
    kind = rs.choice(["cleanup", "harvest"] if COUNTER else ["cleanup", "harvest", "cleanup", "harvest", "cleanup_features", "harvest_features"])
    if QUAD:
        kind = "harvest_features"
    feat = kind.endswith("_features")
    n = 2 if QUAD else int(rs.randint(2 if feat else 1, 10))
    firing = bool(rs.randint(2)) and not feat
    contract = None if rs.rand() < 0.3 else ("cleanup" if kind.startswith("cleanup") else "harvest_local")
    inequity = n > 1 and rs.rand() < 0.15 and not feat
    collective = (not inequity) and rs.rand() < 0.15 and not feat
    horizon = int(rs.choice([7, 23, 60, 1000]))
    E = int(rs.choice([1, 3, 66, 131, 517, 1024]) if QUAD else rs.choice([65, 128, 300]))
    kw = dict(contract=contract, firing=firing, horizon=horizon, auto_reset=True, collective=collective, inequity=inequity,
              alpha=float(rs.rand() * 5), beta=float(rs.rand()))
    if COUNTER:
        kw["rng"] = "counter"
    trace = (not feat) and rs.rand() < 0.5
    if trace:
        kw["beam_trace"] = True
    if not feat and rs.rand() < 0.3:  # a hand-made layout (ce_config.ascii_map): the CM kernel instances
        sys.path.insert(0, "tests/golden")
        from make_golden import CLEANUP_MID, CLEANUP_SMALL, HARVEST_SMALL
        rows = HARVEST_SMALL if kind == "harvest" else (CLEANUP_SMALL if rs.rand() < 0.5 else CLEANUP_MID)
        cap = sum(r.count("P") for r in rows)
        n = min(n, cap)
        if inequity and n < 2:
            kw["inequity"] = inequity = False
        kw["ascii_map"] = rows
    env, orc = BatchedEnv(kind, E, n, **kw), Oracle(kind, E, n, **kw)
    seeds = rs.randint(0, 2 ** 31 - 1, size=E).astype(np.uint64)
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    na = env.num_actions + (1 if feat else 0)  # the feature envs' code paths accept one more (effect-free) action
    p = rs.dirichlet(np.ones(na) * rs.choice([0.3, 1.0, 5.0]))
    fields = FIELDS + (["waste_perm"] if kind == "cleanup" else []) + (["beam_map"] if trace else [])
    if feat:
        fields = [f for f in FIELDS if f not in ("spawn_perm", "obs")]
    T = int(rs.choice([120, 400, 1100]) if QUAD else rs.choice([40, 120]))
    if QUAD and rs.rand() < 0.5:  # mostly walking: more apples eaten, longer draw lists, more generation ends
        p = rs.dirichlet(np.array([4.0, 4.0, 4.0, 4.0, 0.5, 0.5, 0.5, 0.5][:na]))
    ok = True
    fused = rs.rand() < 0.5
    t = 0
    while t < T:
        if fused:  # a chunk of steps in one launch on the engine, step by step on the oracle
            import torch
            c = int(min(T - t, rs.randint(1, 10)))
            a = rs.choice(na, size=(c, E, n), p=p).astype(np.uint8)
            if a.max() >= env.num_actions:  # (the effect-free extra action of the feature envs is a per-step code path)
                a = np.minimum(a, env.num_actions - 1)
            dev = torch.from_numpy(a).cuda()
            env.rollout_fused(dev.data_ptr(), c, int(rs.choice([0, 2, 4])))
            env.synchronize()
            for k in range(c):
                orc.step(a[k])
            t += c
        else:
            a = rs.choice(na, size=(E, n), p=p).astype(np.uint8)
            env.step(a)
            orc.step(a)
            t += 1
        for f in fields:
            x, y = (env.download(f, raw=True) if feat and f == "grid" else env.download(f)), getattr(orc, f)
            if f == "rng" and not COUNTER:
                x, y = x.reshape(E, -1, 628)[:, :, :625], y.reshape(E, -1, 628)[:, :, :625]
            same = np.allclose(x, y, rtol=0, atol=1e-9, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
            if not same:
                bad = np.nonzero((x != y).reshape(E, -1).any(axis=1))[0]
                print("MISMATCH", kind, n, kw, "fused" if fused else "per-step", "field", f, "step", t, "envs", bad[:6])
                ok = False
                break
        if not ok:
            break
    runs += 1
    steps_total += T * E
    env.close()
    orc.close()
    if not ok:
        sys.exit(1)
print("soak ok: %d random configs, %d env-steps compared field by field" % (runs, steps_total))
