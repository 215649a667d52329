"""The joint baseline's RLlib-facing tick (BatchedJointBaseEnv: send_actions of {env: {'a0': row}} + poll + a walk over every
env's obs / reward / done / info) at E envs, split into its parts.  python tools/joint_dict_rate.py [E] [kind] [n]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from contracts_amd.vector_env import BatchedJointBaseEnv  # noqa: E402

import gc  # noqa: E402

GC_LOG, _t = [], [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        GC_LOG.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3))


gc.callbacks.append(_gc_cb)
args = [x for x in sys.argv[1:] if x != "-v"]
E = int(args[0]) if len(args) > 0 else 16384
kind = args[1] if len(args) > 1 else "cleanup"
n = int(args[2]) if len(args) > 2 else 8
out = {"envs": E, "kind": kind, "agents": n}
for mode in ("global", "concatenated"):
    venv = BatchedJointBaseEnv(kind, E, n, mode=mode, seed0=73907, horizon=1000)
    na = venv.engine.num_actions
    rs = np.random.RandomState(5)
    planes = [rs.randint(na, size=(E, n)).astype(np.uint8) for _ in range(4)]
    acts = [{e: {"a0": p[e]} for e in range(E)} for p in planes]
    venv.poll()
    venv.send_actions(acts[0])
    venv.poll()
    parts = {"send_actions_ms": [], "poll_ms": [], "walk_ms": []}
    del GC_LOG[:]
    for k in range(12):
        t0 = time.perf_counter()
        venv.send_actions(acts[k % 4])
        t1 = time.perf_counter()
        obs, rew, dones, infos, _ = venv.poll()
        t2 = time.perf_counter()
        s = 0.0
        for (e, o), r, d, i in zip(obs.items(), rew.values(), dones.values(), infos.values()):
            s += o["a0"]["image"][0, 0, 0]
        t3 = time.perf_counter()
        parts["send_actions_ms"].append((t1 - t0) * 1e3)
        parts["poll_ms"].append((t2 - t1) * 1e3)
        parts["walk_ms"].append((t3 - t2) * 1e3)
    if "-v" in sys.argv:
        print(mode, {k: [round(x, 1) for x in v] for k, v in parts.items()}, "full collections (ms):", [round(ms, 1) for g, ms in GC_LOG if g == 2], file=sys.stderr)
    del GC_LOG[:]
    row = {k: round(float(np.median(v)), 3) for k, v in parts.items()}
    row["tick_ms"] = round(sum(row.values()), 3)
    row["agent_steps_per_s"] = round(E * n / (row["tick_ms"] * 1e-3))
    out[mode] = row
    venv.stop()
print(json.dumps(out))
