#!/usr/bin/env python3
"""per-step / fused throughput against the number of env slices (HIP streams) for any BASELINE workload:
tools/streams_sweep.py [C1|C2|C3|C4|C5 ...] (default C4); GPU_MAX_HW_QUEUES is read from the env"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contracts_amd.engine import BatchedEnv
import bench

K, PRE = 400, 300
for name in (sys.argv[1:] or ["C4"]):
    wl = bench.WORKLOADS[name]
    E, n, kind = wl["E"], wl["n"], wl["kind"]
    env = BatchedEnv(kind, E, n, contract=wl["contract"], auto_reset=True, **({} if kind == "selfdrive" else {"horizon": 1000}))
    env.seed(seed0=73907); env.reset()
    sd = kind == "selfdrive"
    acts = torch.empty((PRE + K, E, n), dtype=torch.float32 if sd else torch.uint8, device="cuda")
    env.synth_actions(73908, 0, PRE + K, acts.data_ptr()); env.synchronize()
    streams = [torch.cuda.Stream() for _ in range(8)]
    H = [s.cuda_stream for s in streams]
    env.rollout_device(acts.data_ptr(), PRE, H[:3]); torch.cuda.synchronize()
    state = env.state_dict()
    base = acts.data_ptr() + PRE * E * n * (4 if sd else 1)
    out = {"workload": name, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    def timed(fn):
        best = 0.0
        for _ in range(3):
            env.load_state_dict(state); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = max(best, E * n * K / (time.perf_counter() - t0) / 1e9)
        return round(best, 3)
    traj = env.alloc_trajectory(16)
    for S in (1, 2, 3, 4, 6, 8):
        out["step_S%d" % S] = timed(lambda: env.rollout_device(base, K, H[:S] if S > 1 else None))
        out["fused16_S%d" % S] = timed(lambda: env.rollout_fused(base, K, 16, traj, H[:S] if S > 1 else None))
    print(json.dumps(out))
    env.close()
