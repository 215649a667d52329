#!/usr/bin/env python3
"""Throughput of the headline family against the number of env replicas on one GPU (per-step launches on 3 streams and
fused 16-step rollouts with a 16-plane trajectory ring of every output): where the chip saturates, and that batches far
beyond the BASELINE size run (HBM footprint printed; 288 GB per MI355X)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contracts_amd.engine import BatchedEnv

n = 8
for E in [int(x) for x in (sys.argv[1:] or [1024, 4096, 16384, 65536, 262144, 1048576])]:
    K = 320 if E <= 65536 else (96 if E <= 262144 else 32)
    PRE = 64
    torch.cuda.reset_peak_memory_stats()
    free0 = torch.cuda.mem_get_info()[0]
    env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
    env.seed(seed0=73907); env.reset()
    acts = torch.empty((PRE + K, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(73908, 0, PRE + K, acts.data_ptr()); env.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    H = [s.cuda_stream for s in streams]
    env.rollout_device(acts.data_ptr(), PRE, H); torch.cuda.synchronize()
    base = acts.data_ptr() + PRE * E * n
    out = {"envs": E, "steps": K}
    def timed(fn):
        best = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = max(best, E * n * K / (time.perf_counter() - t0) / 1e9)
        return round(best, 3)
    out["per_step_G"] = timed(lambda: env.rollout_device(base, K, H))
    planes = 16 if E <= 262144 else 2
    traj = env.alloc_trajectory(planes)
    out["fused16_G"] = timed(lambda: env.rollout_fused(base, K, 16, traj, H))
    out["trajectory_planes"] = planes
    out["hbm_in_use_GB"] = round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 2)
    env.check_faults()
    print(json.dumps(out), flush=True)
    del traj, acts
    env.close()
    torch.cuda.empty_cache()
