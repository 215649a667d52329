#!/usr/bin/env python3
"""per-step / fused throughput against the number of env slices (HIP streams); GPU_MAX_HW_QUEUES is read from the env"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contracts_amd.engine import BatchedEnv
E, n, K, PRE = 16384, 8, 400, 300
env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
env.seed(seed0=73907); env.reset()
acts = torch.empty((PRE + K, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, PRE + K, acts.data_ptr()); env.synchronize()
streams = [torch.cuda.Stream() for _ in range(12)]
H = [s.cuda_stream for s in streams]
env.rollout_device(acts.data_ptr(), PRE, H[:3]); torch.cuda.synchronize()
state = env.state_dict()
base = acts.data_ptr() + PRE * E * n
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
def timed(fn):
    env.load_state_dict(state); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return round(E * n * K / (time.perf_counter() - t0) / 1e9, 3)
traj = env.alloc_trajectory(16)
for S in (1, 2, 3, 4, 5, 6, 8, 12):
    out["step_S%d" % S] = timed(lambda: env.rollout_device(base, K, H[:S] if S > 1 else None))
    out["fused16_S%d" % S] = timed(lambda: env.rollout_fused(base, K, 16, traj, H[:S] if S > 1 else None))
print(json.dumps(out))
