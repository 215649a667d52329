"""Throughput of every BASELINE.json config on one GPU (2 env slices on 2 streams, like bench.py) (not the headline bench; results go to
profiles/<round>_configs.json for DESIGN.md).  Same timing method as bench.py."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv

CONFIGS = [
    # BASELINE config 0 (the reference's CPU plumbing case) batched: algorithmic bytes = list stamps + agents + accumulators
    # read and written once, int16 feature rows, rewards, infos (same accounting as SURVEY §8d)
    ("C1 harvest (HarvestFeatures) n=2 + HarvestFeaturemodLocalContract, 16384 envs", "harvest_features", 2, 16384, "harvest_local", 887),
    ("C1b cleanup (CleanupFeatures) n=2 + CleanupContract, 16384 envs", "cleanup_features", 2, 16384, "cleanup", 1155),
    ("C2 cleanup_new n=4 + CleanupContract, 4096 envs", "cleanup", 4, 4096, "cleanup", 4211),
    ("C3 harvest_new n=8 + HarvestFeaturemodLocalContract, 16384 envs", "harvest", 8, 16384, "harvest_local", 7313),
    ("C4 cleanup_new n=8 + CleanupContract, 16384 envs (headline)", "cleanup", 8, 16384, "cleanup", 7235),
    ("C4x cleanup_new n=8 + CleanupContract, 65536 envs", "cleanup", 8, 65536, "cleanup", 7235),
    ("C5 selfdrive n=4 + SelfdriveContractDistprop, 32768 envs", "selfdrive", 4, 32768, "selfdrive_distprop", 863),
]
out = []
for name, kind, n, E, contract, algo in CONFIGS:
    env = BatchedEnv(kind, E, n, contract=contract, auto_reset=True)
    env.seed(seed0=73907)
    env.reset()
    K, W = 400, 50
    dt = torch.float32 if kind == "selfdrive" else torch.uint8
    acts = torch.empty((W + K, E, n), dtype=dt, device="cuda")
    env.synth_actions(73908, 0, W + K, acts.data_ptr())
    stride = E * n * acts.element_size()
    S = 2
    streams = [torch.cuda.Stream() for _ in range(S)]
    handles = [st.cuda_stream for st in streams]
    env.rollout_device(acts.data_ptr(), W, handles)
    torch.cuda.synchronize()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
    for e0, st in zip(ev0, streams):
        e0.record(st)
    t0 = time.perf_counter()
    env.rollout_device(acts.data_ptr() + W * stride, K, handles)
    for e1, st in zip(ev1, streams):
        e1.record(st)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / S / K  # mean duration of one launch (E/S envs); S run concurrently
    env.check_faults() if kind != "selfdrive" else None
    row = {"config": name, "agent_steps_per_s": E * n * K / el, "streams": S, "envs_per_launch": E // S, "kernel_ms": ms,
           "algorithmic_GBs": algo * E / (ms * 1e-3) / 1e9, "roofline_frac": algo * E / (ms * 1e-3) / 1e9 / 8000.0}
    print(json.dumps(row))
    out.append(row)
    env.close()
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json", "w"), indent=1)
