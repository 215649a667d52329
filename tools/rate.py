#!/usr/bin/env python3
"""One throughput number per line for A/B runs of engine builds (interleave builds with tools/ab.sh):

    CONTRACTS_AMD_LIB=lib.so python tools/rate.py C4 C2:fused C1 cleanup_features,2,16384 ...

RATE_PREROLL=N adds N untimed steps first; RATE_STREAMS=S cuts the batch into S slices on S streams (default 3).  A spec is a BASELINE config key (C1..C5, bench.py's WORKLOADS) or kind,agents,envs, optionally :fused and / or
@counter (the counter-RNG mode).  Protocol: 300-step
pre-roll, then the median of 5 repeats of 300 steps (fused: 304 = 19 launches of 16), three env slices on three streams."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from contracts_amd.engine import BatchedEnv  # noqa: E402

CONTRACT = {"cleanup": "cleanup", "harvest": "harvest_local", "selfdrive": "selfdrive_distprop", "harvest_features": "harvest_local",
            "cleanup_features": "cleanup"}
tag = os.path.basename(os.environ.get("CONTRACTS_AMD_LIB", "default")).replace("libcontracts_engine", "").replace(".so", "") or "HEAD"
for spec in sys.argv[1:]:
    name, _, mode = spec.partition(":")
    mode, _, rng = mode.partition("@")  # C4@counter, C4:fused@counter: the counter-RNG mode (grid kinds)
    if "@" in name:
        name, _, rng = name.partition("@")
    if name in bench.WORKLOADS:
        w = bench.WORKLOADS[name]
        kind, n, E = w["kind"], w["n"], w["E"]
    else:
        kind, n, E = name.split(",")
        n, E = int(n), int(E)
    fused = mode == "fused"
    K, T, PRE, S = (304, 16, 300, 3) if fused else (300, 0, 300, 3)
    S = int(os.environ.get("RATE_STREAMS", S))  # env slices / streams (default 3, bench.py's)
    LONG = int(os.environ.get("RATE_PREROLL", "0"))  # extra untimed steps first (replaying the pre-roll planes): steady states

    env = BatchedEnv(kind, E, n, contract=CONTRACT[kind], horizon=1000, auto_reset=True, rng=rng or "mt19937")
    env.seed(seed0=73907)
    env.reset()
    dt = torch.float32 if kind == "selfdrive" else torch.uint8
    acts = torch.empty((PRE + K, E, n), dtype=dt, device="cuda")
    env.synth_actions(73908, 0, PRE + K, acts.data_ptr())
    env.synchronize()
    plane = E * n * acts.element_size()
    streams = [torch.cuda.Stream() for _ in range(S)]
    handles = [s.cuda_stream for s in streams]
    for _ in range(LONG // PRE):
        env.rollout_device(acts.data_ptr(), PRE, handles)
    env.rollout_device(acts.data_ptr(), PRE, handles)
    torch.cuda.synchronize()
    traj = env.alloc_trajectory(T) if fused else None
    el = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fused:
            env.rollout_fused(acts.data_ptr() + PRE * plane, K, T, traj, handles)
        else:
            env.rollout_device(acts.data_ptr() + PRE * plane, K, handles)
        torch.cuda.synchronize()
        el.append(time.perf_counter() - t0)
    env.check_faults()
    med = statistics.median(el[1:])
    print("%-8s %-34s %8.3f G agent-steps/s  %7.2f us/step" % (tag, spec, E * n * K / med / 1e9, med / K * 1e6), flush=True)
    env.close()
    del acts, traj
