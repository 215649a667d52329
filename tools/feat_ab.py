"""Throughput of the two feature-vector envs (n = 2, 16 384 envs, two streams) for A/B runs of engine builds:
    CONTRACTS_AMD_LIB=path/to/lib.so python tools/feat_ab.py"""
import sys, time, os
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv
for kind, contract in (("harvest_features", "harvest_local"), ("cleanup_features", "cleanup")):
    E, n = 16384, 2
    env = BatchedEnv(kind, E, n, contract=contract, auto_reset=True)
    env.seed(seed0=73907); env.reset()
    K, W = 600, 50
    acts = torch.empty((W + K, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(73908, 0, W + K, acts.data_ptr())
    S = 2
    streams = [torch.cuda.Stream() for _ in range(S)]
    handles = [st.cuda_stream for st in streams]
    env.rollout_device(acts.data_ptr(), W, handles); torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout_device(acts.data_ptr() + W * E * n, K, handles); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(os.path.basename(os.environ.get("CONTRACTS_AMD_LIB", "default")), kind, round(E * n * K / el / 1e9, 3), "G")
    env.close()
