import sys, time
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv
for E in (32768, 262144, 1048576):
    n = 4
    env = BatchedEnv("selfdrive", E, n, contract="selfdrive_distprop", auto_reset=True)
    env.seed(seed0=73907); env.reset()
    K, W = 200, 20
    acts = torch.empty((W + K, E, n), dtype=torch.float32, device="cuda")
    env.synth_actions(73908, 0, W + K, acts.data_ptr())
    S = 2
    streams = [torch.cuda.Stream() for _ in range(S)]
    handles = [st.cuda_stream for st in streams]
    env.rollout_device(acts.data_ptr(), W, handles); torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout_device(acts.data_ptr() + W * E * n * 4, K, handles); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(E, "envs:", round(E * n * K / el / 1e9, 2), "G agent-steps/s", round(el / K * 1e6, 1), "us/step", "GB/s algorithmic", round(863 * E * K / el / 1e9))
    env.close()
