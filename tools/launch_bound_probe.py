"""Are the small-batch configs host-launch-bound?  Steps/s and the host's enqueue time per step for 1-3 slices on 1-3 streams
(selfdrive n=4 x 32768 envs, cleanup n=4 x 4096 envs).  Answer on MI355X: no — enqueue is 2.7 us per launch, a step takes
16 us; those batches are bound by the latency of one wave's program."""
import sys, time
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv
for kind, E, n, contract, dt in (("selfdrive", 32768, 4, "selfdrive_distprop", torch.float32), ("cleanup", 4096, 4, "cleanup", torch.uint8)):
    for S in (1, 2, 3):
        env = BatchedEnv(kind, E, n, contract=contract, auto_reset=True)
        env.seed(seed0=73907); env.reset()
        K, W = 400, 20
        acts = torch.empty((W + K, E, n), dtype=dt, device="cuda")
        env.synth_actions(73908, 0, W + K, acts.data_ptr())
        streams = [torch.cuda.Stream() for _ in range(S)]
        handles = None if S == 1 else [st.cuda_stream for st in streams]
        env.rollout_device(acts.data_ptr(), W, handles); torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.rollout_device(acts.data_ptr() + W * E * n * acts.element_size(), K, handles)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(kind, E, "S", S, round(E * n * K / el / 1e9, 2), "G", round(el / K * 1e6, 1), "us/step; host enqueue", round(t_host / K * 1e6, 1), "us/step")
        env.close()
