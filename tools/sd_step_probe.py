import sys, time, torch, os
sys.path.insert(0, ".")
from contracts_amd.engine import BatchedEnv
E, n, K = 32768, 4, 600
env = BatchedEnv("selfdrive", E, n, contract="selfdrive_distprop", horizon=1000, auto_reset=True)
env.seed(seed0=1); env.reset()
acts = torch.empty((K, E, n), dtype=torch.float32, device="cuda")
env.synth_actions(5, 0, K, acts.data_ptr()); env.synchronize()
for S in (1, 3):
    streams = [torch.cuda.Stream() for _ in range(S)]
    handles = [s.cuda_stream for s in streams] if S > 1 else None
    env.rollout_device(acts.data_ptr(), 100, handles); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout_device(acts.data_ptr(), K, handles); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(os.path.basename(os.environ.get("CONTRACTS_AMD_LIB", "HEAD")), "S=%d %.2f us/step" % (S, (t2 - t0) / K * 1e6), flush=True)
