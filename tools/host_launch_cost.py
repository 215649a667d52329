import sys, time, torch
sys.path.insert(0, ".")
from contracts_amd.engine import BatchedEnv
for kind, n, E, contract in (("selfdrive", 4, 32768, "selfdrive_distprop"), ("cleanup", 4, 4096, "cleanup"), ("harvest_features", 2, 16384, "harvest_local"), ("cleanup", 8, 16384, "cleanup")):
    env = BatchedEnv(kind, E, n, contract=contract, horizon=1000, auto_reset=True)
    env.seed(seed0=1); env.reset()
    K = 600
    acts = torch.empty((K, E, n), dtype=torch.float32 if kind == "selfdrive" else torch.uint8, device="cuda")
    env.synth_actions(5, 0, K, acts.data_ptr()); env.synchronize()
    for S in (1, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        handles = [s.cuda_stream for s in streams] if S > 1 else None
        env.rollout_device(acts.data_ptr(), 100, handles); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.rollout_device(acts.data_ptr(), K, handles); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-17s S=%d  host enqueue %.2f us/step  total %.2f us/step" % (kind, S, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
    env.close()
