#!/usr/bin/env python3
"""PCIe-inclusive rate of the batched boundary when the caller lives on the host: every step uploads the action plane
(ce_step_host) and brings the step's results back (ce_download of obs / reward / done / info / features) for all E
envs of the headline workload.  Reported beside bench.py's HBM-resident figure, never instead of it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contracts_amd.engine import BatchedEnv

E, n, steps = 16384, 8, 60
env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
env.seed(seed0=73907)
env.reset()
rs = np.random.RandomState(0)
acts = rs.randint(8, size=(steps, E, n)).astype(np.uint8)
for fields in (("reward", "done"), ("obs", "reward", "done", "info", "features")):
    for t in range(5):
        env.step(acts[t])
        [env.download(f, raw=True) for f in fields]
    t0 = time.perf_counter()
    nbytes = 0
    for t in range(5, steps):
        env.step(acts[t])
        nbytes += sum(env.download(f, raw=True).nbytes for f in fields)
    dt = time.perf_counter() - t0
    print("host-resident caller, %-40s %6.2f ms per step = %6.1f M agent-steps/s, %5.1f MB down per step (%.1f GB/s)" % (
        "+".join(fields), dt / (steps - 5) * 1e3, (steps - 5) * E * n / dt / 1e6, nbytes / (steps - 5) / 1e6, nbytes / dt / 1e9))
env.close()
