import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv
E, n = 8192, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
env.seed(seed0=73907); env.reset()
T = 240
acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, T, acts.data_ptr())
acc = []
for t in range(T):
    env.step_device(acts.data_ptr() + t * E * n)
    if t >= 200:
        d = env.download("debug").astype(np.int64)
        fast = (d[:, 10] > d[:, 1]) & (d[:, 11] > d[:, 10]) & (d[:, 11] < d[:, 2] + 10)   # fast path taken this step
        ok = (d[:, 12] > d[:, 2]) & (d[:, 13] > d[:, 12]) & (d[:, 13] <= d[:, 3])
        acc.append([np.mean((d[:, 2] - d[:, 1])), fast.mean(), np.mean((d[fast, 10] - d[fast, 1])), np.mean((d[fast, 11] - d[fast, 10])), np.mean((d[fast, 2] - d[fast, 11])),
                    np.mean(d[:, 3] - d[:, 2]), ok.mean(), np.mean(d[ok, 12] - d[ok, 2]), np.mean(d[ok, 13] - d[ok, 12]), np.mean(d[ok, 3] - d[ok, 13])])
a = np.array(acc).mean(axis=0)
print("moves total %.0f | fast-path share %.2f: targets+pair check %.0f, m-shuffle draws %.0f, commit %.0f" % tuple(a[:5]))
print("consume+beams total %.0f | (valid %.2f) consume+mark %.0f, n-shuffle %.0f, beams %.0f" % tuple(a[5:]))
