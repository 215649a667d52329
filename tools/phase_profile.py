"""Diagnostic: per-phase cycle shares of k_grid_step from an instrumented build
(CE_PHASE_STAMPS=1 python -m contracts_amd.build --force).  Never used for reported numbers."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv

E, n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
env.seed(seed0=73907)
env.reset()
T = 260
acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, T, acts.data_ptr())
names = ["load", "moves", "consume+beams", "spawn", "rewards+metrics", "features", "contract+done", "store", "obs"]
acc = np.zeros(9)
cnt = 0
for t in range(T):
    env.step_device(acts.data_ptr() + t * E * n)
    if t >= 200:
        d = env.download("debug").astype(np.int64)
        acc += np.diff(d[:, :10], axis=1).mean(axis=0)
        tot = (d[:, 9] - d[:, 0])
        cnt += 1
acc /= cnt
print("mean cycles per wave per phase (s_memtime ticks):")
for k, v in zip(names, acc):
    print("  %-18s %9.0f  %5.1f%%" % (k, v, 100 * v / acc.sum()))
sub = env.download("debug").astype(np.int64)
d13 = sub[:, 13] - sub[:, 12]
sh = (d13 > 0) & (d13 < 500000) & (sub[:, 10] > sub[:, 12]) & (sub[:, 10] < sub[:, 13]) & (sub[:, 4] > sub[:, 13]) & (sub[:, 4] - sub[:, 13] < 500000)
print("  spawn breakdown (last step): bulk %.0f | apple scan %.0f | %d%% of envs shuffle: draws %.0f, apply %.0f, waste pick %.0f" % (
    (sub[:, 11] - sub[:, 3]).mean(), (sub[:, 12] - sub[:, 11]).mean(), 100 * sh.mean(),
    (sub[sh, 10] - sub[sh, 12]).mean() if sh.any() else 0, (sub[sh, 13] - sub[sh, 10]).mean() if sh.any() else 0,
    (sub[sh, 4] - sub[sh, 13]).mean() if sh.any() else 0))
print("  total %.0f ; p50/p90/max of last step %s" % (acc.sum(), np.percentile(tot, [50, 90, 100])))
