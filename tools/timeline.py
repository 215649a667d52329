"""Diagnostic: occupancy-over-time of one step launch from the phase-stamp build (wave start/end stamps)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv

E, n = 16384, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
env.seed(seed0=73907)
env.reset()
T = 240
acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, T, acts.data_ptr())
for t in range(T):
    env.step_device(acts.data_ptr() + t * E * n)
d = env.download("debug").astype(np.int64)
s, e = d[:, 14].copy(), d[:, 15].copy()  # s_memrealtime: 100 MHz, chip-wide
t0 = s.min()
s, e = (s - t0) / 100.0, (e - t0) / 100.0  # microseconds
span = e.max()
print("launch span %.1f us; wave duration (us) mean %.0f p50 %.0f p90 %.0f max %.0f" % (span, (e - s).mean(), np.median(e - s), np.percentile(e - s, 90), (e - s).max()))
bins = np.linspace(0, span, 21)
for i in range(20):
    mid = 0.5 * (bins[i] + bins[i + 1])
    active = ((s <= mid) & (e > mid)).sum()
    started = ((s >= bins[i]) & (s < bins[i + 1])).sum()
    print("t=%3d%%  active waves %5d  started %5d" % (5 * i + 2, active, started))
shuf = d[:, 13] > d[:, 12]
print("duration shuffling %.0f vs not %.0f" % ((e - s)[shuf].mean(), (e - s)[~shuf].mean()))
