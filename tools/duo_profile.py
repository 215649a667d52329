#!/usr/bin/env python3
"""Diagnostic: where the two waves of k_grid_step_duo spend a step (s_memtime stamps, -DCE_DUO_STAMPS build; never shipped):
    CE_VARIANT=duostamps CE_VARIANT_FLAGS=-DCE_DUO_STAMPS python -m contracts_amd.build
    CONTRACTS_AMD_LIB=contracts_amd/csrc/libcontracts_engine_duostamps.so python tools/duo_profile.py [agents] [envs]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from contracts_amd.engine import BatchedEnv  # noqa: E402

n, E = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), (int(sys.argv[2]) if len(sys.argv) > 2 else 1365)
env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
env.seed(seed0=73907)
env.reset()
T = 400
acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, T, acts.data_ptr())
rows = []
for t in range(T):
    env.step_device(acts.data_ptr() + t * E * n)
    if t >= 300:
        d = env.download("debug").astype(np.int64)
        wp_before = None
        rows.append(d - d[:, :1])
d = np.concatenate(rows)
shuf = (d[:, 12] - d[:, 11]) > 1500  # the helper really drew
names = {1: "main: loaded", 2: "main: moves done", 3: "main: beams done (spawn entry)", 4: "main: at hand-over 1", 5: "main: past hand-over 1",
         6: "main: at hand-over 2", 7: "main: past hand-over 2", 8: "main: state stored (end)", 9: "helper: start", 10: "helper: loaded (start line)",
         11: "helper: stream walked to the shuffle", 12: "helper: draws done", 13: "helper: at hand-over 1", 14: "helper: past hand-over 2",
         15: "helper: views written (end)"}
for label, sel in (("steps whose helper drew", shuf), ("steps whose helper skipped", ~shuf)):
    print("%s: %.0f%% of env-steps" % (label, 100 * sel.mean()))
    for k in sorted(names):
        v = d[sel, k]
        print("   %-40s %8.0f cycles after the main wave's start" % (names[k], v.mean()))
