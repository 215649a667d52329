"""Diagnostic: where a wave of k_feat_step_quad (HarvestFeatures n = 2, four envs per wave) spends its life — s_memtime stamps
at the phase boundaries of an instrumented build, waits for outstanding memory included in the phase that issued them:

    CE_PHASE_STAMPS=1 python -m contracts_amd.build
    CONTRACTS_AMD_LIB=contracts_amd/csrc/libcontracts_engine_stamps.so python tools/quad_profile.py [envs] [fused]

`fused`: the stamps of the last step of 16-step launches of k_feat_rollout_quad instead (the phases of the step body only).

Never used for reported numbers."""
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402

from contracts_amd.engine import BatchedEnv  # noqa: E402

E, n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 2
FUSED = "fused" in sys.argv[2:]
env = BatchedEnv("harvest_features", E, n, contract="harvest_local", auto_reset=True)
env.seed(seed0=73907)
env.reset()
T = 360
acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, T, acts.data_ptr())
names = ["loads", "map (copy + paint)", "moves + consume", "eligible + scan", "twist", "spawn", "closest + close counts",
         "rewards / transfers / metrics", "stores", "one-env rows"]
rows, life = [], []
if FUSED:
    for t0 in range(0, T - 15, 16):
        env.rollout_fused(acts.data_ptr() + t0 * E * n, 16)
        if t0 >= 240:
            d = env.download("debug").astype(np.int64)[::4]
            d = d[:, [11, 12, 2, 3, 4, 5, 6, 7, 8, 13]]  # loop top, window requested, the body's phases, outputs stored
            rows.append(np.diff(d, axis=1))
            life.append(d[:, -1] - d[:, 0])
    names = ["window + near-end key requests", "wait for the window (stamp artefact)"] + names[2:8] + ["outputs"]
else:
    for t in range(T):
        env.step_device(acts.data_ptr() + t * E * n)
        if t >= 300:
            d = env.download("debug").astype(np.int64)[::4, :11]
            rows.append(np.diff(d, axis=1))
            life.append(d[:, 10] - d[:, 0])
rows, life = np.concatenate(rows), np.concatenate(life)
print("s_memtime ticks (shader clock) per wave, %d waves x %d samples:" % (E // 4, len(rows) // (E // 4)))
for k, name in enumerate(names):
    c = rows[:, k]
    print("  %-32s mean %7.1f   p50 %6.0f  p99 %6.0f  max %6.0f" % (name, c.mean(), np.percentile(c, 50), np.percentile(c, 99), c.max()))
print("  wave life: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f ticks" % (
    life.mean(), np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99), life.max()))
