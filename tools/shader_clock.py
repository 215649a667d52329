"""Diagnostic: the shader clock the step kernel's waves actually run at, per batch size — s_memtime cycles of a wave's life
(phase stamps 0 and 9) over its s_memrealtime ticks (stamps 14 and 15: a constant 100 MHz counter).  Needs the stamps build:
    CE_PHASE_STAMPS=1 python -m contracts_amd.build && CONTRACTS_AMD_LIB=contracts_amd/csrc/libcontracts_engine_stamps.so python tools/shader_clock.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from contracts_amd.engine import BatchedEnv

T = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for E, n in ((1024, 8), (4096, 4), (4096, 8), (8192, 8), (16384, 8), (32768, 8), (65536, 8)):
    env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
    env.seed(seed0=73907)
    env.reset()
    P = min(T, 200)  # action planes, replayed
    acts = torch.empty((P, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(73908, 0, P, acts.data_ptr())
    streams = [torch.cuda.Stream() for _ in range(3)]
    for _ in range(max(1, T // P)):
        env.rollout_device(acts.data_ptr(), P, [s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    d = env.download("debug").astype(np.int64)
    cyc, real = d[:, 9] - d[:, 0], d[:, 15] - d[:, 14]
    ok = (real > 0) & (cyc > 0)
    ghz = cyc[ok] / (real[ok] * 10.0)  # cycles per ns
    print("%6d envs x %d agents: wave life %7.0f cycles = %6.2f us, shader clock median %.2f GHz (p10 %.2f, p90 %.2f)"
          % (E, n, np.median(cyc[ok]), np.median(real[ok]) * 0.01, np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90)), flush=True)
    env.close()
