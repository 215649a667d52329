"""Host-side format conversion rate: ce_obs_u8_to_f64 (uint8 pitched views -> the reference's float64 images) by thread count.

  python tools/host_convert_rate.py [--envs 4096] [--agents 8]

CE_HOST_SSE_ONLY=1 selects the 16-byte streaming-store form for an A/B against the 32-byte one.  No GPU needed.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contracts_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    L = _lib.load()
    E, n = a.envs, a.agents
    rng = np.random.RandomState(0)
    pitched = rng.randint(0, 256, size=(E, n * 720), dtype=np.uint8)
    out = np.empty((E, n, 15, 15, 3))
    want = None
    for thr in (1, 2, 4, 8, 16, 32):
        if thr > (os.cpu_count() or 1):
            break
        L.ce_obs_u8_to_f64(pitched.ctypes.data, out.ctypes.data, E, n, n * 720, 720, 48, thr)
        if want is None:
            want = pitched.reshape(E, n, 15, 48)[..., :45].reshape(E, n, 15, 15, 3) / 255.0
        assert np.array_equal(out, want)
        best = 1e9
        for _ in range(a.reps):
            t = time.perf_counter()
            L.ce_obs_u8_to_f64(pitched.ctypes.data, out.ctypes.data, E, n, n * 720, 720, 48, thr)
            best = min(best, time.perf_counter() - t)
        print("threads %2d: %.2f ms  %.1f GB/s written" % (thr, best * 1e3, out.nbytes / best / 1e9))


if __name__ == "__main__":
    main()
