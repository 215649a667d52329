"""Where a dict-protocol tick's observation path spends its time: the D2H copy of the pitched uint8 block, the float64
conversion, a chunked pipeline of both driven from Python, and the one-call form the vector env uses (ce_download_obs_f64),
each timed alone on an idle host.

  python tools/obs_path_probe.py [--envs 16384] [--agents 8] [--kind cleanup]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contracts_amd.engine import BatchedEnv  # noqa: E402


def best_of(fn, reps=8):
    b = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t)
    return b * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--kind", default="cleanup")
    a = ap.parse_args()
    E, n = a.envs, a.agents
    eng = BatchedEnv(a.kind, E, n)
    eng.seed(seed0=1)
    eng.reset()
    b = eng.b
    u8 = eng.host_alloc((E, b.obs_env_stride), np.uint8)
    f64 = np.empty((E, n, 15, 15, 3))
    f64[:] = 0
    eng.synchronize()

    def dma(lo=0, hi=E):
        eng.download_async("obs", u8[lo:hi], lo, hi - lo)
        eng.synchronize()

    print("bytes: u8 %.1f MB, f64 %.1f MB" % (u8.nbytes / 1e6, f64.nbytes / 1e6))
    t = best_of(dma)
    print("D2H whole block:        %.2f ms  (%.1f GB/s)" % (t, u8.nbytes / t / 1e6))
    t = best_of(lambda: dma(0, E // 4))
    print("D2H one quarter:        %.2f ms  (%.1f GB/s)" % (t, u8.nbytes / 4 / t / 1e6))
    for T in (8, 16, 32, 64):
        t = best_of(lambda: eng.obs_u8_to_f64(u8, f64, T))
        print("convert, %2d threads:    %.2f ms  (%.1f GB/s written)" % (T, t, f64.nbytes / t / 1e6))
    for parts in (1, 2, 4, 8):
        cuts = [E * i // parts for i in range(parts + 1)]
        for T in (16, 32):
            def pipe():
                eng.download_async("obs", u8[cuts[0]:cuts[1]], cuts[0], cuts[1] - cuts[0])
                for i in range(parts):
                    eng.synchronize()
                    if i + 1 < parts:
                        eng.download_async("obs", u8[cuts[i + 1]:cuts[i + 2]], cuts[i + 1], cuts[i + 2] - cuts[i + 1])
                    eng.obs_u8_to_f64(u8[cuts[i]:cuts[i + 1]], f64[cuts[i]:cuts[i + 1]], T)
            print("pipeline, %d parts, %2d threads: %.2f ms" % (parts, T, best_of(pipe)))
    one_call(eng, u8, f64)


def one_call(eng, u8, f64):
    want = f64.copy()
    for parts in (1, 2, 4, 8, 16):
        for T in (16, 32):
            f64[:] = 0
            t = best_of(lambda: eng.download_obs_f64(u8, f64, T, parts=parts))
            assert np.array_equal(f64, want)
            print("ce_download_obs_f64, %2d parts, %2d threads: %.2f ms" % (parts, T, t))


if __name__ == "__main__":
    main()
