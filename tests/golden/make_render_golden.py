#!/usr/bin/env python3
"""Golden fixtures for the render path (map_env.py:354-392,460-475 — what run_render.py:41 collects every step):
`full_map_to_colors()` = map + agents + the step's FIRE / CLEAN beams (`beam_pos`), produced by RUNNING the upstream
reference (ref_harness.py) with firing enabled and actions biased towards the beam actions.  Build-container only."""
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference  # noqa: E402


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def beam_codes(env, shape):
    m = np.zeros(shape, np.uint8)
    for r, c, ch in env.beam_pos:  # list order: a later beam overwrites an earlier one (map_env.py:371-373)
        m[r, c] = {b"F": 1, b"C": 2}[ch]
    return m


def run(R, kind, n, seed, T, horizon):
    np.random.seed(seed)
    random.seed(seed)
    env = (R.CleanupEnv if kind == "cleanup" else R.HarvestEnv)(num_agents=n, disable_firing=False, horizon=horizon)
    n_act = 9 if kind == "cleanup" else 8
    weights = np.ones(n_act)
    weights[7:] = 4.0  # CLEAN / FIRE often enough for crossing beams
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "rgb_sha", "rgb", "beam", "done", "mt")}
    env.reset()
    first = env.full_map_to_colors()
    out = {"kind": kind, "n": n, "seed": seed, "horizon": horizon, "reset_rgb": first.astype(np.uint8),
           "rgb_dtype": str(first.dtype), "render_rgb_sha": sha(env.render(mode="rgb_array").astype(np.uint8))}
    for t in range(T):
        a = ars.choice(n_act, size=n, p=weights / weights.sum())
        _, _, d, _ = env.step({"a%d" % i: int(a[i]) for i in range(n)})
        img = env.full_map_to_colors()
        assert img.min() >= 0 and img.max() <= 255
        rec["actions"].append(a.astype(np.uint8))
        rec["rgb_sha"].append(sha(img.astype(np.uint8)))
        if t < 12:
            rec["rgb"].append(img.astype(np.uint8))
        rec["beam"].append(beam_codes(env, img.shape[:2]))
        rec["done"].append(np.uint8(d["__all__"]))
        st = np.random.get_state()
        rec["mt"].append([int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)])
        if d["__all__"]:
            env.reset()
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    R = load_reference()
    jobs = {"render_cleanup_n5_firing": ("cleanup", 5, 76001, 160, 90), "render_harvest_n6_firing": ("harvest", 6, 76002, 160, 90)}
    for name, (kind, n, seed, T, horizon) in jobs.items():
        out = run(R, kind, n, seed, T, horizon)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        b = out["beam"]
        print("%-28s %7.1f KB  beam cells/step %.1f  (F %d, C %d)  steps with beams %d/%d" % (
            name, os.path.getsize(path) / 1024, (b > 0).sum() / len(b), (b == 1).sum(), (b == 2).sum(), (b.reshape(len(b), -1) > 0).any(1).sum(), len(b)))


if __name__ == "__main__":
    main()
