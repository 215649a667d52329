#!/usr/bin/env python3
"""Golden fixtures for JointEnv (two_stage_train.py:476-617) over the pixel envs, global and concatenated
observation modes, produced by RUNNING the upstream reference (ref_harness.py).  Build-container only."""
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


def u8(img):
    u = np.rint(np.asarray(img) * 255.0).astype(np.uint8)
    assert np.array_equal(u / 255, img)
    return u


def run(R, kind, n, seed, T, mode):
    from environments.two_stage_train import JointEnv
    np.random.seed(seed)
    random.seed(seed)
    base = (R.CleanupEnv if kind == "cleanup" else R.HarvestEnv)(num_agents=n)
    env = JointEnv(base, num_agents=n, global_obs=mode == "global", concatenated_obs=mode == "concat")
    n_act = 8 if kind == "cleanup" else 7
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "obs_sha", "obs", "rew", "done", "info_keys", "info_vals", "mt")}
    o = env.reset()
    out = {"kind": kind, "n": n, "seed": seed, "mode": mode, "reset_obs": u8(o["a0"]["image"]),
           "obs_space_shape": np.array(env.observation_space["image"].shape), "act_nvec": np.array(env.action_space.nvec)}
    for t in range(T):
        a = ars.randint(n_act, size=n)
        o, r, d, info = env.step({"a0": a})
        img = u8(o["a0"]["image"])
        rec["actions"].append(a.astype(np.uint8))
        rec["obs_sha"].append(sha(img))
        if t < 6:
            rec["obs"].append(img)
        rec["rew"].append(float(r["a0"]))
        rec["done"].append(np.uint8(d["__all__"]))
        keys = sorted(info["a0"].keys())
        rec["info_keys"].append(",".join(keys))
        rec["info_vals"].append(np.concatenate([np.atleast_1d(np.asarray(info["a0"][k], np.float64)).ravel() for k in keys]))
        st = np.random.get_state()
        rec["mt"].append([int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)])
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    R = load_reference()
    jobs = {"joint_cleanup_n3_global": ("cleanup", 3, 74001, 80, "global"), "joint_cleanup_n3_concat": ("cleanup", 3, 74002, 80, "concat"),
            "joint_harvest_n2_global": ("harvest", 2, 74003, 80, "global")}
    for name, (kind, n, seed, T, mode) in jobs.items():
        out = run(R, kind, n, seed, T, mode)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s %7.1f KB  obs %s" % (name, os.path.getsize(path) / 1024, out["reset_obs"].shape))


if __name__ == "__main__":
    main()
