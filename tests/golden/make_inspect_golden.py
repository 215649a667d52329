#!/usr/bin/env python3
"""Golden fixtures for the read-only inspection helpers of MapEnv / CleanupEnv / HarvestEnv (map_env.py:350-374,
397-411,880-913; cleanup_new.py:370-394; Agent.py): after every step of a trace of the upstream reference
(ref_harness.py) the character map with agents and beams, the agents' poses and char ids, the visibility vectors,
the current apple / waste lists, the permitted area, the (persistently shuffled) spawn and waste lists and each
agent's colour view.  Build-container only."""
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference  # noqa: E402

ORIENT = {"UP": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3}


def pts(lst, width):
    out = np.full((width, 2), -1, np.int16)
    for k, p in enumerate(lst):
        out[k] = p
    return out


def run(R, kind, n, seed, T, horizon):
    np.random.seed(seed)
    random.seed(seed)
    env = (R.CleanupEnv if kind == "cleanup" else R.HarvestEnv)(num_agents=n, disable_firing=False, horizon=horizon)
    n_act = 9 if kind == "cleanup" else 8
    ars = np.random.RandomState(seed + 1)
    keys = ["a%d" % i for i in range(n)]
    NA = len(env.apple_points)
    rec = {k: [] for k in ("actions", "map", "poses", "char_ids", "visible", "apples", "wastes", "permitted", "spawn_points",
                           "waste_points", "views", "done")}
    env.reset()

    def snap(a, done):
        rec["actions"].append(a)
        rec["map"].append(env.get_map_with_agents().view(np.uint8).reshape(env.world_map.shape))
        rec["poses"].append(np.array([[env.agents[k].pos[0], env.agents[k].pos[1], ORIENT[env.agents[k].orientation]] for k in keys], np.int16))
        rec["char_ids"].append(np.array([env.agents[k].get_char_id()[0] for k in keys], np.uint8))
        rec["visible"].append(np.stack([env.find_visible_agents(k) for k in keys]))
        env.compute_current_apples()
        rec["apples"].append(pts(env.current_apple_points, NA))
        if kind == "cleanup":
            env.compute_current_wastes()
            rec["wastes"].append(pts(env.current_waste_points, 119))
            rec["permitted"].append(env.compute_permitted_area())
            rec["waste_points"].append(np.array(env.waste_points, np.int16))
        rec["spawn_points"].append(np.array(env.spawn_points, np.int16))
        rec["views"].append(np.stack([np.asarray(env.color_view(env.agents[k])) for k in keys]).astype(np.uint8))
        rec["done"].append(np.uint8(done))

    snap(np.zeros(n, np.uint8), False)  # right after reset
    for t in range(T):
        a = ars.randint(n_act, size=n).astype(np.uint8)
        _, _, d, _ = env.step({k: int(a[i]) for i, k in enumerate(keys)})
        snap(a, d["__all__"])
        if d["__all__"]:
            env.reset()
    out = {"kind": kind, "n": n, "seed": seed, "horizon": horizon, "apple_points": np.array(env.apple_points, np.int16)}
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    R = load_reference()
    jobs = {"inspect_cleanup_n4": ("cleanup", 4, 76101, 70, 50), "inspect_harvest_n5": ("harvest", 5, 76102, 70, 50)}
    for name, (kind, n, seed, T, horizon) in jobs.items():
        out = run(R, kind, n, seed, T, horizon)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-24s %7.1f KB  steps %d" % (name, os.path.getsize(path) / 1024, len(out["actions"]) - 1))


if __name__ == "__main__":
    main()
