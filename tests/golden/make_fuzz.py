#!/usr/bin/env python3
"""Known-answer fuzz fixtures: ONE reference step from hand-crafted (injected) states.

Random rollouts rarely produce the dense agent clusters where `MapEnv.update_moves` gets
interesting (contested cells, swaps, chains, cycles, two agents sharing a cell — SURVEY.md §8c
KA1-KA14).  Here agents are dropped into a small box (sometimes two on one cell), the map is
filled with random apples / waste, beams are enabled, and the reference performs one step with a
fresh `np.random.seed(seed)`.  Inputs and the reference's outputs are stored; the CPU oracle and the
HIP engine are both checked against them (tests/test_fuzz_states.py).

Build-container only (imports the reference through ref_harness.py).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CHAR2CODE, ORIENT2INT, grid_codes, obs_u8, perm_of  # noqa: E402
from ref_harness import load_reference  # noqa: E402

CODE2CHAR = {v: k for k, v in CHAR2CODE.items()}
INT2ORIENT = {v: k for k, v in ORIENT2INT.items()}


def inject(env, kind, grid, agents):
    """overwrite the reference env's map + agents with the crafted state"""
    H, W = grid.shape
    for r in range(H):
        for c in range(W):
            env.single_update_map(r, c, CODE2CHAR[int(grid[r, c])])
    for i, ag in enumerate(env.agents.values()):
        ag.set_pos(np.array([int(agents[i, 0]), int(agents[i, 1])]))
        ag.set_orientation(INT2ORIENT[int(agents[i, 2])])
        ag.reward_this_turn = 0
        if kind == "cleanup":
            ag.cleaned_squares = 0
    env.compute_current_apples()
    if kind == "cleanup":
        env.compute_current_wastes()
    # colour map: agents are painted at the end of every step (map_env.py:257-261); emulate that
    for ag in env.agents.values():
        env.single_update_world_color_map(ag.pos[0], ag.pos[1], ag.get_char_id())


def craft(rs, kind, env, n):
    base = env.base_map
    H, W = base.shape
    grid = np.zeros((H, W), np.uint8)
    for r in range(H):
        for c in range(W):
            ch = base[r, c]
            if ch == b"@":
                grid[r, c] = 1
            elif kind == "cleanup":
                if ch == b"B" and rs.rand() < 0.35:
                    grid[r, c] = 2
                elif ch in (b"H", b"R"):
                    grid[r, c] = 3 if rs.rand() < rs.choice([0.15, 0.35, 0.6]) else 4
                elif ch == b"S":
                    grid[r, c] = 5
            else:
                if ch == b"A" and rs.rand() < rs.choice([0.1, 0.5, 0.9]):
                    grid[r, c] = 2
    # agents in a small box so that they interact
    bh, bw = rs.randint(2, 5), rs.randint(2, 5)
    while True:
        r0, c0 = rs.randint(1, H - 1 - bh + 1), rs.randint(1, W - 1 - bw + 1)
        cells = [(r, c) for r in range(r0, r0 + bh) for c in range(c0, c0 + bw) if grid[r, c] != 1]
        if len(cells) >= max(2, n // 2):
            break
    agents = np.zeros((n, 3), np.int64)
    replace = len(cells) < n or rs.rand() < 0.25  # sometimes two agents share a cell (KA8 aftermath)
    idx = rs.choice(len(cells), size=n, replace=replace)
    for i in range(n):
        agents[i, :2] = cells[idx[i]]
        agents[i, 2] = rs.randint(4)
        if grid[agents[i, 0], agents[i, 1]] == 2:
            grid[agents[i, 0], agents[i, 1]] = 0  # an agent never stands on an apple between steps
    return grid, agents


def run(kind, n, count, seed0, firing):
    R = load_reference()
    rs = np.random.RandomState(seed0)
    np.random.seed(seed0)
    env = (R.CleanupEnv if kind == "cleanup" else R.HarvestEnv)(num_agents=n, disable_firing=not firing)
    env.reset()
    n_act = (9 if firing else 8) if kind == "cleanup" else (8 if firing else 7)
    static_waste = None
    if kind == "cleanup":
        static_waste = [[r, c] for r in range(env.base_map.shape[0]) for c in range(env.base_map.shape[1])
                        if env.base_map[r, c] in (b"H", b"R")]
    keys = ["a%d" % i for i in range(n)]
    rec = {k: [] for k in ("in_grid", "in_agents", "in_waste_perm", "seed", "actions", "out_grid", "out_agents", "base_rew",
                           "eaten", "second", "feature_obs", "obs_sha", "mt_pos", "out_waste_perm", "obs")}
    for s in range(count):
        grid, agents = craft(rs, kind, env, n)
        inject(env, kind, grid, agents)
        p = np.ones(n_act)
        p[:4] = 3.0  # move-heavy
        acts = rs.choice(n_act, size=n, p=p / p.sum())
        seed = int(rs.randint(1, 2 ** 31 - 1))
        rec["in_grid"].append(grid)
        rec["in_agents"].append(agents.astype(np.uint8))
        if kind == "cleanup":
            rec["in_waste_perm"].append(perm_of(env.waste_points, static_waste).astype(np.uint8))
        rec["seed"].append(seed)
        rec["actions"].append(acts.astype(np.uint8))
        env.timesteps = 5
        np.random.seed(seed)
        o, r, d, info = env.step({k: int(acts[i]) for i, k in enumerate(keys)})
        rec["out_grid"].append(grid_codes(env))
        rec["out_agents"].append(np.array([[a.pos[0], a.pos[1], ORIENT2INT[a.orientation]] for a in env.agents.values()], np.uint8))
        rec["base_rew"].append(np.array([r[k] for k in keys], np.int32))
        rec["eaten"].append(np.array([info[k]["eaten_apples"] for k in keys], np.uint8))
        sk = "cleaned_squares" if kind == "cleanup" else "eaten_close_apples"
        rec["second"].append(np.array([info[k][sk] for k in keys], np.uint8))
        rec["feature_obs"].append(np.stack([info[k]["feature_obs"] for k in keys]))
        ob = np.stack([obs_u8(o[k]["image"]) for k in keys])
        rec["obs_sha"].append(np.frombuffer(hashlib.sha256(ob.tobytes()).digest(), np.uint8))
        if s < 12:
            rec["obs"].append(ob)
        rec["mt_pos"].append(np.random.get_state()[2])
        if kind == "cleanup":
            rec["out_waste_perm"].append(perm_of(env.waste_points, static_waste).astype(np.uint8))
    out = {"kind": kind, "n": n, "firing": int(firing)}
    for k, v in rec.items():
        out[k] = np.array(v) if len(v) else np.zeros((0,), np.uint8)
    return out


def main():
    jobs = {
        "fuzz_cleanup_n8_fire": dict(kind="cleanup", n=8, count=400, seed0=11, firing=True),
        "fuzz_cleanup_n5": dict(kind="cleanup", n=5, count=250, seed0=12, firing=False),
        "fuzz_cleanup_n9_fire": dict(kind="cleanup", n=9, count=150, seed0=15, firing=True),
        "fuzz_harvest_n8_fire": dict(kind="harvest", n=8, count=400, seed0=13, firing=True),
        "fuzz_harvest_n3": dict(kind="harvest", n=3, count=200, seed0=14, firing=False),
    }
    for name, kw in jobs.items():
        out = run(**kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        moved = (out["in_agents"][:, :, :2] != out["out_agents"][:, :, :2]).any(axis=2).mean()
        shared = np.mean([len({tuple(a[:2]) for a in ag}) < len(ag) for ag in out["out_agents"]])
        print("%-24s %4d scenarios %6.1f KB  moved=%.2f shared-cell-after=%.3f rew[min,max]=[%d,%d]" % (
            name, len(out["seed"]), os.path.getsize(path) / 1024, moved, shared, out["base_rew"].min(), out["base_rew"].max()))


if __name__ == "__main__":
    main()
