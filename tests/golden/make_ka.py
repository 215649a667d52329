#!/usr/bin/env python3
"""Named known-answer scenarios: hand-placed agents, one reference step each.
Build-container only.  Output: tests/golden/ka_cleanup.npz and ka_harvest.npz (same record layout as the fuzz
fixtures, plus `names`)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fuzz import inject  # noqa: E402
from make_golden import ORIENT2INT, grid_codes, obs_u8, perm_of  # noqa: E402
from ref_harness import load_reference  # noqa: E402

L, R, U, D, STAY, CW, CCW, CLEAN, FIRE = 0, 1, 2, 3, 4, 5, 6, 7, 8
UP, RIGHT, DOWN, LEFT = 0, 1, 2, 3
FAR = [(20, 7, UP), (22, 9, UP), (18, 11, UP), (16, 7, UP)]  # parking spots for agents not under test


def base_grid(env, apples=(), waste_h=None):
    g = np.zeros(env.base_map.shape, np.uint8)
    for r in range(g.shape[0]):
        for c in range(g.shape[1]):
            ch = env.base_map[r, c]
            g[r, c] = {b"@": 1, b"H": 3, b"R": 4, b"S": 5}.get(ch, 0)
    if waste_h is not None:  # exactly `waste_h` waste cells hold H (row-major), the rest R
        k = 0
        for r in range(g.shape[0]):
            for c in range(g.shape[1]):
                if env.base_map[r, c] in (b"H", b"R"):
                    g[r, c] = 3 if k < waste_h else 4
                    k += 1
    for (r, c) in apples:
        g[r, c] = 2
    return g


def scenarios(env):
    S = []

    def add(name, agents, acts, grid=None, seeds=(1,)):
        ag = list(agents) + FAR[len(agents):]
        ac = list(acts) + [STAY] * (4 - len(acts))
        for sd in seeds:
            S.append((name + ("/s%d" % sd if len(seeds) > 1 else ""), np.array(ag), np.array(ac),
                      base_grid(env) if grid is None else grid, sd))

    add("KA1 lone move into free cell", [(5, 8, UP)], [U])
    add("KA1b moves are egocentric (facing LEFT, MOVE_UP goes left)", [(5, 8, LEFT)], [U])
    add("KA2 move into wall stays", [(1, 7, UP)], [U])
    add("KA3 two agents, one free cell: shuffle winner", [(5, 7, UP), (5, 9, UP)], [R, L], seeds=(1, 2, 3, 4, 5, 6))
    add("KA4a contested cell, occupant STAYs", [(5, 7, UP), (5, 9, UP), (5, 8, UP)], [R, L, STAY])
    add("KA4b contested cell, occupant TURNs", [(5, 7, UP), (5, 9, UP), (5, 8, UP)], [R, L, CW])
    add("KA4c contested cell, occupant CLEANs", [(5, 7, UP), (5, 9, UP), (5, 8, UP)], [R, L, CLEAN])
    add("KA4d move into a TURNing agent is refused", [(5, 7, UP), (5, 8, UP)], [R, CCW])
    add("KA5 head-on swap: both stay", [(5, 7, UP), (5, 8, UP)], [R, L])
    add("KA6 chain follow: both move", [(5, 7, UP), (5, 8, UP)], [R, R])
    add("KA6b chain follow, leader listed first", [(5, 8, UP), (5, 7, UP)], [R, R])
    add("KA7 3-cycle cannot exist on a grid; 4-cycle rotates all", [(5, 7, UP), (5, 8, UP), (6, 8, UP), (6, 7, UP)], [R, D, L, U])
    add("KA8 occupant leaves but gets blocked: two agents share a cell",
        [(5, 8, UP), (5, 9, UP), (4, 8, UP), (6, 8, UP)], [R, STAY, D, U], seeds=(1, 2, 3, 4))
    add("KA8b contested cell vacated by a successful mover", [(5, 8, UP), (4, 8, UP), (6, 8, UP)], [R, D, U], seeds=(1, 2, 3))
    add("KA9a CLEAN: first H is cleaned, beam stops, side rays start beside the agent", [(2, 8, LEFT)], [CLEAN])
    add("KA9b CLEAN stops at an agent", [(2, 9, LEFT), (2, 8, UP)], [CLEAN, STAY])
    add("KA9c CLEAN stops at the wall", [(1, 3, UP)], [CLEAN])
    add("KA9d two cleaners on the same H: shuffle order decides", [(2, 7, LEFT), (2, 8, LEFT)], [CLEAN, CLEAN], seeds=(1, 2, 3, 4))
    add("KA10a FIRE: -1 firer, -50 first agent hit, passes over H", [(2, 9, LEFT), (2, 3, UP), (2, 5, UP)], [FIRE, STAY, STAY])
    add("KA10b FIRE side ray hits", [(6, 8, UP), (5, 9, UP), (3, 7, UP)], [FIRE, STAY, STAY])
    add("KA10c FIRE at a shared cell hits the later agent", [(6, 8, UP), (4, 8, UP), (4, 8, UP)], [FIRE, STAY, STAY])
    add("KA11a eat an apple (+1, eaten_apples)", [(5, 11, UP)], [R], grid=base_grid(env, apples=[(5, 12)]))
    add("KA11b two agents reach one apple: first in shuffle eats... then both on it?", [(5, 11, UP), (5, 13, UP)], [R, L],
        grid=base_grid(env, apples=[(5, 12)]), seeds=(1, 2, 3))
    for nh in (0, 1, 47, 48, 60, 119):
        add("KA12 waste density with %d H" % nh, [(5, 8, UP)], [STAY], grid=base_grid(env, waste_h=nh), seeds=(7, 8))
    for o in (UP, RIGHT, DOWN, LEFT):
        add("KA13 crop in the map corner, orientation %d" % o, [(1, 1, o), (23, 16, o), (1, 16, o), (23, 1, o)], [STAY] * 4)
    return S


def harvest_scenarios(env):
    """KA14 (eaten_close_apples around the <4 threshold), harvest FIRE (action 7), apple under a staying agent"""
    H, W = env.base_map.shape
    apple_cells = [(r, c) for r in range(H) for c in range(W) if env.base_map[r, c] == b"A"]
    far = [(14, 1, UP), (14, 3, UP), (14, 5, UP), (14, 9, UP)]
    S = []

    def grid_with(apples):
        g = np.zeros((H, W), np.uint8)
        g[env.base_map == b"@"] = 1
        for (r, c) in apples:
            g[r, c] = 2
        return g

    def add(name, agents, acts, apples, seeds=(1,)):
        ag = list(agents) + far[len(agents):]
        ac = list(acts) + [STAY] * (4 - len(acts))
        for sd in seeds:
            S.append((name + ("/s%d" % sd if len(seeds) > 1 else ""), np.array(ag), np.array(ac), grid_with(apples), sd))

    # a target apple cell with a free cell on its left and at least 5 apple cells within distance^2 <= 5
    def near(t):
        return [q for q in apple_cells if q != t and (q[0] - t[0]) ** 2 + (q[1] - t[1]) ** 2 <= 5]
    t = next(q for q in apple_cells if q[0] < 12 and env.base_map[q[0], q[1] - 1] == b" " and len(near(q)) >= 5)
    for k in (0, 2, 3, 4, 5):
        add("KA14 eat an apple with %d other apples within d^2<=5" % k, [(t[0], t[1] - 1, UP)], [R], [t] + near(t)[:k])
    add("KA14b eat an apple with 1 other apple within d^2<=5", [(t[0], t[1] - 1, UP)], [R], [t] + near(t)[:1])
    add("KA10h harvest FIRE (action 7): -1 firer, -50 first agent hit, passes over apples",
        [(t[0], t[1] - 1, RIGHT), (t[0], t[1] + 3, UP), (t[0], t[1] + 4, UP)], [7, STAY, STAY], [t] + near(t)[:3])
    add("KA10i harvest FIRE side ray", [(t[0], t[1] - 1, RIGHT), (t[0] + 1, t[1] + 1, UP)], [7, STAY], [t])
    add("KA11c no respawn under a staying agent, dense neighbourhood", [(t[0], t[1], UP)], [STAY], near(t), seeds=tuple(range(1, 25)))
    add("KA11d respawn of the same cell without the agent", [], [], near(t), seeds=tuple(range(1, 25)))
    return S


def record_harvest(Rf):
    np.random.seed(3)
    env = Rf.HarvestEnv(num_agents=4, disable_firing=False)
    env.reset()
    keys = ["a%d" % i for i in range(4)]
    rec = {k: [] for k in ("in_grid", "in_agents", "seed", "actions", "out_grid", "out_agents", "base_rew",
                           "eaten", "second", "feature_obs", "obs_sha", "mt_pos", "obs", "names")}
    for name, agents, acts, grid, seed in harvest_scenarios(env):
        inject(env, "harvest", grid, agents)
        rec["names"].append(name)
        rec["in_grid"].append(grid)
        rec["in_agents"].append(agents.astype(np.uint8))
        rec["seed"].append(seed)
        rec["actions"].append(acts.astype(np.uint8))
        env.timesteps = 5
        np.random.seed(seed)
        o, r, d, info = env.step({k: int(acts[i]) for i, k in enumerate(keys)})
        rec["out_grid"].append(grid_codes(env))
        rec["out_agents"].append(np.array([[a.pos[0], a.pos[1], ORIENT2INT[a.orientation]] for a in env.agents.values()], np.uint8))
        rec["base_rew"].append(np.array([r[k] for k in keys], np.int32))
        rec["eaten"].append(np.array([info[k]["eaten_apples"] for k in keys], np.uint8))
        rec["second"].append(np.array([info[k]["eaten_close_apples"] for k in keys], np.uint8))
        rec["feature_obs"].append(np.stack([info[k]["feature_obs"] for k in keys]))
        ob = np.stack([obs_u8(o[k]["image"]) for k in keys])
        rec["obs_sha"].append(np.frombuffer(hashlib.sha256(ob.tobytes()).digest(), np.uint8))
        if len(rec["obs"]) < 10:
            rec["obs"].append(ob)
        rec["mt_pos"].append(np.random.get_state()[2])
        if "/s" not in name or name.endswith("/s1"):
            print("%-75s rew %s close %s" % (name[:75], rec["base_rew"][-1].tolist(), rec["second"][-1].tolist()))
    out = {"kind": "harvest", "n": 4, "firing": 1}
    for k, v in rec.items():
        out[k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "ka_harvest.npz"), **out)
    g = out
    t_in, t_out = g["in_grid"], g["out_grid"]
    print("spawns per scenario (KA11c / KA11d):", [(int(((t_out[i] == 2) & (t_in[i] != 2)).sum())) for i, nm in enumerate(g["names"]) if nm.startswith("KA11")])


def main():
    Rf = load_reference()
    record_harvest(Rf)
    np.random.seed(3)
    env = Rf.CleanupEnv(num_agents=4, disable_firing=False)
    env.reset()
    static_waste = [[r, c] for r in range(25) for c in range(18) if env.base_map[r, c] in (b"H", b"R")]
    keys = ["a%d" % i for i in range(4)]
    rec = {k: [] for k in ("in_grid", "in_agents", "in_waste_perm", "seed", "actions", "out_grid", "out_agents", "base_rew",
                           "eaten", "second", "feature_obs", "obs_sha", "mt_pos", "out_waste_perm", "obs", "names")}
    for name, agents, acts, grid, seed in scenarios(env):
        inject(env, "cleanup", grid, agents)
        rec["names"].append(name)
        rec["in_grid"].append(grid)
        rec["in_agents"].append(agents.astype(np.uint8))
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
        rec["second"].append(np.array([info[k]["cleaned_squares"] for k in keys], np.uint8))
        rec["feature_obs"].append(np.stack([info[k]["feature_obs"] for k in keys]))
        ob = np.stack([obs_u8(o[k]["image"]) for k in keys])
        rec["obs_sha"].append(np.frombuffer(hashlib.sha256(ob.tobytes()).digest(), np.uint8))
        rec["obs"].append(ob)
        rec["mt_pos"].append(np.random.get_state()[2])
        rec["out_waste_perm"].append(perm_of(env.waste_points, static_waste).astype(np.uint8))
        print("%-75s pos %s rew %s cleaned %s" % (name[:75], rec["out_agents"][-1][:, :2].tolist()[:3], rec["base_rew"][-1].tolist(),
                                                  rec["second"][-1].tolist()))
    out = {"kind": "cleanup", "n": 4, "firing": 1}
    for k, v in rec.items():
        out[k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "ka_cleanup.npz"), **out)


if __name__ == "__main__":
    main()
