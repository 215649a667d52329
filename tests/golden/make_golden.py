#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/*.npz by RUNNING the upstream
reference (imported from /root/reference through ref_harness.py).  Build-container
only; the fixtures (inputs + expected outputs, no source text) are committed and are
what travels to the GPU box.

Protocol: np.random.seed(s); random.seed(s) -> construct base env
(+ contract + SeparateContractSubgameStage wrapper) -> reset() -> T steps with the
action dict built in key order a0..a{n-1}.  `episodes` > 1 calls reset() again after
each episode (pins the persistent spawn/waste shuffles across resets).

Usage:  python tests/golden/make_golden.py            # writes all fixtures
"""
import contextlib
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference  # noqa: E402

CHAR2CODE = {b" ": 0, b"@": 1, b"A": 2, b"H": 3, b"R": 4, b"S": 5}
ORIENT2INT = {"UP": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3}


def grid_codes(env):
    wm = env.world_map
    out = np.zeros(wm.shape, np.uint8)
    for ch, code in CHAR2CODE.items():
        out[wm == ch] = code
    known = np.isin(wm, list(CHAR2CODE.keys()))
    assert known.all(), "unexpected cell char in world_map"
    return out


def obs_u8(img):
    u = np.rint(img * 255.0).astype(np.uint8)
    assert np.array_equal(u / 255, img), "obs is not u8/255"
    return u


def _mt_fingerprint():
    st = np.random.get_state()
    return np.array([st[2], int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)], np.int64)


mt_fingerprint = _mt_fingerprint


def perm_of(points, static_points):
    """index (into the static row-major list) of each entry of a shuffled point list;
    duplicates (Cleanup spawn list) resolve to the first matching static index."""
    lut = {}
    for i, p in enumerate(static_points):
        lut.setdefault(tuple(p), i)
    return np.array([lut[tuple(p)] for p in points], np.int16)


def run_grid_trace(R, kind, n, seed, T, episodes=1, firing=False, contract=True, action_p=None,
                   act_seed=None, store_obs_steps=None, extra_env_kwargs=None, stream=None):
    """`stream`: a counter_stream.CounterWords the caller has routed np.random to (make_counter_golden.py) — the constructor,
    every reset() and every step() are then one operation each on that stream, and the per-step generator record is the
    stream's (key0, key1, generation) row instead of the MT19937 fingerprint"""
    global mt_fingerprint
    op = stream.op if stream is not None else contextlib.nullcontext
    if stream is not None:
        mt_fingerprint = stream.state_row
    try:
        return _run_grid_trace(R, kind, n, seed, T, episodes, firing, contract, action_p, act_seed, store_obs_steps,
                               extra_env_kwargs, op, stream is not None)
    finally:
        mt_fingerprint = _mt_fingerprint


def _run_grid_trace(R, kind, n, seed, T, episodes, firing, contract, action_p, act_seed, store_obs_steps, extra_env_kwargs,
                    op, counter):
    np.random.seed(seed)
    random.seed(seed)
    kw = dict(num_agents=n, disable_firing=not firing)
    kw.update(extra_env_kwargs or {})
    with op():
        if kind == "cleanup":
            env = R.CleanupEnv(**kw)
            con = R.contract_list.CleanupContract(n)
            n_act = 9 if firing else 8
        else:
            env = R.HarvestEnv(**kw)
            con = R.contract_list.HarvestFeaturemodLocalContract(n)
            n_act = 8 if firing else 7
    horizon = env.horizon
    static_spawn = sorted({tuple(p) for p in env.spawn_points})
    top = R.SeparateContractSubgameStage(env, con, n, True) if contract else env
    ars = np.random.RandomState(seed + 1 if act_seed is None else act_seed)
    keys = ["a%d" % i for i in range(n)]
    nfeat = (12 + n) if kind == "cleanup" else (10 + 2 * n)
    if store_obs_steps is None:
        store_obs_steps = T

    rec = {k: [] for k in ("actions", "grid", "agents", "base_rew", "rew", "eaten", "cleaned", "eaten_close",
                           "feature_obs", "done", "obs", "obs_sha", "mt", "theta", "waste_perm", "spawn_perm",
                           "ep_start", "reset_grid", "reset_agents", "reset_obs", "reset_mt", "reset_features")}
    static_waste = None
    if kind == "cleanup":
        static_waste = [[r, c] for r in range(env.base_map.shape[0]) for c in range(env.base_map.shape[1])
                        if env.base_map[r, c] in (b"H", b"R")]
    rec["ctor_spawn_perm"] = np.array([static_spawn.index(tuple(p)) for p in env.spawn_points], np.int16)
    rec["ctor_agents"] = np.array([[a.pos[0], a.pos[1], ORIENT2INT[a.orientation]] for a in env.agents.values()], np.int16)
    rec["ctor_mt"] = mt_fingerprint()

    step_idx = 0
    for ep in range(episodes):
        with op():
            o = top.reset()
        rec["ep_start"].append(step_idx)
        rec["reset_grid"].append(grid_codes(env))
        rec["reset_agents"].append(np.array([[a.pos[0], a.pos[1], ORIENT2INT[a.orientation]] for a in env.agents.values()], np.uint8))
        if env.image_obs:
            rec["reset_obs"].append(np.stack([obs_u8(o[k]["image"]) for k in keys]))
        rec["reset_mt"].append(mt_fingerprint())
        if not env.image_obs:
            rec["reset_features"].append(np.stack([o[k] if not contract else o[k] for k in keys])[:, :nfeat])
        rec["theta"].append(float(top.params["a0"][0]) if contract else 0.0)
        steps_this_ep = T if isinstance(T, int) else T[ep]
        for t in range(steps_this_ep):
            if action_p is None:
                a = ars.randint(n_act, size=n)
            else:
                a = ars.choice(n_act, size=n, p=action_p)
            acts = {k: int(a[i]) for i, k in enumerate(keys)}
            with op():
                o, r, d, info = top.step(acts)
            base = env.total_reward_dict
            rec["actions"].append(a.astype(np.uint8))
            rec["grid"].append(grid_codes(env))
            rec["agents"].append(np.array([[ag.pos[0], ag.pos[1], ORIENT2INT[ag.orientation]] for ag in env.agents.values()], np.uint8))
            rec["base_rew"].append(np.array([base[k][-1] for k in keys], np.float64))
            rec["rew"].append(np.array([float(r[k]) for k in keys], np.float64))
            rec["eaten"].append(np.array([info[k]["eaten_apples"] for k in keys], np.uint8))
            rec["cleaned"].append(np.array([info[k].get("cleaned_squares", 0) for k in keys], np.uint8))
            rec["eaten_close"].append(np.array([info[k].get("eaten_close_apples", 0) for k in keys], np.uint8))
            rec["feature_obs"].append(np.stack([info[k]["feature_obs"] for k in keys]).astype(np.float64))
            assert rec["feature_obs"][-1].shape == (n, nfeat)
            rec["done"].append(np.uint8(d["__all__"]))
            if env.image_obs:
                ob = np.stack([obs_u8(o[k]["image"]) for k in keys])
            else:
                assert all(np.array_equal(o[k][:nfeat], info[k]["feature_obs"]) for k in keys)
                ob = np.zeros((n, 15, 15, 3), np.uint8)
            if step_idx < store_obs_steps:
                rec["obs"].append(ob)
            rec["obs_sha"].append(np.frombuffer(hashlib.sha256(ob.tobytes()).digest(), np.uint8))
            rec["mt"].append(mt_fingerprint())
            if kind == "cleanup":
                rec["waste_perm"].append(perm_of(env.waste_points, static_waste).astype(np.uint8))
            step_idx += 1
            if d["__all__"]:
                assert t == horizon - 1
        mk = sorted(env.metrics.keys())
        rec["metrics_keys_ep%d" % ep] = np.array(",".join(mk))
        rec["metrics_vals_ep%d" % ep] = np.array([float(env.metrics[k]) for k in mk], np.float64)
        sp = [static_spawn.index(tuple(p)) for p in env.spawn_points]
        rec["spawn_perm"].append(np.array(sp, np.int16))

    ek = extra_env_kwargs or {}
    out = {"rng_mode": "counter"} if counter else {}
    out.update({"kind": kind, "n": n, "seed": seed, "firing": int(firing), "contract": int(contract), "horizon": horizon,
           "collective": int(bool(ek.get("use_collective_reward"))), "inequity": int(bool(ek.get("inequity_averse_reward"))),
           "alpha": float(ek.get("alpha", 0.0)), "beta": float(ek.get("beta", 0.0)), "image_obs": int(ek.get("image_obs", True)),
           "static_spawn": np.array(static_spawn, np.int16)})
    for k, v in rec.items():
        if isinstance(v, list):
            out[k] = np.array(v) if len(v) else np.zeros((0,), np.uint8)
        else:
            out[k] = v
    if static_waste is not None:
        out["static_waste"] = np.array(static_waste, np.int16)
    if "ascii_map" in ek:
        out["ascii_map"] = np.array([str(r) for r in ek["ascii_map"]])
    return out


def run_selfdrive_trace(R, n, seed, max_steps=400, episodes=2, act_scale=0.15, collision_on=False):
    """Selfdrive + SelfdriveContractDistprop.  RLlib stops sending actions for agents whose
    done flag is set; the action dict therefore holds the not-yet-done agents only.  Actions are
    float32 values handed over as float64 (SURVEY §7 hard part 6: reproduces the float64
    arithmetic of the numpy-1.x era the reference was written for)."""
    np.random.seed(seed)
    random.seed(seed)
    env = R.SelfAcceleratingCarEnv(num_agents=n, collision_on=collision_on)
    con = R.contract_list.SelfdriveContractDistprop(n)
    top = R.SeparateContractSubgameStage(env, con, n, False)
    ars = np.random.RandomState(seed + 1)
    keys = ["a%d" % i for i in range(n)]
    L = 2 * n + 5
    rec = {k: [] for k in ("actions", "active", "obs", "rew", "base_rew", "done", "pos", "vel", "just_passed", "theta",
                           "ep_start", "reset_obs", "transfers_metric", "crossed", "dist_to_front", "amb_rank", "amb_dtf",
                           "is_crashed")}
    step_idx = 0
    for ep in range(episodes):
        o = top.reset()
        rec["ep_start"].append(step_idx)
        rec["reset_obs"].append(np.stack([o[k] for k in keys]))
        rec["theta"].append(float(top.params["a0"][0]))
        done = {k: False for k in keys}
        for t in range(max_steps):
            a32 = ars.uniform(-act_scale, act_scale, size=n).astype(np.float32)
            active = np.array([not done[k] for k in keys], np.uint8)
            acts = {k: np.array([float(a32[i])], np.float64) for i, k in enumerate(keys) if active[i]}
            o, r, d, info = top.step(acts)
            ob = np.full((n, L + 2), np.nan)
            rw = np.full((n,), np.nan)
            jp = np.zeros((n,), np.uint8)
            for i, k in enumerate(keys):
                if active[i]:
                    ob[i] = o[k]
                    rw[i] = r[k]
                    jp[i] = info[k]["just_passed"]
            rec["actions"].append(a32)
            rec["active"].append(active)
            rec["obs"].append(ob)
            rec["rew"].append(rw)
            rec["done"].append(np.array([d[k] for k in keys] + [d["__all__"]], np.uint8))
            rec["pos"].append(np.array([env.agent_positions[k] for k in keys]))
            rec["vel"].append(np.array([env.agent_vels[k] for k in keys]))
            rec["just_passed"].append(jp)
            # update_infos' bookkeeping (:127-149) and the infos the first acting key carries (:183-189, :203-204)
            first = next(k for i, k in enumerate(keys) if active[i])
            assert all(info[k]["ambulance_rank"] == 0.0 and info[k]["is_crashed"] == 0 for k in acts if k != first)
            rec["dist_to_front"].append(np.array([env.dist_to_front[k] for k in keys], np.float64))
            rec["amb_rank"].append(float(info[first]["ambulance_rank"]))
            rec["amb_dtf"].append(float(info[first]["ambulance_dist_to_front"]))
            rec["is_crashed"].append(int(info[first]["is_crashed"]))
            rec["transfers_metric"].append(float(env.metrics["transfers"]))
            cr = [int(c[1:]) for c in env.crossed_agents] + [-1] * (n - len(env.crossed_agents))
            rec["crossed"].append(np.array(cr, np.int8))
            step_idx += 1
            for k in keys:
                done[k] = bool(d[k])
            if d["__all__"]:
                break
    out = {"kind": "selfdrive", "n": n, "seed": seed, "collision_on": int(collision_on)}
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


CLEANUP_SMALL = ["@@@@@@@@@@@@",
                 "@HRH   BBBB@",
                 "@RHR P  BBB@",
                 "@HRH   BBBB@",
                 "@RHRSSS BBB@",
                 "@HRH P BBBB@",
                 "@RHR    BBB@",
                 "@HRH P  BBB@",
                 "@RHR   BBBB@",
                 "@@@@@@@@@@@@"]
CLEANUP_MID = ["@@@@@@@@@@@@@@@@"] + ["@%s%s%s@" % ("HRHRH" if r % 2 else "RHRHR", "  P " if r % 5 == 0 else ("SSSS" if r == 8 else "    "),
                                               "BBBBB" if r % 3 else " BBBB") for r in range(18)] + ["@@@@@@@@@@@@@@@@"]
HARVEST_SMALL = ["@@@@@@@@@@@@@@@@@@@@",
                 "@ P  A   AA    P   @",
                 "@   AAA   A  A     @",
                 "@ A  A      AAA  A @",
                 "@AAA    P    A  AAA@",
                 "@ A   A   A      A @",
                 "@    AAA AAA   P   @",
                 "@  P  A   A    A   @",
                 "@       A     AAA  @",
                 "@  A   AAA     A   @",
                 "@ AAA   A   P      @",
                 "@@@@@@@@@@@@@@@@@@@@"]
CLEANER_P = [.1, .1, .15, .1, .05, .1, .1, .3]


def custom_map_jobs(S0):
    return {
        "m1_cleanup_small_n3": dict(kind="cleanup", n=3, seed=S0 + 60, T=[100, 100, 60], episodes=3, store_obs_steps=40, action_p=CLEANER_P,
                                    extra_env_kwargs=dict(ascii_map=CLEANUP_SMALL, horizon=100)),
        "m2_cleanup_mid_n4_fire": dict(kind="cleanup", n=4, seed=S0 + 61, T=[150, 90], episodes=2, store_obs_steps=30, firing=True,
                                       action_p=CLEANER_P + [0.0], extra_env_kwargs=dict(ascii_map=CLEANUP_MID, horizon=150)),
        "m3_harvest_small_n4": dict(kind="harvest", n=4, seed=S0 + 62, T=[120, 80], episodes=2, store_obs_steps=40, firing=True,
                                    extra_env_kwargs=dict(ascii_map=HARVEST_SMALL, horizon=120)),
        "m4_cleanup_small_n1_nocontract": dict(kind="cleanup", n=1, seed=S0 + 63, T=150, store_obs_steps=10, contract=False, action_p=CLEANER_P,
                                               extra_env_kwargs=dict(ascii_map=CLEANUP_SMALL)),
    }


def main():
    R = load_reference()
    S0 = 73907
    jobs = {
        # name: (fn, kwargs)
        "g1_cleanup_n4": dict(kind="cleanup", n=4, seed=S0, T=[1000, 150], episodes=2, store_obs_steps=120),
        "g2_harvest_n8": dict(kind="harvest", n=8, seed=S0 + 1, T=[1000, 60], episodes=2, store_obs_steps=100),
        "g3_cleanup_n8": dict(kind="cleanup", n=8, seed=S0 + 2, T=[1000, 100], episodes=2, store_obs_steps=100),
        # CLEAN-heavy policy: drives waste density through both thresholds so apples/waste respawn
        "g3b_cleanup_n8_cleaner": dict(kind="cleanup", n=8, seed=S0 + 3, T=700, store_obs_steps=0,
                                       action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "g3c_cleanup_n4_cleaner": dict(kind="cleanup", n=4, seed=S0 + 4, T=600, store_obs_steps=40,
                                       action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "g4_cleanup_n8_fire": dict(kind="cleanup", n=8, seed=S0 + 5, T=300, firing=True, store_obs_steps=60),
        "g4_harvest_n5_fire": dict(kind="harvest", n=5, seed=S0 + 6, T=300, firing=True, store_obs_steps=60),
        "g6_cleanup_n2": dict(kind="cleanup", n=2, seed=S0 + 7, T=300, store_obs_steps=60,
                              action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "g6_harvest_n2": dict(kind="harvest", n=2, seed=S0 + 8, T=300, store_obs_steps=60),
        "g6_cleanup_n1": dict(kind="cleanup", n=1, seed=S0 + 9, T=200, store_obs_steps=30,
                              action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "g6_harvest_n1_nocontract": dict(kind="harvest", n=1, seed=S0 + 10, T=200, store_obs_steps=30, contract=False),
        "g8_cleanup_n9": dict(kind="cleanup", n=9, seed=S0 + 11, T=200, store_obs_steps=20, firing=True),
    }
    jobs["g9_cleanup_n4_collective_fire"] = dict(kind="cleanup", n=4, seed=S0 + 30, T=250, firing=True, contract=False,
                                                 store_obs_steps=10, extra_env_kwargs=dict(use_collective_reward=True))
    jobs["g9_cleanup_n4_inequity_fire"] = dict(kind="cleanup", n=4, seed=S0 + 31, T=250, firing=True, contract=False,
                                               store_obs_steps=10,
                                               extra_env_kwargs=dict(inequity_averse_reward=True, alpha=5.0, beta=0.05))
    jobs["g9_harvest_n3_inequity_contract"] = dict(kind="harvest", n=3, seed=S0 + 32, T=250, store_obs_steps=10,
                                                   extra_env_kwargs=dict(inequity_averse_reward=True, alpha=0.5, beta=0.25))
    jobs["g9_cleanup_n3_features"] = dict(kind="cleanup", n=3, seed=S0 + 33, T=200, store_obs_steps=0, contract=False,
                                          action_p=[.1, .1, .15, .1, .05, .1, .1, .3], extra_env_kwargs=dict(image_obs=False))
    jobs["g9_harvest_n4_features"] = dict(kind="harvest", n=4, seed=S0 + 34, T=200, store_obs_steps=0, contract=False,
                                          extra_env_kwargs=dict(image_obs=False))
    # whole (short-horizon) episodes under inequity aversion: the float reward accumulators behind raw_env_rewards,
    # equality and sustainability at the done step, and their transferred twins, across a reset
    jobs["g9b_cleanup_n3_inequity_done"] = dict(kind="cleanup", n=3, seed=S0 + 35, T=[60, 45], episodes=2, firing=True,
                                                contract=False, store_obs_steps=5,
                                                extra_env_kwargs=dict(inequity_averse_reward=True, alpha=5.0, beta=0.05, horizon=60))
    jobs["g9b_harvest_n4_inequity_contract_done"] = dict(kind="harvest", n=4, seed=S0 + 36, T=[80, 30], episodes=2,
                                                         store_obs_steps=5,
                                                         extra_env_kwargs=dict(inequity_averse_reward=True, alpha=0.5, beta=0.25, horizon=80))
    # custom layouts (the reference's `ascii_map` argument; walled in, within the engine's caps — include/contracts_engine.h):
    # a small cleanup map whose waste list fits one 64-lane register, a larger one that needs two, a small harvest map
    for name, kw in custom_map_jobs(S0).items():
        jobs[name] = kw
    for s in range(6):  # short multi-seed traces (RNG / reset variety)
        jobs["g7_cleanup_n8_s%d" % s] = dict(kind="cleanup", n=8, seed=1000 + 17 * s, T=120, store_obs_steps=8,
                                             action_p=[.1, .1, .15, .1, .05, .1, .1, .3] if s % 2 else None)
        jobs["g7_harvest_n8_s%d" % s] = dict(kind="harvest", n=8, seed=2000 + 17 * s, T=120, store_obs_steps=8)
    only = set(sys.argv[1:])
    for name, kw in jobs.items():
        if only and name not in only:
            continue
        out = run_grid_trace(R, **kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s steps=%5d  %7.1f KB" % (name, len(out["actions"]), os.path.getsize(path) / 1024))
    for name, kw in {"g5_selfdrive_n4": dict(n=4, seed=S0 + 20, episodes=3),
                     "g5_selfdrive_n2": dict(n=2, seed=S0 + 21, episodes=3),
                     "g5_selfdrive_n6": dict(n=6, seed=S0 + 22, episodes=2),
                     # collision_on=True: disobeying the merge order ends the episode (check_if_crashed :81-90, :195-215)
                     "g5_selfdrive_n4_collision": dict(n=4, seed=S0 + 23, episodes=6, collision_on=True),
                     "g5_selfdrive_n3_collision": dict(n=3, seed=S0 + 24, episodes=6, collision_on=True, act_scale=0.05)}.items():
        if only and name not in only:
            continue
        out = run_selfdrive_trace(R, **kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s steps=%5d  %7.1f KB" % (name, len(out["actions"]), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
