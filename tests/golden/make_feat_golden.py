#!/usr/bin/env python3
"""Golden fixtures for the feature-vector envs HarvestFeatures / CleanupFeatures (`harvest` / `cleanup`,
BASELINE config 0), produced by RUNNING the upstream reference through ref_harness.py.
Build-container only; the fixtures (inputs + expected outputs) are committed.

Protocol: np.random.seed(s); random.seed(s) -> construct env (+ SeparateContractSubgameStage(convolutional=False))
-> reset() -> T steps, actions from RandomState(s + 1), action dict in key order.  Both RNG streams the envs use
(Python `random` for the spawn shuffle and the respawn doubles, numpy global for the orientations) are
fingerprinted after every call.

Usage: python tests/golden/make_feat_golden.py [names...]
"""
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference  # noqa: E402


def fp_np():
    st = np.random.get_state()
    return [int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)]


def fp_py():
    st = random.getstate()[1]
    return [int(st[624]), int(hashlib.sha256(np.array(st[:624], np.uint32).tobytes()).hexdigest()[:8], 16)]


def order_of(points, static_points, width):
    lut = {tuple(p): i for i, p in enumerate(static_points)}
    out = np.full((width,), -1, np.int16)
    for k, p in enumerate(points):
        out[k] = lut[tuple(p)]
    return out


def run_trace(R, kind, n, seed, T, episodes=2, contract=True, horizon=1000, action_p=None):
    np.random.seed(seed)
    random.seed(seed)
    if kind == "harvest_features":
        from environments.harvest_features import HarvestFeatures as Cls
        con = R.contract_list.HarvestFeaturemodLocalContract(n)
        n_act, nfeat = 7, 10 + 2 * n
    else:
        from environments.cleanup_features import CleanupFeatures as Cls
        con = R.contract_list.CleanupContract(n)
        n_act, nfeat = 8, 12 + n
    env = Cls(num_agents=n, horizon=horizon)
    top = R.SeparateContractSubgameStage(env, con, n, False) if contract else env
    keys = ["a%d" % i for i in range(n)]
    ars = np.random.RandomState(seed + 1)
    apple_pts = [list(p) for p in env.apple_points]
    waste_pts = [list(p) for p in env.waste_points] if kind == "cleanup_features" else []
    NA, NW = len(apple_pts), len(waste_pts)
    rec = {k: [] for k in ("actions", "agents", "base_rew", "rew", "info0", "info1", "feature_obs", "done", "apple_order",
                           "waste_order", "mt_np", "mt_py", "theta", "ep_start", "reset_agents", "reset_obs",
                           "reset_apple_order", "reset_waste_order", "reset_mt_np", "reset_mt_py")}

    def agents_now():
        return np.array([[env.agent_pos[k][0], env.agent_pos[k][1], env.agent_orientation[k]] for k in keys], np.uint8)

    out = {"kind": kind, "n": n, "seed": seed, "contract": int(contract), "horizon": horizon,
           "ctor_agents": agents_now(), "ctor_mt_np": np.array(fp_np(), np.int64), "ctor_mt_py": np.array(fp_py(), np.int64),
           "apple_points": np.array(apple_pts, np.int16), "waste_points": np.array(waste_pts, np.int16).reshape(-1, 2),
           "spawn_points": np.array(env.spawn_points, np.int16)}
    step_idx = 0
    for ep in range(episodes):
        o = top.reset()
        rec["ep_start"].append(step_idx)
        rec["reset_agents"].append(agents_now())
        rec["reset_obs"].append(np.stack([np.asarray(o[k], np.float64)[:nfeat] for k in keys]))
        rec["reset_apple_order"].append(order_of(env.current_apple_points, apple_pts, NA))
        rec["reset_waste_order"].append(order_of(env.current_waste_points, waste_pts, NW) if NW else np.zeros((0,), np.int16))
        rec["reset_mt_np"].append(fp_np())
        rec["reset_mt_py"].append(fp_py())
        rec["theta"].append(float(top.params["a0"][0]) if contract else 0.0)
        steps = T if isinstance(T, int) else T[ep]
        for t in range(steps):
            a = ars.randint(n_act, size=n) if action_p is None else ars.choice(n_act, size=n, p=action_p)
            acts = {k: int(a[i]) for i, k in enumerate(keys)}
            o, r, d, info = top.step(acts)
            rec["actions"].append(a.astype(np.uint8))
            rec["agents"].append(agents_now())
            rec["base_rew"].append(np.array([env.total_reward_dict[k][-1] for k in keys], np.float64))
            rec["rew"].append(np.array([float(r[k]) for k in keys], np.float64))
            if kind == "harvest_features":
                rec["info0"].append(np.array([info[k]["eaten_apples"] for k in keys], np.uint8))
                rec["info1"].append(np.array([info[k]["eaten_close_apples"] for k in keys], np.uint8))
                assert all(np.array_equal(info[k]["feature_obs"], np.asarray(o[k])[:nfeat]) for k in keys)
            else:
                rec["info0"].append(np.zeros((n,), np.uint8))
                rec["info1"].append(np.array([info[k]["cleaned_squares"] for k in keys], np.uint8))
            rec["feature_obs"].append(np.stack([np.asarray(o[k], np.float64)[:nfeat] for k in keys]))
            rec["done"].append(np.uint8(d["__all__"]))
            rec["apple_order"].append(order_of(env.current_apple_points, apple_pts, NA))
            rec["waste_order"].append(order_of(env.current_waste_points, waste_pts, NW) if NW else np.zeros((0,), np.int16))
            rec["mt_np"].append(fp_np())
            rec["mt_py"].append(fp_py())
            step_idx += 1
        mk = sorted(env.metrics.keys())
        out["metrics_keys_ep%d" % ep] = np.array(",".join(mk))
        out["metrics_vals_ep%d" % ep] = np.array([float(env.metrics[k]) for k in mk], np.float64)
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def run_image_trace(R, n, seed, T, horizon):
    """HarvestFeatures(image_obs=True): the observations are crops of the incrementally painted colour map
    (harvest_features.py:124-125,259-264); the feature rows still travel in the infos"""
    from environments.harvest_features import HarvestFeatures
    np.random.seed(seed)
    random.seed(seed)
    env = HarvestFeatures(num_agents=n, horizon=horizon, image_obs=True)
    keys = ["a%d" % i for i in range(n)]
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "obs", "feature_obs", "rew", "done", "ep_start", "reset_obs", "world")}
    out = {"kind": "harvest_features", "n": n, "seed": seed, "horizon": horizon,
           "obs_space_shape": np.array(env.observation_space.shape), "ctor_world": env.world_map_color.copy()}
    step_idx = 0
    for ep, steps in enumerate(T):
        o = env.reset()
        rec["ep_start"].append(step_idx)
        rec["reset_obs"].append(np.stack([np.array(o[k]) for k in keys]))
        for t in range(steps):
            a = ars.choice(7, size=n, p=[.2, .2, .2, .2, .1, .05, .05])
            o, r, d, info = env.step({k: int(a[i]) for i, k in enumerate(keys)})
            rec["actions"].append(a.astype(np.uint8))
            rec["obs"].append(np.stack([np.array(o[k]) for k in keys]))
            rec["feature_obs"].append(np.stack([np.asarray(info[k]["feature_obs"], np.float64) for k in keys]))
            rec["rew"].append(np.array([float(r[k]) for k in keys]))
            rec["done"].append(np.uint8(d["__all__"]))
            rec["world"].append(int(hashlib.sha256(env.world_map_color.tobytes()).hexdigest()[:8], 16))
            step_idx += 1
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    R = load_reference()
    S0 = 73907
    jobs = {
        "feat_harvest_n2": dict(kind="harvest_features", n=2, seed=S0 + 40, T=[1000, 80]),          # BASELINE config 0
        # walk-heavy, no contract: agents eat far more than they turn, so the per-step draw lists grow past 16 and 32 cells
        "feat_harvest_n2_walk": dict(kind="harvest_features", n=2, seed=S0 + 47, T=[600, 50], horizon=600, contract=False,
                                     action_p=[.24, .24, .24, .24, .01, .015, .015]),
        "feat_harvest_n5": dict(kind="harvest_features", n=5, seed=S0 + 41, T=[200, 60], horizon=200),
        "feat_harvest_n8_nocontract": dict(kind="harvest_features", n=8, seed=S0 + 42, T=[150, 40], horizon=150, contract=False),
        "feat_cleanup_n2": dict(kind="cleanup_features", n=2, seed=S0 + 43, T=[1000, 80],
                                action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "feat_cleanup_n5": dict(kind="cleanup_features", n=5, seed=S0 + 44, T=[250, 60], horizon=250,
                                action_p=[.1, .1, .15, .1, .05, .1, .1, .3]),
        "feat_cleanup_n8_nocontract": dict(kind="cleanup_features", n=8, seed=S0 + 45, T=[150, 40], horizon=150, contract=False),
    }
    only = set(sys.argv[1:])
    if not only or "featimg_harvest_n6" in only:  # not a feat_* name: the oracle replays skip it (adapter-level rendering)
        out = run_image_trace(R, n=6, seed=S0 + 46, T=[260, 60], horizon=260)
        path = os.path.join(HERE, "featimg_harvest_n6.npz")
        np.savez_compressed(path, **out)
        print("%-30s steps=%5d  %7.1f KB" % ("featimg_harvest_n6", len(out["actions"]), os.path.getsize(path) / 1024))
    for name, kw in jobs.items():
        if only and name not in only:
            continue
        out = run_trace(R, **kw)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-30s steps=%5d  %7.1f KB" % (name, len(out["actions"]), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
