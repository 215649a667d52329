#!/usr/bin/env python3
"""Golden fixtures for the negotiate / combined stages and the NegotiationSolver of two_stage_train.py (:188-470, :619-776) over CleanupEnv, produced by
RUNNING the upstream reference (ref_harness.py).  The frozen subgame policies of the negotiate stage are the harness's
StubPPOTrainer (deterministic, observation-independent).  Build-container only."""
import hashlib
import os
import random
import sys

import numpy as np
import scipy.special  # noqa: F401  (the reference uses scipy.special.softmax after a bare `import scipy`)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import install_stub_trainer, load_reference  # noqa: E402


def fps():
    st, ps = np.random.get_state(), random.getstate()[1]
    return [int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16),
            int(ps[624]), int(hashlib.sha256(np.array(ps[:624], np.uint32).tobytes()).hexdigest()[:8], 16)]


def pack_obs(o, keys):
    img = np.stack([np.rint(np.asarray(o[k]["image"]) * 255).astype(np.uint8) for k in keys])
    con = np.stack([np.asarray(o[k]["contract"], np.float64) for k in keys])
    return np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8), con


def run_combined(R, n, seed, horizon, T):
    import environments.two_stage_train as tst
    np.random.seed(seed)
    random.seed(seed)
    base = R.CleanupEnv(num_agents=n, horizon=horizon)
    con = R.contract_list.CleanupContract(n)
    env = tst.SeparateContractCombinedStage(base, con, n, True)
    keys = ["a%d" % i for i in range(n)]
    na = base.continuous_action_space.shape[0]
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "obs_sha", "contract_obs", "rew", "done", "fp", "reset_at")}
    o = env.reset()
    rec["reset_at"].append(0)
    sha, c0 = pack_obs(o, keys)
    out = {"kind": "cleanup", "n": n, "seed": seed, "horizon": horizon, "reset_sha": sha, "reset_contract": c0,
           "action_low": env.action_space.low, "action_high": env.action_space.high}
    for t in range(T):
        a = np.concatenate([ars.uniform(-3, 3, size=(n, na)), ars.uniform(0, 0.2, size=(n, 1)), ars.uniform(0.3, 1.0, size=(n, 1))], axis=1)
        o, r, d, info = env.step({k: a[i] for i, k in enumerate(keys)})
        sha, c = pack_obs(o, keys)
        rec["actions"].append(a)
        rec["obs_sha"].append(sha)
        rec["contract_obs"].append(c)
        rec["rew"].append([float(r[k]) for k in keys])
        rec["done"].append(np.uint8(d["__all__"]))
        rec["fp"].append(fps())
        if d["__all__"]:
            o = env.reset()
            rec["reset_at"].append(t + 1)
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def run_negotiate(R, n, seed, horizon, episodes):
    import environments.two_stage_train as tst
    tst.ppo = install_stub_trainer()
    np.random.seed(seed)
    random.seed(seed)
    base = R.CleanupEnv(num_agents=n, horizon=horizon)
    con = R.contract_list.CleanupContract(n)
    env = tst.SeparateContractNegotiateStage(base, con, n, horizon, {"n_act": 8, "seed": seed + 7}, "stub-env", "stub-path", True, False)
    keys = ["a%d" % i for i in range(n)]
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "obs_sha", "contract_obs", "rew", "done", "fp", "accepted", "proposed", "trainer_calls")}
    out = {"kind": "cleanup", "n": n, "seed": seed, "horizon": horizon, "action_low": env.action_space.low, "action_high": env.action_space.high}
    for ep in range(episodes):
        o = env.reset()
        for stage in range(2):
            a = np.concatenate([ars.uniform(0, 0.2, size=(n, 1)), ars.uniform(0.5, 1.0, size=(n, 1))], axis=1)
            o, r, d, info = env.step({k: a[i] for i, k in enumerate(keys)})
            sha, c = pack_obs(o, keys)
            rec["actions"].append(a)
            rec["obs_sha"].append(sha)
            rec["contract_obs"].append(c)
            rec["rew"].append([float(r[k]) for k in keys])
            rec["done"].append(np.uint8(d["__all__"]))
            rec["fp"].append(fps())
        rec["accepted"].append(int(env.metrics["accepted"]))
        rec["proposed"].append(np.asarray(env.metrics["contract"], np.float64))
        rec["trainer_calls"].append(len(env.frozen_trainer.calls))
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def run_solver(R, n, seed, horizon, episodes, rule, samples, steps):
    import environments.two_stage_train as tst
    tst.ppo = install_stub_trainer()
    np.random.seed(seed)
    random.seed(seed)
    base = R.CleanupEnv(num_agents=n, horizon=horizon)
    con = R.contract_list.CleanupContract(n)
    env = tst.NegotiationSolver(base, con, n, horizon, {"n_act": 8, "seed": seed + 7}, "stub-env", "stub-path", True, False,
                                contract_samples=samples, decision_rule=rule)
    env.contract_param_space.seed(seed + 11)  # gym spaces sample from their own np_random: seeded, or not reproducible
    keys = ["a%d" % i for i in range(n)]
    ars = np.random.RandomState(seed + 1)
    rec = {k: [] for k in ("actions", "obs_sha", "contract_obs", "rew", "done", "fp", "chosen", "reset_sha", "reset_contract")}
    out = {"kind": "cleanup", "n": n, "seed": seed, "horizon": horizon, "rule": rule, "samples": samples, "steps": steps}
    for ep in range(episodes):
        o = env.reset()
        sha, c = pack_obs(o, keys)
        rec["reset_sha"].append(sha)
        rec["reset_contract"].append(c)
        rec["chosen"].append(np.asarray(env.contract_param, np.float64))
        for t in range(steps):
            a = ars.randint(0, 8, size=n)
            o, r, d, info = env.step({k: int(a[i]) for i, k in enumerate(keys)})
            sha, c = pack_obs(o, keys)
            rec["actions"].append(a)
            rec["obs_sha"].append(sha)
            rec["contract_obs"].append(c)
            rec["rew"].append([float(r[k]) for k in keys])
            rec["done"].append(np.uint8(d["__all__"]))
            rec["fp"].append(fps())
    for k, v in rec.items():
        out[k] = np.array(v)
    return out


def main():
    R = load_reference()
    for name, out in (("stage_combined_cleanup_n4", run_combined(R, 4, 75001, 30, 100)),
                      ("stage_negotiate_cleanup_n4", run_negotiate(R, 4, 75002, 25, 4)),
                      ("stage_negotiate_cleanup_n2", run_negotiate(R, 2, 75003, 20, 3)),
                      ("stage_solver_majority_cleanup_n4", run_solver(R, 4, 75004, 40, 3, "majority", 12, 40)),
                      ("stage_solver_max_cleanup_n3", run_solver(R, 3, 75005, 30, 4, "max", 25, 30))):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-30s %6.1f KB  steps %d" % (name, os.path.getsize(path) / 1024, len(out["actions"])))


if __name__ == "__main__":
    main()
