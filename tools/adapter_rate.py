#!/usr/bin/env python3
"""steps/s of the single-env drop-in adapters (E = 1, host round trip per call, process-global RNG mirrored)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contracts_amd.contract import contract_list as cl
from contracts_amd.environments.cleanup_new import CleanupEnv
from contracts_amd.environments.feature_envs import HarvestFeatures
from contracts_amd.environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage


def rate(name, top, env, keys, act_fn, n, steps=300):
    top.reset()
    for t in range(20):
        o, r, d, i = top.step({k: act_fn(t) for k in keys})
    t0 = time.perf_counter()
    done = 0
    for t in range(20, steps):
        o, r, d, i = top.step({k: act_fn(t) for k in keys})
        if d["__all__"]:
            top.reset()
            done += 1
    dt = time.perf_counter() - t0
    print("%-34s %6.0f env-steps/s = %7.0f agent-steps/s (%.2f ms per step() call, %d resets)" % (
        name, (steps - 20) / dt, (steps - 20) * n / dt, dt / (steps - 20) * 1e3, done))
    env.close()


rs = np.random.RandomState(0)
for rng in ("global", "private"):
    np.random.seed(1)
    random.seed(1)
    env = CleanupEnv(num_agents=8, rng=rng)
    rate("cleanup_new n=8 + contract (%s)" % rng, SeparateContractSubgameStage(env, cl.CleanupContract(8), 8, True), env,
         ["a%d" % i for i in range(8)], lambda t: int(rs.randint(8)), 8)
np.random.seed(1)
random.seed(1)
env = HarvestFeatures(num_agents=2)
rate("harvest (features) n=2 + contract", SeparateContractSubgameStage(env, cl.HarvestFeaturemodLocalContract(2), 2, False), env,
     ["a0", "a1"], lambda t: int(rs.randint(7)), 2)
np.random.seed(1)
random.seed(1)
env = SelfAcceleratingCarEnv(num_agents=4)
top = SeparateContractSubgameStage(env, cl.SelfdriveContractDistprop(4), 4, False)
top.reset()
t0, steps = time.perf_counter(), 0
alive = ["a%d" % i for i in range(4)]
for t in range(300):
    o, r, d, i = top.step({k: np.array([0.05]) for k in alive})
    steps += 1
    alive = [k for k in alive if not d[k]]
    if d["__all__"]:
        top.reset()
        alive = ["a%d" % i for i in range(4)]
dt = time.perf_counter() - t0
print("%-34s %6.0f env-steps/s (%.2f ms per step() call)" % ("selfdrive n=4 + Distprop", steps / dt, dt / steps * 1e3))
env.close()
