#!/usr/bin/env python3
"""steps/s of the single-env drop-in adapters (E = 1, host round trip per call, process-global RNG mirrored)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contracts_amd.contract import contract_list as cl
from contracts_amd.environments.cleanup_new import CleanupEnv
from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage

for rng in ("global", "private"):
    np.random.seed(1)
    env = CleanupEnv(num_agents=8, rng=rng) if rng != "global" else CleanupEnv(num_agents=8)
    top = SeparateContractSubgameStage(env, cl.CleanupContract(8), 8, True)
    top.reset()
    rs = np.random.RandomState(0)
    acts = rs.randint(8, size=(300, 8))
    keys = ["a%d" % i for i in range(8)]
    for t in range(20):
        top.step(dict(zip(keys, acts[t].tolist())))
    t0 = time.perf_counter()
    for t in range(20, 300):
        top.step(dict(zip(keys, acts[t].tolist())))
    dt = time.perf_counter() - t0
    print("rng=%s: %.0f env-steps/s = %.0f agent-steps/s (%.2f ms per step() call)" % (rng, 280 / dt, 280 * 8 / dt, dt / 280 * 1e3))
    env.close()
