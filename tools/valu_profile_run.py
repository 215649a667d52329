"""One leg of the per-phase instruction profile: loads a mid-episode state saved by the shipped engine and
steps it with the library named by CONTRACTS_AMD_LIB (a truncated build leaves the state untouched, so
every step repeats the same work).  Usage: python3 tools/valu_profile_run.py make|run STATEFILE"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contracts_amd.engine import BatchedEnv

mode, path = sys.argv[1], sys.argv[2]
E, n = 8192, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", auto_reset=True)
acts = torch.empty((400, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, 400, acts.data_ptr())
if mode == "make":
    env.seed(seed0=73907)
    env.reset()
    env.rollout_device(acts.data_ptr(), 300, None)
    env.synchronize()
    env.save(path)
else:
    env.load(path)
    for t in range(300, 330):
        env.step_device(acts.data_ptr() + t * E * n)
    env.synchronize()
env.close()
