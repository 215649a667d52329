"""steady-state stepping of the two feature-vector envs (BASELINE config 0, batched) for tools/pmc_feat.sh"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
from contracts_amd.engine import BatchedEnv

for kind, contract in (("harvest_features", "harvest_local"), ("cleanup_features", "cleanup")):
    E, n, T = 8192, 2, 120
    env = BatchedEnv(kind, E, n, contract=contract, auto_reset=True)
    env.seed(seed0=73907)
    env.reset()
    acts = torch.empty((T, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(73908, 0, T, acts.data_ptr())
    env.rollout_device(acts.data_ptr(), T, None)
    torch.cuda.synchronize()
    env.close()
