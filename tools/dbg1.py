import sys; sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, random
import golden_check as gc
from contracts_amd.engine import BatchedEnv
g = gc.load("g2_harvest_n8"); seed=int(g["seed"])
print("expected ctor", g["ctor_mt"], "reset", g["reset_mt"][0])
for E in (1,3):
    env = BatchedEnv("harvest", E, 8, contract="harvest_local")
    env.seed(np.full(E, seed, np.uint64)); print("E",E,"after seed+construct pos", env.download("rng")[:,624])
    env.reset(); print("after reset pos", env.download("rng")[:,624], "agents", env.download("agents")[0,:,:3].tolist()==g["reset_agents"][0].tolist())
    env.close()
# adapter-like: construct-only from uploaded state
np.random.seed(seed)
env = BatchedEnv("harvest", 1, 8)
st = np.random.get_state(legacy=True); w = np.zeros((1,628),np.uint32); w[0,:624]=st[1]; w[0,624]=st[2]
env.upload("rng", w); env.construct(); print("construct-only pos", env.download("rng")[:,624], "spawn", env.download("spawn_perm")[0].tolist())
env.set_contract("harvest_local", 0.0, 10.0, 0.0)
env.reset(); print("reset pos", env.download("rng")[:,624], env.download("agents")[0,:,:3].tolist()==g["reset_agents"][0].tolist(), env.download("error_flags"))
