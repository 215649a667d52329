"""Debug: first obs pixel where the engine and the oracle differ on a fuzz fixture."""
import sys
import numpy as np
sys.path[:0] = [".", "tests"]
import golden_check as gc
from contracts_amd.engine import BatchedEnv
from oracle.pyoracle import Oracle
name = sys.argv[1] if len(sys.argv) > 1 else "fuzz_harvest_n3"
g = gc.load(name)
kind, n, S = str(g["kind"]), int(g["n"]), len(g["seed"])
kw = dict(firing=bool(int(g["firing"])))
env, orc = BatchedEnv(kind, S, n, **kw), Oracle(kind, S, n, **kw)
agents = np.zeros((S, n, 4), np.uint8); agents[:, :, :3] = g["in_agents"]
for impl in (env, orc):
    impl.seed(g["seed"].astype(np.uint64), replay_constructor=False)
for f, v in (("grid", g["in_grid"]), ("agents", agents), ("timestep", np.full((S,), 5, np.int32))):
    env.upload(f, v); getattr(orc, f)[...] = v
if kind == "cleanup":
    env.upload("waste_perm", g["in_waste_perm"]); orc.waste_perm[...] = g["in_waste_perm"]
orc.import_state()
env.step(g["actions"]); orc.step(g["actions"])
x, y = env.download("obs"), orc.obs
bad = np.argwhere(x != y)
print("differing bytes", len(bad))
for s in np.unique(bad[:, 0])[:3]:
    b = bad[bad[:, 0] == s]
    print("scenario", s, "agents (row,col,orient)", orc.agents[s][:, :3].tolist())
    for e in b[:6]:
        print("  viewer", e[1], "pixel", e[2], e[3], "ch", e[4], "engine", x[tuple(e)], "oracle", y[tuple(e)])
