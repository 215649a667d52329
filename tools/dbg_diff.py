"""Debug: first field / env / step where the engine and the oracle part ways (cleanup, random actions)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from contracts_amd.engine import BatchedEnv
from oracle.pyoracle import Oracle
kind, n, E, T = (sys.argv[1] if len(sys.argv) > 1 else "cleanup"), 4, 64, 120
kw = dict(contract="cleanup" if kind == "cleanup" else "harvest_local", horizon=1000, auto_reset=True)
env, orc = BatchedEnv(kind, E, n, **kw), Oracle(kind, E, n, **kw)
seeds = np.arange(E).astype(np.uint64) + 5
for o in (env, orc):
    o.seed(seeds); o.reset()
rs = np.random.RandomState(3)
fields = ["rng", "waste_perm", "grid", "agents", "reward", "obs"]
for t in range(T):
    a = rs.randint(env.num_actions, size=(E, n)).astype(np.uint8)
    pre_pos = orc.rng[:, 624].copy(); pre_grid = orc.grid.copy()
    env.step(a); orc.step(a)
    for f in fields:
        x, y = env.download(f), getattr(orc, f)
        if f == "rng":
            x, y = x[:, :625], y[:, :625]
        if not np.array_equal(x, y):
            bad = np.nonzero((x != y).reshape(E, -1).any(axis=1))[0]
            e = bad[0]
            print("step", t, "field", f, "envs", bad[:8], "pre-step pos of env", e, "=", pre_pos[e])
            if f == "rng":
                print(" pos engine/oracle", x[e, 624], y[e, 624], "state words differ:", int((x[e, :624] != y[e, :624]).sum()))
            if f == "grid":
                print(' pre-step: wastes', int((pre_grid[e]==3).sum()), 'apples', int((pre_grid[e]==2).sum()), 'post oracle apples', int((y[e]==2).sum()), 'post engine apples', int((x[e]==2).sum()), 'actions', a[e].tolist(), 'post pos', int(orc.rng[e,624]))
                d = np.argwhere(x[e] != y[e]); print(" cells", d[:6].tolist(), "engine", [int(x[e][tuple(c)]) for c in d[:6]], "oracle", [int(y[e][tuple(c)]) for c in d[:6]])
            sys.exit(0)
print("no difference in", T, "steps")
