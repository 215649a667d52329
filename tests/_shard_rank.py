"""Rank process of the multi-GPU layout tests: started by contracts_amd.parallel.spawn_local_ranks (the launcher behind
`bench.py --gpus N`), bootstraps a parallel.Group, steps its shard of the env axis and writes per-env digests.
--engine steps the HIP engine (all ranks on GPU 0); without it the CPU oracle stands in (no GPU needed)."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from contracts_amd import parallel, synth  # noqa: E402


def rollout(engine, kind, n, base, count, steps, seed0=73907):
    kw = dict(contract="cleanup", horizon=30, auto_reset=True, env_index_base=base)
    if engine:
        from contracts_amd.engine import BatchedEnv
        env = BatchedEnv(kind, count, n, **kw)
    else:
        from oracle.pyoracle import Oracle
        env = Oracle(kind, count, n, **kw)
    env.seed(seed0=seed0)  # env b gets seed0 + env_index_base + b
    env.reset()
    acts = synth.synth_actions_u8(seed0 + 1, base, count, n, 0, steps, 8)  # keyed by the GLOBAL env index
    for t in range(steps):
        env.step(acts[t])
    grid, obs, rew = (env.download(f) if engine else getattr(env, f) for f in ("grid", "obs", "reward"))
    out = [hashlib.sha256(grid[e].tobytes() + obs[e].tobytes() + np.round(rew[e], 9).tobytes()).hexdigest()
           for e in range(count)]
    env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--per-rank", type=int, default=6)
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--engine", action="store_true")
    a = ap.parse_args()
    group = parallel.Group("gloo")
    base, count = group.shard(a.per_rank)
    digests = rollout(a.engine, "cleanup", 4, base, count, a.steps)
    group.barrier()
    elapsed = 1.0 + group.rank  # the higher rank is "slower": MAX must return world's
    res = {"rank": group.rank, "world": group.world, "base": base, "digests": digests, "max": group.max(elapsed),
           "sum": group.sum(count)}
    group.barrier()
    json.dump(res, open("%s.%d" % (a.out, group.rank), "w"))
    group.close()


if __name__ == "__main__":
    main()
