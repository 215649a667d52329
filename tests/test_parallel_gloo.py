"""CPU, world_size 2 over gloo: the multi-GPU layout of the hot path (plain shard of the env axis, seeds
keyed by the GLOBAL env index, no data-path collective; MAX over ranks for the timing).  The CPU oracle
stands in for the engine so the test runs without GPUs: two ranks stepping their shards must reproduce
exactly what one rank stepping the whole batch produces."""
import hashlib
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from contracts_amd.parallel import env_shard, max_over_ranks, sum_over_ranks


def _rollout(kind, n, base, count, steps):
    from oracle.pyoracle import Oracle
    orc = Oracle(kind, count, n, contract="cleanup", horizon=30, auto_reset=True, env_index_base=base)
    orc.seed(seed0=73907)  # env b gets 73907 + env_index_base + b
    orc.reset()
    digests = []
    for t in range(steps):
        # actions keyed by (global env index, t, agent) exactly like the engine's counter hash would be
        g = np.arange(base, base + count)[:, None]
        a = ((g * 31 + t * 17 + np.arange(n)[None, :] * 7) % 8).astype(np.uint8)
        orc.step(a)
    for e in range(count):
        digests.append(hashlib.sha256(orc.grid[e].tobytes() + orc.obs[e].tobytes() + orc.reward[e].tobytes()).hexdigest())
    return digests


def _worker(rank, world, port, per_rank, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, count = env_shard(rank, world, per_rank)
    d = _rollout("cleanup", 4, base, count, 45)
    elapsed = 1.0 + rank  # rank 1 is "slower"
    mx = max_over_ranks(elapsed)
    tot = sum_over_ranks(count)
    dist.barrier()
    out[rank] = (base, d, mx, tot)
    dist.destroy_process_group()


def test_two_rank_shard_equals_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    per_rank = 6
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, per_rank, out), nprocs=2, join=True)
    whole = _rollout("cleanup", 4, 0, 2 * per_rank, 45)
    got = [None] * (2 * per_rank)
    for rank in (0, 1):
        base, d, mx, tot = out[rank]
        assert mx == 2.0 and tot == 2 * per_rank
        got[base:base + per_rank] = d
    assert got == whole
