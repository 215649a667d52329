"""Multi-GPU layout of the hot path: env replicas are independent, so N GPUs = a plain shard of the
env axis — one process per GPU, rank g owns the global env indices [g*E, (g+1)*E), no collective on the
data path.  Seeds are keyed by the GLOBAL env index, so results do not depend on the
number of GPUs.  The only cross-rank traffic is the benchmark's barrier and a MAX over elapsed times."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def env_shard(rank, world, envs_per_rank):
    """(env_index_base, num_envs) of this rank under weak scaling (fixed envs per GPU)"""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return rank * envs_per_rank, envs_per_rank


def split_envs(total_envs, world):
    """strong-scaling split of a fixed global batch: contiguous slices, remainder to the first ranks"""
    q, r = divmod(total_envs, world)
    out, base = [], 0
    for g in range(world):
        cnt = q + (1 if g < r else 0)
        out.append((base, cnt))
        base += cnt
    return out


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over the process group (identity when not initialised)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
