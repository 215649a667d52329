"""Multi-GPU layout of the hot path: env replicas are independent, so N GPUs = a plain shard of the env axis — one
process per GPU, rank g owns the global env indices [g*E, (g+1)*E), no collective on the data path (SURVEY.md §8e;
the reference's own scale-out is one env per Ray worker process, runner.py:51-82).  Seeds are keyed by the GLOBAL env
index, so results do not depend on the number of GPUs.  The only cross-rank traffic is a barrier and a MAX over elapsed
times (RCCL through torch.distributed, backend "nccl"; "gloo" for the CPU tests).

This module is what bench.py runs: the launcher for `bench.py --gpus N` (fresh child processes, started before anything
touches the GPU), the rank bootstrap, the shard arithmetic and the timing reduction.  tests/test_parallel_gloo.py drives
the same functions with world_size 2 over gloo.
"""
import os
import socket
import subprocess
import sys


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched_by_torchrun():
    """true inside a rank process (torch.distributed.run or spawn_local_ranks exported the rendezvous variables)"""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


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


def slice_bounds(num_envs, num_slices):
    """contiguous env slices of one rank's batch, one per HIP stream: [(first, count), ...] (same rule as ce_rollout)"""
    b = [num_envs * i // num_slices for i in range(num_slices + 1)]
    return [(b[i], b[i + 1] - b[i]) for i in range(num_slices)]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_local_ranks(script, argv, nprocs, extra_env=None, timeout=None, poll_s=0.2):
    """`python script argv...` as nprocs fresh processes of one node, rank i with RANK = LOCAL_RANK = i and a 127.0.0.1
    rendezvous — what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` exports.  Must be called before the
    calling process has touched the GPU (children are started with subprocess, never exec'd over a process that holds
    a HIP context).  Children inherit stdout / stderr (rank 0 prints the report).

    Fail fast: all children are polled together; the first non-zero exit, the overall deadline `timeout` (seconds, one
    limit for the whole job, not per rank) or a SIGTERM / SIGINT to this process ends the remaining ranks (exactly the
    processes started here, by PID) instead of leaving them blocked in a barrier their dead sibling will never reach.
    Returns the largest exit code (124 on the deadline, 143 when terminated)."""
    import signal
    import time
    port = free_port()
    procs = []
    for r in range(nprocs):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nprocs), LOCAL_WORLD_SIZE=str(nprocs),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    stop = {"sig": 0}

    def on_signal(signum, frame):
        stop["sig"] = signum

    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:  # not the main thread: no handler, the deadline still applies
            pass
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed:
                rc = max(abs(c) for c in failed)
                break
            if all(c == 0 for c in codes):
                break
            if stop["sig"]:
                rc = 128 + stop["sig"]
                break
            if deadline is not None and time.monotonic() > deadline:
                rc = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:  # exactly the processes started here, by PID
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


class Group:
    """The process group of a run: rank bootstrap, barrier and the reductions the benchmark needs.  world == 1 never
    imports torch.distributed."""

    def __init__(self, backend="nccl", device=None, force=False):
        """force: initialise the process group even at world == 1 (a one-rank communicator) — how the RCCL leg of this class
        is exercised on a single-GPU box (tests/test_parallel_rccl_gpu.py); never set by bench.py"""
        self.rank, self.local_rank, self.world = rank_info()
        self.backend = backend
        self.device = device  # "cuda:k" for nccl, None / "cpu" for gloo
        self._dist = None
        if self.world > 1 or force:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device(device))
            else:
                dist.init_process_group(backend=backend)
            self._dist = dist

    def shard(self, envs_per_rank):
        return env_shard(self.rank, self.world, envs_per_rank)

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def _reduce(self, value, op):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self._dist.all_reduce(t, op=getattr(self._dist.ReduceOp, op))
        return float(t.item())

    def max(self, value):
        """MAX over ranks of a python float (the benchmark's elapsed time)"""
        return self._reduce(value, "MAX")

    def min(self, value):
        return self._reduce(value, "MIN")

    def sum(self, value):
        return self._reduce(value, "SUM")

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over an already initialised default group (identity otherwise)"""
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
