"""GPU: the RCCL leg of contracts_amd.parallel.Group — the only cross-rank traffic of `bench.py --gpus N` (a barrier and MAX /
MIN / SUM over float64 scalars, backend "nccl" = RCCL on ROCm).  A single-GPU box cannot hold two RCCL ranks (one communicator
rank per device), so this runs the same calls on a ONE-rank communicator in a fresh process: the backend loads, the process group
initialises with `device_id`, float64 all-reduces with the three operators and the barrier go through RCCL, the group is torn
down.  The multi-rank logic above it (shards, slowest-rank timing, report) is covered with gloo (tests/test_parallel_gloo.py,
tests/test_bench_multirank_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
import torch
from contracts_amd import parallel
torch.cuda.set_device(0)
g = parallel.Group("nccl", "cuda:0", force=True)
assert g.world == 1 and g._dist is not None and g._dist.get_backend() == "nccl"
g.barrier()
assert g.max(3.25) == 3.25 and g.min(-7.5) == -7.5 and g.sum(1e-3) == 1e-3
assert parallel.max_over_ranks(2.0, "cuda:0") == 2.0 and parallel.sum_over_ranks(4.0, "cuda:0") == 4.0
base, cnt = g.shard(16384)
assert (base, cnt) == (0, 16384)
g.barrier()
g.close()
print("rccl one-rank group ok")
"""


@pytest.mark.gpu
def test_group_over_rccl_one_rank():
    env = dict(os.environ)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "rccl one-rank group ok" in out.stdout


def _free_port():
    sys.path.insert(0, ROOT)
    from contracts_amd import parallel
    return parallel.free_port()
