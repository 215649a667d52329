"""bench.py's multi-rank path (env-axis shard by global index, barrier, MAX over ranks, rank-0 report) launched the
way the driver launches it — `python -m torch.distributed.run --nproc-per-node N` — on a single-GPU box: both ranks
share GPU 0 and rendezvous over gloo (functional check only; the numbers of such a run mean nothing)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, CONTRACTS_BENCH_BACKEND="gloo", CONTRACTS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "10",
           "--envs-per-gpu", "2048", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # only rank 0 reports
    row = json.loads(lines[0])
    assert row["n_gpus"] == 2 and row["steps"] == 60 and row["scaling"] == "weak" and row["value"] > 0
    assert row["config"]["global_envs"] == 2 * 2048 and row["config"]["parallelism"].startswith("env-shard x2")
    assert "cpu_baseline" not in row
