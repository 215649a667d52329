"""bench.py's multi-rank path (env-axis shard by global index, barrier, MAX over ranks, rank-0 report) on a single-GPU
box: both ranks share GPU 0 and rendezvous over gloo (functional check only; the numbers of such a run mean nothing).
Two launch modes: the way the driver launches it (`python -m torch.distributed.run --nproc-per-node N`) and bench.py's
own launcher (`python bench.py --gpus N`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--steps", "60", "--warmup", "10", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--no-configs",
        "--min-seconds", "0.05"]


def _check(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # only rank 0 reports
    row = json.loads(lines[0])
    assert row["n_gpus"] == 2 and row["steps"] == 60 and row["scaling"] == "weak" and row["value"] > 0
    assert row["config"]["global_envs"] == 2 * 2048 and row["config"]["parallelism"].startswith("env-shard x2")
    assert row["repeats"] >= 3 and row["value_min"] <= row["value"] <= row["value_max"]
    assert "cpu_baseline" not in row
    # the roofline figure follows from the wall clock of the line itself
    rf = row["roofline"]
    assert abs(rf["frac"] * rf["peak"] * 1e9 * row["ms_per_step"] * 1e-3 - 7235 * 2048) < 0.01 * 7235 * 2048
    assert row["fused"]["value"] > 0 and row["fused"]["steps_per_launch"] == 16
    return row


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, CONTRACTS_BENCH_BACKEND="gloo", CONTRACTS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py")] + ARGS
    _check(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    env = dict(os.environ, CONTRACTS_BENCH_BACKEND="gloo", CONTRACTS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + ARGS
    _check(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600))


@pytest.mark.gpu
def test_bench_default_line_has_every_config():
    """one GPU, short: the line carries the headline, the fused mode, every BASELINE config and the CPU baseline"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--min-seconds", "0.05",
           "--cpu-seconds", "1.0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    row = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert row["metric"] == "agent-steps/sec" and row["n_gpus"] == 1 and row["dtype"] == "u8"
    assert [c["config"] for c in row["configs"]] == ["C2", "C3", "C5", "C1"]
    for c in row["configs"]:
        assert c["value"] > 0 and 0 < c["roofline"]["frac"] < 1
    assert row["cpu_baseline"]["kind"] == "port" and row["cpu_baseline"]["value"] > 0
    rf = row["roofline"]
    assert abs(rf["frac"] * 8000e9 * row["ms_per_step"] * 1e-3 - 7235 * 16384) < 0.01 * 7235 * 16384
