"""CPU, world_size 2 over gloo: the multi-GPU layout of the hot path (plain shard of the env axis, seeds and synthetic
actions keyed by the GLOBAL env index, no data-path collective; MAX over ranks for the timing) through the code that
bench.py runs — contracts_amd.parallel's launcher (spawn_local_ranks), rank bootstrap (Group), shard arithmetic and
reductions.  The CPU oracle stands in for the engine here; tests/test_bench_multirank_gpu.py runs the same rank script
with the HIP engine, and bench.py --gpus 2 itself, on the GPU box."""
import json
import os
import sys

import pytest

from contracts_amd import parallel

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, world, per_rank, engine=False):
    sys.path.insert(0, HERE)
    import _shard_rank
    out = str(tmp_path / "shard")
    argv = ["--out", out, "--per-rank", str(per_rank)] + (["--engine"] if engine else [])
    rc = parallel.spawn_local_ranks(os.path.join(HERE, "_shard_rank.py"), argv, world, timeout=600)
    assert rc == 0
    got = [None] * (world * per_rank)
    for r in range(world):
        res = json.load(open("%s.%d" % (out, r)))
        assert res["world"] == world and res["max"] == float(world) and res["sum"] == world * per_rank
        assert res["base"] == r * per_rank
        got[res["base"]:res["base"] + per_rank] = res["digests"]
    whole = _shard_rank.rollout(engine, "cleanup", 4, 0, world * per_rank, 45)
    assert got == whole  # N ranks stepping their shards == one rank stepping the whole batch, env for env


def test_two_rank_shard_equals_single_rank(tmp_path):
    _run(tmp_path, 2, 6)


@pytest.mark.gpu
def test_two_rank_shard_equals_single_rank_engine(tmp_path):
    """the same, with both ranks stepping the HIP engine (they share GPU 0; gloo rendezvous)"""
    _run(tmp_path, 2, 70, engine=True)


def test_shard_arithmetic():
    assert parallel.env_shard(3, 8, 16384) == (49152, 16384)
    with pytest.raises(ValueError):
        parallel.env_shard(8, 8, 4)
    assert parallel.split_envs(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert parallel.slice_bounds(16384, 3) == [(0, 5461), (5461, 5461), (10922, 5462)]
    assert sum(c for _, c in parallel.slice_bounds(1000, 7)) == 1000


def test_bench_self_launch_starts_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` without torchrun: the parent only parses and spawns (no torch / HIP in it); on this
    GPU-less box the rank processes then refuse loudly — the engine has no CPU path — and the parent reports failure"""
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, CONTRACTS_BENCH_BACKEND="gloo")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stderr.count("needs an MI355X") == 2  # both ranks were started and both refused


def test_launcher_fails_fast_when_one_rank_dies(tmp_path):
    """one rank exits non-zero while its sibling would block for a minute (a barrier its dead peer never reaches): the
    launcher notices within its polling interval, ends the survivor (by PID) and reports the failing exit code"""
    import time
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                      "time.sleep(60)\n")
    t0 = time.monotonic()
    rc = parallel.spawn_local_ranks(str(script), [], 2, timeout=120)
    assert rc == 7 and time.monotonic() - t0 < 40  # (well under the survivor's 60 s; lenient for a loaded build box)


def test_launcher_deadline_is_for_the_whole_job(tmp_path):
    import time
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(60)\n")
    t0 = time.monotonic()
    rc = parallel.spawn_local_ranks(str(script), [], 3, timeout=2.0)
    assert rc == 124 and time.monotonic() - t0 < 40


def test_eight_rank_shard_equals_single_rank(tmp_path):
    """the 8-GPU layout of BASELINE config 3 in miniature: eight ranks over gloo, three envs each, against one rank
    stepping all twenty-four (global-index seeds and actions: the digests must line up env for env)"""
    _run(tmp_path, 8, 3)
