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
        "--min-seconds", "0.05", "--full", "--full-out", ""]


def _lines(out):
    """--full: the full record, then — LAST — the compact record the driver parses (round 6: one stdout line by default)"""
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2, out.stdout  # only rank 0 reports
    full, compact = json.loads(lines[0]), json.loads(lines[1])
    assert len(lines[1]) < 4096 and compact["value"] == full["value"] and compact["ms_per_step"] == full["ms_per_step"]
    assert compact["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4) and compact["summary"]["C4"]["G"] == full["summary"]["C4"]["G"]
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert compact[k] == full[k], k
    assert compact["config"]["workload"] == full["config"]["workload"] and compact["ranks_seen"] == full["ranks"]["ranks_seen"]
    return full, compact, lines[0]


def _check(out, world=2, envs=2048):
    assert out.returncode == 0, out.stderr[-2000:]
    row, compact, _ = _lines(out)
    assert "cpu_baseline" not in compact or compact["cpu_baseline"] is None
    assert compact["summary"]["parity_all_ok"] is None  # no parity leg ran (world > 1 / --no-cpu-baseline): not a pass
    assert row["n_gpus"] == world and row["steps"] == 60 and row["scaling"] == "weak" and row["value"] > 0
    assert row["config"]["global_envs"] == world * envs and row["config"]["parallelism"].startswith("env-shard x%d" % world)
    assert row["repeats"] >= 3 and row["value_min"] <= row["value"] <= row["value_max"]
    assert "cpu_baseline" not in row
    # the job's width as the ranks themselves counted it (an all-reduce SUM of 1), and the per-rank clocks behind the MAX
    rk = row["ranks"]
    assert rk["ranks_seen"] == world == rk["world_size"] and rk["envs_per_rank"] == envs
    assert 0 < rk["ms_per_step_rank_min"] <= rk["ms_per_step_rank_max"] and abs(rk["ms_per_step_rank_max"] - row["ms_per_step"]) < 1e-9
    # the roofline figure follows from the wall clock of the line itself
    rf = row["roofline"]
    assert abs(rf["frac"] * rf["peak"] * 1e9 * row["ms_per_step"] * 1e-3 - 7235 * envs) < 0.01 * 7235 * envs
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
def test_bench_one_rank_over_rccl():
    """the driver's launcher with ONE rank and the process group forced on (backend nccl = RCCL): bench.py's own barrier, MAX /
    MIN over the elapsed times and the ranks_seen SUM run through RCCL on device tensors — the calls an N-GPU run makes, on the
    one communicator size a single-GPU box can hold (two RCCL ranks cannot share a device)"""
    env = dict(os.environ, CONTRACTS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CONTRACTS_BENCH_BACKEND", None)
    env.pop("CONTRACTS_BENCH_SHARE_GPU", None)
    args = [a if a != "2" else "1" for a in ARGS]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    row, compact, _ = _lines(out)
    assert row["n_gpus"] == 1 and row["ranks"]["ranks_seen"] == 1 and row["ranks"]["world_size"] == 1 and row["value"] > 0
    assert row["config"]["global_envs"] == 2048 and compact["ranks_seen"] == 1 and row["ranks"]["backend"] == "nccl"


@pytest.mark.gpu
def test_bench_eight_ranks_share_one_gpu():
    """the launch the driver uses for the 8-GPU scaling point (`torch.distributed.run --nproc-per-node 8 bench.py --gpus
    8`): eight rank processes, rendezvous, shard bases g * E, barrier + MAX over ranks, one line from rank 0 — here all on
    GPU 0 with a small shard each (functional only)"""
    env = dict(os.environ, CONTRACTS_BENCH_BACKEND="gloo", CONTRACTS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [a if a != "2" else "8" for a in ARGS]
    args[args.index("--envs-per-gpu") + 1] = "512"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py")] + args
    _check(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900), world=8, envs=512)


@pytest.mark.gpu
def test_bench_default_line_has_every_config():
    """one GPU, short: the line carries the headline, the fused mode, every BASELINE config and the CPU baseline"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.05",
           "--cpu-seconds", "1.0", "--config-steps", "48", "--config-seconds", "0.05", "--config-cpu-seconds", "0.5", "--full",
           "--full-out", ""]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    row, compact, raw = _lines(out)
    assert compact["cpu_baseline"]["kind"] == "port" and compact["cpu_baseline"]["cores"] >= 1 and "sample" in compact["cpu_baseline"]
    assert compact["parity_in_run"]["ok"] is True and compact["parity_in_run"]["slices"] == 3  # on the timed row's launch shape
    assert compact["summary"]["parity_all_ok"] is True and compact["summary"]["parity_legs_run"] == 6
    assert row["metric"] == "agent-steps/sec" and row["n_gpus"] == 1 and row["dtype"] == "u8"
    assert [c["config"] for c in row["configs"]] == ["C2", "C3", "C5", "C1"]
    for c in row["configs"]:
        assert c["value"] > 0 and 0 < c["roofline"]["frac"] < 1 and c["steps"] == 48
        assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] >= 1
        # fused rows time whole launches and report the launches actually issued
        assert c["fused"]["steps"] % c["fused"]["steps_per_launch"] == 0
        assert abs(c["fused"]["launches_per_step"] * c["fused"]["steps_per_launch"] - 3) < 1e-9
    assert row["cpu_baseline"]["kind"] == "port" and row["cpu_baseline"]["value"] > 0
    # BASELINE.md's Python-reference constant beside every CPU baseline
    assert row["cpu_baseline"]["python_reference_per_core"] == 4259
    assert [c["cpu_baseline"]["python_reference_per_core"] for c in row["configs"]] == [4203, 1567, 29578, 1846]
    assert row["ranks"]["ranks_seen"] == 1 and "Infinity Cache" in row["roofline"]["note"]
    # --steps 20 with 16-step fused launches: 32 timed steps, two launches per slice
    assert row["fused"]["steps"] == 32 and row["fused"]["steps_per_launch"] == 16
    assert abs(row["fused"]["launches_per_step"] - 3 / 16.0) < 1e-9
    cl = row["closed_loop"]
    assert cl["value"] > 0 and cl["host_iterations_per_step"] == 1 and cl["timed_seconds"] >= 0.5 and "eager" in cl["modes"]
    assert {"policy_eager", "policy_graph"} <= set(cl["modes"]) and cl["modes"]["policy_eager"]["host_calls_per_step"] == 6
    bd = row["boundary"]
    assert "error" not in bd, bd
    assert bd["tensor_path"]["value"] > bd["dict_protocol"]["value"] > bd["dict_protocol_rebuilt"]["value"] > 0
    # the joint baseline through the batched hook (round 6): JointEnv semantics over one handle, both observation modes
    jt = bd["joint"]
    assert "error" not in jt and jt["envs"] == 16384 and jt["global"]["tensor_value"] > jt["global"]["dict_value"] > 0
    assert jt["concatenated"]["tensor_value"] > 0 and compact["summary"]["joint_global_tensor_G"] == round(jt["global"]["tensor_value"] / 1e9, 4)
    # the counter-RNG rows sit beside the line, never in it: the headline's rng is the reference's
    assert row["config"]["rng"] == "mt19937-numpy-compat"
    cr = row["counter_rng"]
    assert "error" not in cr, cr
    assert cr["headline_batch"]["counter"]["value"] > 0 and cr["headline_batch"]["counter"]["fused"]["steps_per_launch"] == 16
    big = cr["large_batch"]
    assert big["envs_per_gpu"] == 262144 and big["counter"]["value"] > big["mt19937"]["value"] > 0  # 16 B vs 2.5 KB of state per env
    assert cr["closed_loop"]["value"] > 0
    rf = row["roofline"]
    assert abs(rf["frac"] * 8000e9 * row["ms_per_step"] * 1e-3 - 7235 * 16384) < 0.01 * 7235 * 16384
    # round 5: the line checks its own kernels against the oracle (headline + every config row), says where its PMC constant
    # comes from, and ends in a summary that holds every row
    assert list(row)[-1] == "summary" and raw.rstrip().endswith("}}")
    sm = row["summary"]
    assert sm["parity_all_ok"] is True and set(sm) >= {"C4", "C2", "C3", "C5", "C1", "closed_loop_G", "closed_loop_inkernel_G", "dict_M", "tensor_G", "counter_G", "beyond_cache_frac"}
    assert len(json.dumps(sm)) < 2048
    for r in [row] + row["configs"]:
        par = r["parity_in_run"]
        assert par["ok"] is True and par["envs"] >= 256 and par["steps"] >= 32 and "mismatches" not in par, par
        assert par["slices"] == 3 and par["handle_envs"] == r.get("envs_per_gpu", 16384)  # the timed row's own batch and slicing
        assert par["per_step_steps"] + par["fused_steps"] == par["steps"] and par["fused_steps"] % 16 == 0 and par["fused_steps"] > 0
        assert {"agents", "rng", "reward", "done"} <= set(par["fields"]) or {"sd_state", "rng", "obs_f64"} <= set(par["fields"])
        assert "_oracle_state" not in r["cpu_baseline"]
        key = r["config"] if isinstance(r["config"], str) else "C4"  # (the headline's `config` is its workload description)
        assert abs(sm[key]["G"] - round(r["value"] / 1e9, 4)) < 1e-9 and sm[key]["parity_ok"] is True
    cpar = cr["parity_in_run"]  # the counter-mode kernels against the oracle's restatement of that stream, in the run
    assert cpar["ok"] is True and cpar["envs"] == 2048 and cpar["steps"] == 1152 and sm["counter_parity_ok"] is True
    # roofline.traffic of the headline is measured IN the run (two rocprofv3 --pmc child passes); the committed constant stays beside it
    assert rf["traffic_live"] in (True, False), rf
    if rf["traffic_live"]:
        assert "measured in this run" in rf["traffic_source"] and 1.0 < rf["traffic_ratio"] < 3.0 and compact["roofline"]["traffic_live"] is True
        assert rf["traffic_committed"] and abs(rf["traffic"] / rf["traffic_committed"] - 1.0) < 0.15  # the two agree within the run-to-run spread
    else:
        assert rf["traffic_live_error"]
    assert rf["traffic_committed_source"] is None or "profiles/traffic.json @" in rf["traffic_committed_source"]
    assert all("roofline_frac_is" in c["fused"] for c in row["configs"])
    # closed loop: the benchmark policy inside the step kernel (one launch per slice and tick), the launch loop in C (one host
    # call per tick), and the best row whose policy is a kernel of its own
    assert {"inkernel_eager", "inkernel_sliced", "policy_graph_all", "inkernel_graph_all"} <= set(cl["modes"])
    assert cl["modes"]["inkernel_sliced"]["host_calls_per_step"] == 1 and cl["modes"]["inkernel_eager"]["launches_per_slice_and_step"] == 1
    assert cl["best_with_separate_policy_kernel"]["value"] > 0 and not cl["best_with_separate_policy_kernel"]["issue"].startswith("inkernel")
    # `value` is the best row whose policy is a kernel of its own; the toy policy inside the step kernel is reported beside it
    assert not cl["issue"].startswith("inkernel") and cl["value"] == cl["best_with_separate_policy_kernel"]["value"] or cl["slices_sweep"]
    assert cl["inkernel_policy"]["issue"].startswith("inkernel") and sm["closed_loop_inkernel_G"] == round(cl["inkernel_policy"]["value"] / 1e9, 4)
    dp = bd["dict_protocol"]
    assert dp["recycle_dicts"] is True and dp["value"] > dp["value_incl_action_dicts"] > dp["value_with_consumer_copies"] > 0


@pytest.mark.gpu
def test_bench_time_budget_keeps_the_required_parts():
    """--time-budget: on a box slow enough to exhaust it (here: a budget of one second) the optional sections are skipped and
    named, the line still carries the headline, its roofline, the CPU baseline and the parity check of the timed kernels"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--min-seconds", "0.05", "--cpu-seconds", "1.0",
           "--time-budget", "1", "--no-live-traffic", "--full", "--full-out", ""]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    row, compact, _ = _lines(out)
    assert row["configs"] == [] and not ({"closed_loop", "boundary", "counter_rng"} & set(row))
    sk = row["time_budget"]["skipped_sections"]
    assert sk == ["configs.C2", "configs.C3", "configs.C5", "configs.C1", "closed_loop", "boundary", "counter_rng"], sk
    assert compact["summary"]["skipped_for_time"] == sk and row["time_budget"]["seconds"] == 1.0
    assert compact["value"] > 0 and 0 < compact["roofline"]["frac"] < 1 and compact["cpu_baseline"]["value"] > 0
    assert compact["parity_in_run"]["ok"] is True and compact["summary"]["parity_legs_run"] == 1 and row["fused"]["value"] > 0
