#!/usr/bin/env python3
"""bench.py — headline benchmark of the rollout hot path on MI355X.

Workload (BASELINE.json north star / configs[3]): CleanupEnv `cleanup_new`, 8 agents, CleanupContract,
16384 env replicas per GPU (131072 over 8 GPUs; weak scaling, plain shard of the env axis, no
collectives), horizon 1000 with in-engine auto-reset, uniform i.i.d. synthetic actions generated on
device by the counter hash keyed (seed, global env index, t, agent) and resident in HBM before the
timed region.  A "step" = one pass of the hot path over the whole env batch of this rank (MapEnv.step +
obs crop + contract transfer for every env), issued as `--streams` (default 3) ce_step_range launches over
contiguous env slices on separate HIP streams: slices are independent, so one slice's kernel tail overlaps
the other's head (the same double buffering an RL sampler uses to overlap policy inference with stepping).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 16384
N_AGENTS = 8
SEED0 = 73907  # the reference's seed multiplier (runner.py:130)
# algorithmic bytes per env-step, cleanup n=8 (state read+written once, uint8 obs)
ALGO_BYTES_PER_ENV_STEP = 7235
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--agents", type=int, default=N_AGENTS)
    ap.add_argument("--kind", default="cleanup", choices=["cleanup", "harvest"])
    ap.add_argument("--streams", type=int, default=3,
                    help="split each rank's env batch into this many contiguous slices stepped on separate HIP streams "
                         "(slices are independent; one slice's kernel tail overlaps the next slice's head)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """cores this process may actually use: scheduler affinity capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(kind, n, contract, target_s):
    """The CPU oracle (oracle/oracle.c, pinned bit-exact to the reference's golden traces) timed on this
    box's host cores with OpenMP over envs: a bounded sample of the same workload, at all threads (the
    reported value) and at one thread."""
    import ctypes
    from oracle.pyoracle import Oracle
    threads = _usable_cores()
    gomp = ctypes.CDLL("libgomp.so.1")
    rs = np.random.RandomState(1)
    na = 8 if kind == "cleanup" else 7

    def run(nthreads, E, seconds):
        gomp.omp_set_num_threads(nthreads)
        orc = Oracle(kind, E, n, contract=contract, horizon=1000, auto_reset=True)
        orc.seed(seed0=SEED0)
        orc.reset()
        acts = rs.randint(na, size=(32, E, n)).astype(np.uint8)
        for t in range(8):
            orc.step(acts[t % 32])
        steps, t0 = 0, time.perf_counter()
        while True:
            for _ in range(8):
                orc.step(acts[steps % 32])
                steps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
        orc.close()
        return E * n * steps / dt, steps, dt

    v_all, steps, dt = run(threads, 256 * threads, target_s)
    v_one, steps1, dt1 = run(1, 256, min(4.0, target_s))
    return {"value": v_all, "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "single_thread_value": v_one, "cpu_model": _cpu_model(),
            "python_reference_per_core": 4259,  # the reference itself, measured by the survey
            "sample": "%d envs x %d steps, %s n=%d + contract, auto-reset, OpenMP over envs, %.1f s (+ %d envs x %d steps on "
                      "1 thread, %.1f s)" % (256 * threads, steps, kind, n, dt, 256, steps1, dt1)}


def stream_ceiling():
    """measured streaming ceilings of this box: device-to-device copy (read + write bytes) and fill (write
    only) of a 1 GiB buffer through torch's own kernels, GB/s"""
    import torch
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    src.fill_(1)
    out = {}
    for name, fn, moved in (("copy", lambda: dst.copy_(src), 2 * nbytes), ("fill", lambda: dst.fill_(3), nbytes)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = moved * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(a.gpus, 1):
        if rank == 0 and world == 1 and a.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (a.gpus, a.gpus), file=sys.stderr)
            sys.exit(2)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # functional-test knobs (tests/test_bench_multirank_gpu.py): several ranks on ONE GPU with a gloo rendezvous —
    # exercises the sharding / reduction / reporting path where only a single-GPU box is available; never for numbers
    backend = os.environ.get("CONTRACTS_BENCH_BACKEND", "nccl")
    if os.environ.get("CONTRACTS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from contracts_amd.engine import BatchedEnv
    kind, n, E = a.kind, a.agents, a.envs_per_gpu
    contract = "cleanup" if kind == "cleanup" else "harvest_local"
    env = BatchedEnv(kind, E, n, contract=contract, horizon=1000, auto_reset=True, device=local_rank,
                     env_index_base=rank * E)
    env.seed(seed0=SEED0)  # env b (global index) is seeded SEED0 + b: results independent of GPU count
    env.reset()
    K, W = a.steps, a.warmup
    # synthetic inputs: all K+W action planes resident in HBM before timing
    acts = torch.empty((W + K, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(SEED0 + 1, 0, W + K, acts.data_ptr())
    env.synchronize()
    plane = E * n
    base = acts.data_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = max(1, a.streams)
    bounds = [E * i // S for i in range(S + 1)]
    slices = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(S)]
    if S == 1:
        streams = [None]
        handles = [None]
    else:
        streams = [torch.cuda.Stream() for _ in range(S)]
        handles = [st.cuda_stream for st in streams]

    def run_steps(t_begin, count):
        # one launch per (step, slice); the loop itself runs in C (ce_rollout) to keep the host off the path
        env.rollout_device(base + t_begin * plane, count, None if S == 1 else handles)

    run_steps(0, W)
    barrier()
    # per-launch kernel time: HIP events on the stream(s) the kernels are launched on
    if S == 1:
        env.timing_begin()
    else:
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(S)]
        for e0, st in zip(ev0, streams):
            e0.record(st)
    t0 = time.perf_counter()
    run_steps(W, K)
    if S > 1:
        for e1, st in zip(ev1, streams):
            e1.record(st)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if S == 1:
        kernel_ms, launches = env.timing_end()
    else:
        kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in zip(ev0, ev1)])) / K
        launches = K * S
    barrier()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.check_faults()
    # sanity statistics of the simulated workload (not timed)
    mi = env.download("int_metrics")
    stats = {"apples_eaten_running_episode_mean": float(mi[:, 0].mean()), "dirt_cleaned_mean": float(mi[:, 2].mean())}

    if rank == 0:
        total_agent_steps = world * E * n * K
        value = total_agent_steps / elapsed
        # one launch covers E/S envs; S launches (one per stream) run concurrently
        algo_bytes_launch = ALGO_BYTES_PER_ENV_STEP * E // S if (kind == "cleanup" and n == 8) else None
        roof = None
        if algo_bytes_launch:
            achieved = S * algo_bytes_launch / (kernel_ms * 1e-3) / 1e9  # S concurrent launches share the chip
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))  # PMC bytes per env-step (traffic is linear in the envs of a launch)
                    if tj.get("agents") == n and tj.get("kind") == kind:
                        traffic = int(round(tj["hbm_bytes_per_env_step"] * (E // S)))
                except Exception:
                    traffic = None
            ceil = stream_ceiling()
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "measured_copy_GBs": ceil["copy"], "measured_fill_GBs": ceil["fill"],
                    "kernel": "k_grid_step<%s>" % kind, "kernel_ms": kernel_ms, "launches": launches,
                    "algorithmic_bytes_per_launch": algo_bytes_launch, "concurrent_streams": S}
        out = {
            "metric": "agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "cleanup_new 8 agents + CleanupContract, %d envs/GPU, horizon 1000, auto-reset" % E
                       if kind == "cleanup" else "harvest_new %d agents + HarvestFeaturemodLocalContract, %d envs/GPU" % (n, E),
                       "envs_per_gpu": E, "envs_per_launch": E // S, "agents": n, "global_envs": world * E, "rng": "mt19937-numpy-compat",
                       "parallelism": "env-shard x%d, no collectives" % world, "streams_per_gpu": S, "sanity": stats},
            "roofline": roof,
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kind, n, contract, a.cpu_seconds)
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
