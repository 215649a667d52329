#!/usr/bin/env python3
"""bench.py — headline benchmark of the rollout hot path on MI355X.

Workload (BASELINE.json north star / configs[3]): CleanupEnv `cleanup_new`, 8 agents, CleanupContract, 16384 env
replicas per GPU (131072 over 8 GPUs; weak scaling, plain shard of the env axis, no collectives), horizon 1000 with
in-engine auto-reset, uniform i.i.d. synthetic actions from the counter hash keyed (seed, global env index, t, agent),
resident in HBM before the timed region.  A "step" = one pass of the hot path over the whole env batch of this rank
(MapEnv.step + obs crop + contract transfer for every env).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Protocol per measured mode: a fixed untimed pre-roll (PREROLL steps, so the steady state does not depend on --warmup),
W untimed warm-up steps, then R >= 3 repeats of EXACTLY K steps, each bracketed by barrier + device synchronize on both
sides and reduced with MAX over ranks; the report carries the median repeat.  Two modes are measured on the headline:
  per_step  one launch per env-step and env slice (`--streams` slices on separate HIP streams) — what an RL sampler that
            needs the observation before choosing the next action can use; this is `value`;
  fused     ce_rollout_fused: T consecutive steps per launch with the env state resident on chip, every step's outputs
            written to its plane of a trajectory ring (pre-supplied actions: contract search, random-policy rollouts).
The line also carries, measured in the same run (rank 0's GPU, N = 1 only for the host-side ones):
  configs      the other BASELINE configs (C2, C3, C5, C1), per_step and fused, each timed for >= 0.3 s over repeats of
               --config-steps steps (300, independent of --steps, so the driver's short command and a long run agree),
               each with its own cpu_baseline;
  closed_loop  one host iteration per env-step: the step's actions are computed ON THE DEVICE by torch from the previous
               observation, three env slices double-buffered over ce_step_range — what an RL sampler gets;
  boundary     the RLlib vector hook (contracts_amd.vector_env.BatchedBaseEnv) at E = 16384: the tensor path
               (send_actions_array / poll_tensors) and the dict protocol (send_actions / poll), PCIe- and Python-inclusive.
Prints ONE JSON line on rank 0 (DESIGN.md §5 explains every field).
"""
import argparse
import json
import os
import statistics
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from contracts_amd import parallel  # noqa: E402  (no GPU / torch import at module level)

T_START = time.monotonic()  # --time-budget counts from here (interpreter start, before torch is imported)

SEED0 = 73907  # the reference's seed multiplier (runner.py:130)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PREROLL = 300
# algorithmic bytes per env-step (state read + written once, uint8 obs); feature kinds: list stamps +
# agents + accumulators once, int16 feature rows, rewards, infos (same accounting)
WORKLOADS = {
    "C4": dict(kind="cleanup", n=8, E=16384, contract="cleanup", algo=7235,
               name="cleanup_new 8 agents + CleanupContract, 16384 envs/GPU, horizon 1000, auto-reset"),
    "C2": dict(kind="cleanup", n=4, E=4096, contract="cleanup", algo=4211,
               name="cleanup_new 4 agents + CleanupContract, 4096 envs"),
    "C3": dict(kind="harvest", n=8, E=16384, contract="harvest_local", algo=7313,
               name="harvest_new 8 agents + HarvestFeaturemodLocalContract, 16384 envs"),
    # selfdrive episodes end (and reset) at their own pace and every reset draws from both MT19937 streams: the steady state
    # — resets spread over the steps, generations running out at their natural rate — is reached after a few thousand steps
    "C5": dict(kind="selfdrive", n=4, E=32768, contract="selfdrive_distprop", algo=863, preroll=12000,
               name="selfdrive 4 agents + SelfdriveContractDistprop, 32768 envs (float64 path)"),
    "C1": dict(kind="harvest_features", n=2, E=16384, contract="harvest_local", algo=887,
               name="harvest (HarvestFeatures) 2 agents + HarvestFeaturemodLocalContract, batched to 16384 envs"),
}
# BASELINE.md "CPU reference measured by the survey": the Python reference, agent-steps/s of one process, per (kind, agents)
PYTHON_REFERENCE_PER_CORE = {("cleanup", 8): 4259, ("cleanup", 4): 4203, ("harvest", 8): 1567, ("selfdrive", 4): 29578,
                             ("harvest_features", 2): 1846}
DTYPE = {"cleanup": "u8", "harvest": "u8", "harvest_features": "u8", "cleanup_features": "u8", "selfdrive": "f64"}
FUSED_KINDS = ("cleanup", "harvest", "selfdrive", "harvest_features", "cleanup_features")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--kind", default=None, choices=["cleanup", "harvest", "selfdrive", "harvest_features", "cleanup_features"])
    ap.add_argument("--streams", type=int, default=3,
                    help="env slices per rank, each stepped on its own HIP stream (slices are independent; one slice's "
                         "kernel tail overlaps the next slice's head)")
    ap.add_argument("--fused-steps", type=int, default=16, help="steps per launch of the fused mode (0 = skip it)")
    ap.add_argument("--repeats", type=int, default=3, help="minimum number of timed K-step repeats (median reported)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="keep repeating until this much timed wall time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (C1, C2, C3, C5)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--config-steps", type=int, default=300, help="steps per timed repeat of the C1/C2/C3/C5 rows")
    ap.add_argument("--config-seconds", type=float, default=0.3, help="minimum timed wall per mode of a config row")
    ap.add_argument("--config-cpu-seconds", type=float, default=2.5, help="CPU baseline sample per config row")
    ap.add_argument("--no-closed-loop", action="store_true")
    ap.add_argument("--no-boundary", action="store_true")
    ap.add_argument("--no-counter-rng", action="store_true", help="skip the counter-RNG rows beside the MT19937 ones")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/traffic.json instead of two "
                    "rocprofv3 --pmc child passes in the run")
    ap.add_argument("--rng", default="mt19937", choices=["mt19937", "counter"],
                    help="stream of the HEADLINE rows: mt19937 = the reference's (default: the only mode BASELINE's metric is about); "
                         "counter = the engine's own Philox stream, grid kinds only — the line then says so in config.rng")
    ap.add_argument("--full", action="store_true", help="also print the full record (every row, note and per-mode table; ~25 KB) as "
                    "a stdout line BEFORE the compact one")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="file the full record goes to ('' = nowhere); the stdout line is the compact record (< 3 KB)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="--gpus N self-launch: deadline of the whole job, s")
    ap.add_argument("--time-budget", type=float, default=420.0,
                    help="seconds from process start after which the OPTIONAL sections still to come (config rows, closed loop, "
                         "boundary, counter stream, the configs' CPU baselines) are skipped and say so, so that the line — headline, "
                         "roofline, cpu_baseline, parity_in_run — still prints on a slow box; a normal run takes ~70 s; 0 = no limit")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle on the host cores, same envs / seeds / actions as the GPU run's first envs
# ------------------------------------------------------------------------------------------------------------------
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


def cpu_baseline(wl, target_s, single_thread_s=4.0):
    """The CPU oracle (oracle/oracle.c, pinned bit-exact to the reference's golden traces) timed on this box's host cores
    with OpenMP over envs.  Like for like with the GPU run: the first `256 x threads` envs of the same batch (global
    indices 0.., seeds SEED0 + b), the same counter-hash actions (contracts_amd.synth mirrors the device generator), the
    same horizon / auto-reset, from t = 0; all threads (the reported value) and, when single_thread_s > 0, one thread."""
    import ctypes
    import numpy as np
    from contracts_amd import synth
    from oracle.pyoracle import Oracle
    kind, n, contract = wl["kind"], wl["n"], wl["contract"]
    threads = _usable_cores()
    gomp = ctypes.CDLL("libgomp.so.1")
    na = {"cleanup": 8, "harvest": 7, "harvest_features": 7, "cleanup_features": 8}.get(kind)
    CH = 32

    def run(nthreads, E, seconds, keep_state=False):
        gomp.omp_set_num_threads(nthreads)
        orc = Oracle(kind, E, n, contract=contract, horizon=1000, auto_reset=True)
        orc.seed(seed0=SEED0)
        orc.reset()
        steps, spent = 0, 0.0
        while spent < seconds:
            if kind == "selfdrive":  # generated outside the timed span
                acts = synth.synth_actions_f32(SEED0 + 1, 0, E, n, steps, CH)
            else:
                acts = synth.synth_actions_u8(SEED0 + 1, 0, E, n, steps, CH, na)
            t0 = time.perf_counter()
            for t in range(CH):
                orc.step(acts[t])
            spent += time.perf_counter() - t0
            steps += CH
        state = None
        if keep_state:  # what parity_in_run() holds the engine to: every persistent field and the last step's outputs
            state = {f: np.array(getattr(orc, f)) for f in PARITY_FIELDS[kind]}
        orc.close()
        return E * n * steps / spent, steps, spent, state

    E_all = min(256 * threads, wl["E"])
    v_all, steps, dt, state = run(threads, E_all, target_s, keep_state=True)
    out = {"value": v_all, "unit": "agent-steps/s", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
           "_oracle_state": {"envs": E_all, "steps": steps, "fields": state},  # consumed (and removed) by parity_in_run()
           "sample": "envs 0..%d of the same batch (seeds %d + b, the same counter-hash actions, horizon 1000, auto-reset), "
                     "steps 0..%d, %s n=%d + contract, OpenMP over envs on %d threads, %.1f s of stepping"
                     % (E_all - 1, SEED0, steps - 1, kind, n, threads, dt)}
    if single_thread_s > 0:
        v_one, steps1, dt1, _ = run(1, 256, min(single_thread_s, target_s))
        out["single_thread_value"] = v_one
        out["sample"] += " (+ envs 0..255 x %d steps on 1 thread, %.1f s)" % (steps1, dt1)
    ref = PYTHON_REFERENCE_PER_CORE.get((kind, n))
    if ref is not None:
        out["python_reference_per_core"] = ref  # the reference itself (1 process), measured by the survey
    return out


_GRID_FIELDS = ("grid", "agents", "spawn_perm", "rng", "timestep", "theta", "obs", "base_reward", "reward", "done", "info", "features",
                "int_metrics", "f64_metrics", "final_int_metrics", "final_f64_metrics")
_FEAT_FIELDS = tuple(f for f in _GRID_FIELDS if f not in ("spawn_perm", "obs"))
PARITY_FIELDS = {"cleanup": _GRID_FIELDS + ("waste_perm",), "harvest": _GRID_FIELDS, "harvest_features": _FEAT_FIELDS,
                 "cleanup_features": _FEAT_FIELDS,
                 "selfdrive": ("sd_state", "rng", "theta", "obs_f64", "base_reward", "reward", "done", "done_agents", "info", "sd_info",
                               "f64_metrics")}


def parity_in_run(wl, oracle_state, device_index, streams, fused_T):
    """The bench line checks its own kernels: a second engine handle steps the SAME envs the cpu_baseline
    leg just stepped on the oracle (global indices 0.., seeds SEED0 + b, the same counter-hash actions, horizon 1000 with
    in-launch auto-reset, from t = 0) to the same step count — the first part as per-step launches, the rest as fused rollouts,
    i.e. through both kernels the row times — and every persistent field plus the last step's outputs is compared with the
    oracle's: bit-exact for integers / bytes / stream state, <= 1e-9 for the float64 rewards and metrics.  Outside every timed
    span; the oracle is the checker here, never the thing measured."""
    import hashlib
    import numpy as np
    import torch
    from contracts_amd.engine import BatchedEnv
    kind, n = wl["kind"], wl["n"]
    E, K, ref = oracle_state["envs"], oracle_state["steps"], oracle_state["fields"]
    # the handle is the timed row's: the row's WHOLE batch, cut into the same slices on as many streams — so the launches this
    # check makes are the exact kernel instances and launch shapes the row times; the oracle's sample is
    # envs 0..E-1 of it (global indices, seeds and counter-hash actions are keyed by the env index, not by the batch size)
    EH = max(E, wl["E"])
    S = max(1, min(streams, EH))
    env = BatchedEnv(kind, EH, n, contract=wl["contract"], horizon=1000, auto_reset=True, device=device_index,
                     env_index_base=0, rng=wl.get("rng", "mt19937"))
    try:
        env.seed(seed0=SEED0)
        env.reset()
        sts = [torch.cuda.Stream() for _ in range(S)] if S > 1 else None
        handles = [st.cuda_stream for st in sts] if sts else None
        do_fused = bool(fused_T) and kind in FUSED_KINDS
        k_fused = (K // 2) // fused_T * fused_T if do_fused else 0
        k_step = K - k_fused
        dt = torch.float32 if kind == "selfdrive" else torch.uint8
        CH = 512
        buf = torch.empty((CH, EH, n), dtype=dt, device="cuda")
        t = 0
        while t < K:
            fused = t >= k_step
            c = min(CH, (K if fused else k_step) - t)
            if fused:
                c = c // fused_T * fused_T or c
            env.synth_actions(SEED0 + 1, t, c, buf.data_ptr())
            env.synchronize()
            if fused:
                env.rollout_fused(buf.data_ptr(), c, fused_T, None, handles)
            else:
                env.rollout_device(buf.data_ptr(), c, handles)
            torch.cuda.synchronize()
            t += c
        env.check_faults()
        bad, digest = {}, hashlib.sha256()
        for f, y in ref.items():
            x = env.download(f, raw=True) if kind in ("harvest_features", "cleanup_features") and f == "grid" else env.download(f)
            x = x[:E]  # the oracle's sample: the first E envs of the batch
            if f == "rng" and wl.get("rng", "mt19937") == "mt19937":  # key[624] + position of each MT19937 block (pad words are free)
                x, y = x.reshape(E, -1, 628)[:, :, :625], y.reshape(E, -1, 628)[:, :, :625]
            digest.update(np.ascontiguousarray(x).tobytes())
            same = (np.allclose(x, y, rtol=0, atol=1e-9, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y))
            if not same:
                rows = np.nonzero((np.asarray(x) != np.asarray(y)).reshape(E, -1).any(axis=1))[0]
                bad[f] = {"envs_differing": int(rows.size), "first": [int(r) for r in rows[:4]]}
        out = {"ok": not bad, "envs": E, "steps": K, "per_step_steps": k_step, "fused_steps": k_fused, "slices": S,
               "handle_envs": EH, "envs_per_launch": EH // S,
               "episode_ends_crossed": K // 1000 if kind != "selfdrive" else "at each env's own pace",
               "fields": list(ref), "engine_sha256_16": digest.hexdigest()[:16],
               "checker": "oracle/oracle.c (the cpu_baseline leg's final state: same envs, seeds, actions, step count)",
               "tolerance": "bit-exact (integers, bytes, generator state); float64 rewards / metrics <= 1e-9"}
        if bad:
            out["mismatches"] = bad
        return out
    finally:
        env.close()


def oracle_sample(wl, E, K):
    """the oracle stepped K steps on envs 0..E-1 of a workload (seeds SEED0 + b, the counter-hash actions, horizon 1000 with
    auto-reset) — untimed; returns what parity_in_run() compares an engine handle with"""
    import numpy as np
    from contracts_amd import synth
    from oracle.pyoracle import Oracle
    kind, n = wl["kind"], wl["n"]
    na = {"cleanup": 8, "harvest": 7, "harvest_features": 7, "cleanup_features": 8}.get(kind)
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(_usable_cores())  # (the cpu_baseline leg may have left it at one)
    orc = Oracle(kind, E, n, contract=wl["contract"], horizon=1000, auto_reset=True, rng=wl.get("rng", "mt19937"))
    orc.seed(seed0=SEED0)
    orc.reset()
    for t0 in range(0, K, 32):
        c = min(32, K - t0)
        acts = synth.synth_actions_f32(SEED0 + 1, 0, E, n, t0, c) if kind == "selfdrive" else synth.synth_actions_u8(SEED0 + 1, 0, E, n, t0, c, na)
        for t in range(c):
            orc.step(acts[t])
    state = {f: np.array(getattr(orc, f)) for f in PARITY_FIELDS[kind]}
    orc.close()
    return {"envs": E, "steps": K, "fields": state}


def live_traffic(wl, E, streams, seconds=120.0):
    """HBM-side bytes per env-step of the headline's per-step kernel measured IN THIS RUN (the line's
    `roofline.traffic` was a committed constant): two child processes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
    passes, counters only — no tracing) over tools/pmc_driver.py, which steps the same workload (300-step pre-roll with the fused
    kernel, then 64 measured single-step launches per slice); bytes = FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide
    coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both reported in KB, summed over the step kernel's
    dispatches and divided by envs x steps.  Returns (bytes per env-step, note) or (None, why) — the caller then falls back to
    profiles/traffic.json.  Children are started with a deadline and never replace this process."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    kern = {"cleanup": "k_grid_step", "harvest": "k_grid_step", "selfdrive": "k_sd_step", "harvest_features": "k_feat_step",
            "cleanup_features": "k_feat_step"}[wl["kind"]]
    steps, out, t_end = 64, {}, time.time() + seconds
    root = tempfile.mkdtemp(prefix="ce_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(root, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_driver.py"),
                   "--mode", "step", "--kind", wl["kind"], "--agents", str(wl["n"]), "--envs", str(E), "--steps", str(steps),
                   "--streams", str(streams)]
            left = t_end - time.time()
            if left < 10:
                return None, "out of time before the %s pass" % ctr
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=left)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (ctr, p.returncode)
            total = 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kern in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        total += float(row["Counter_Value"])
            if total <= 0:
                return None, "no %s samples of %s" % (ctr, kern)
            out[ctr] = total * 1024.0 / (E * steps)
        return out["FETCH_SIZE"] * 2.0 + out["WRITE_SIZE"], \
            "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/pmc_driver.py, %d single-step " \
            "launches per slice of the same workload; FETCH x 2 + WRITE = %.0f + %.0f B per env-step" % (steps, out["FETCH_SIZE"] * 2.0, out["WRITE_SIZE"])
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out"
    except Exception as exc:  # noqa: BLE001 (reported in traffic_source, the committed constant stands in)
        return None, "%s: %s" % (type(exc).__name__, str(exc)[:120])
    finally:
        shutil.rmtree(root, ignore_errors=True)


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


# ------------------------------------------------------------------------------------------------------------------
# one workload, one mode
# ------------------------------------------------------------------------------------------------------------------
class Runner:
    """one engine handle of this rank + its resident action planes; measures one mode at a time"""

    def __init__(self, group, wl, E, K, W, streams, device_index, fused_T=0):
        import torch
        from contracts_amd.engine import BatchedEnv
        self.torch, self.group, self.wl, self.E, self.K, self.W = torch, group, wl, E, K, W
        kind, n = wl["kind"], wl["n"]
        base, _ = group.shard(E)  # rank g owns global envs [g*E, (g+1)*E); seeds / actions are keyed by the global index
        self.env = BatchedEnv(kind, E, n, contract=wl["contract"], horizon=1000, auto_reset=True, device=device_index,
                              env_index_base=base, rng=wl.get("rng", "mt19937"))
        self.env.seed(seed0=SEED0)
        self.env.reset()
        dt = torch.float32 if kind == "selfdrive" else torch.uint8
        self.esz = 4 if kind == "selfdrive" else 1
        self.plane = E * n * self.esz
        # all action planes resident in HBM before any timing: pre-roll, warm-up, then K planes reused by every repeat
        # (each repeat continues from the state the previous one left, so no two repeats replay the same trajectory)
        # fused mode times whole launches: ceil(K / T) * T steps (reported as its own `steps`)
        self.Kf = -(-K // fused_T) * fused_T if fused_T else K
        planes = PREROLL + W + max(K, self.Kf)
        self.acts = torch.empty((planes, E, n), dtype=dt, device="cuda")
        self.env.synth_actions(SEED0 + 1, 0, planes, self.acts.data_ptr())
        self.env.synchronize()
        S = max(1, min(streams, E))
        self.S = S
        self.streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
        self.handles = [st.cuda_stream for st in self.streams] if S > 1 else None
        self.traj = None
        self.preroll = max(PREROLL, int(wl.get("preroll", PREROLL)))
        for _ in range(self.preroll // PREROLL):  # long pre-rolls replay the pre-roll planes (always per-step launches)
            self.env.rollout_device(self.acts.data_ptr(), PREROLL, self.handles)
        torch.cuda.synchronize()

    def _issue(self, mode, first_plane, count, T):
        ptr = self.acts.data_ptr() + first_plane * self.plane
        if mode == "fused":
            self.env.rollout_fused(ptr, count, T, self.traj, self.handles)
        else:
            self.env.rollout_device(ptr, count, self.handles)  # the launch loop itself runs in C (ce_rollout)

    def _fence(self):
        self.torch.cuda.synchronize()
        self.group.barrier()
        self.torch.cuda.synchronize()

    def measure(self, mode, T=0, min_repeats=3, min_seconds=1.0, max_repeats=25):
        torch, K, W, E, n = self.torch, self.K, self.W, self.E, self.wl["n"]
        if mode == "fused":
            K = self.Kf  # whole launches only: every timed launch runs T steps
            assert K % T == 0
            if self.traj is None:
                self.traj = self.env.alloc_trajectory(T)  # every per-step output kept, ring of T planes
        if W:
            self._issue(mode, PREROLL, W, T)
        elapsed, ev_ms, self.rank_spread = [], [], []
        while len(elapsed) < min_repeats or (sum(elapsed) < min_seconds and len(elapsed) < max_repeats):
            self._fence()
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in self.streams]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in self.streams]
            for e0, st in zip(ev0, self.streams):
                e0.record(st)
            t0 = time.perf_counter()
            self._issue(mode, PREROLL + W, K, T)
            for e1, st in zip(ev1, self.streams):
                e1.record(st)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            self._fence()
            elapsed.append(self.group.max(t1 - t0))  # MAX over ranks
            if self.group.world > 1:  # the fastest rank's clock beside the slowest's: what the MAX hides
                self.rank_spread.append((self.group.min(t1 - t0), elapsed[-1]))
            ev_ms.append(max(a.elapsed_time(b) for a, b in zip(ev0, ev1)))
        self.env.check_faults()
        med = statistics.median(elapsed)
        units = self.group.world * E * n * K
        launches = self.S * K if mode == "per_step" else self.S * (K // T)  # launches actually issued per repeat
        launches_per_step = launches / float(K)
        res = {"mode": mode, "steps": K, "value": units / med, "value_min": units / max(elapsed), "value_max": units / min(elapsed),
               "repeats": len(elapsed), "ms_per_step": med / K * 1e3, "timed_seconds": sum(elapsed),
               # HIP events on the launch streams (slowest stream per repeat, median repeat): the device-side time of
               # the same K steps, without the host's launch / synchronize overhead
               "event_ms_per_step": statistics.median(ev_ms) / K, "streams": self.S, "launches_per_step": launches_per_step}
        if self.rank_spread:  # per-rank clocks of the same repeats: fastest and slowest rank (median repeat each)
            res["ms_per_step_rank_min"] = statistics.median(lo for lo, _ in self.rank_spread) / K * 1e3
            res["ms_per_step_rank_max"] = statistics.median(hi for _, hi in self.rank_spread) / K * 1e3
        if mode == "fused":
            res["steps_per_launch"] = T
            res["trajectory_planes"] = self.traj.P
        return res

    def roofline(self, res, kernel, traffic_key):
        """algorithmic bytes of one step of this rank's batch over the wall time of a step"""
        E, algo = self.E, self.wl["algo"]
        ms = res["ms_per_step"]
        achieved = algo * E / (ms * 1e-3) / 1e9
        envs_per_launch = E / float(self.S)
        steps_per_launch = res.get("steps_per_launch", 1)
        traffic = ratio = source = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            row = tj.get(traffic_key)
            if row and row.get("kind") == self.wl["kind"] and row.get("agents") == self.wl["n"]:
                traffic = int(round(row["hbm_bytes_per_env_step"] * envs_per_launch * steps_per_launch))
                ratio = round(row["hbm_bytes_per_env_step"] / float(algo), 3)  # PMC bytes over algorithmic bytes
                prov = tj.get("_provenance", {})
                # NOT measured in this run: a constant from the committed PMC passes, with where it came from
                source = "profiles/traffic.json @ %s (kernel sources %s; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % (
                    prov.get("git_head", "unknown commit"), prov.get("kernels_sha16", "unrecorded"))
        except (OSError, ValueError):
            traffic = None
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_ratio": ratio, "traffic_source": source, "kernel": kernel,
                "note": "traffic = FETCH_SIZE x 2 + WRITE_SIZE (PMC): requests served by the 256 MB Infinity Cache count like HBM "
                        "accesses, and this batch largely lives there — an upper bound of the HBM bytes; counter_rng.large_batch "
                        "(262144 envs) is the beyond-cache figure",
                "algorithmic_bytes_per_env_step": algo,
                "algorithmic_bytes_per_launch": int(round(algo * envs_per_launch * steps_per_launch)),
                "envs_per_launch": envs_per_launch, "steps_per_launch": steps_per_launch,
                "launch_ms": ms * steps_per_launch,  # S launches run concurrently and finish one step of the batch per ms_per_step
                "event_ms_per_step": res["event_ms_per_step"]}

    def sanity(self):
        if self.wl["kind"] in ("cleanup", "harvest"):
            mi = self.env.download("int_metrics")
            return {"apples_eaten_running_episode_mean": float(mi[:, 0].mean()), "dirt_cleaned_mean": float(mi[:, 2].mean()),
                    "timestep_mean": float(self.env.download("timestep").mean())}
        return {}

    def close(self):
        self.traj = None
        self.env.close()
        del self.acts


def closed_loop(wl, E, device_index, num_slices, min_seconds=0.5, only=None):
    """The RL-usable rate: one host iteration per env-step.  The batch is `num_slices` independent env slices, each with
    its own HIP stream; per slice and step, torch computes the actions ON THE DEVICE from the observation the previous
    step of that slice wrote (the pixel in front of every agent, accumulated into a resident noise plane so the action
    distribution stays the benchmark's uniform one: two elementwise torch kernels), then ce_step_range steps the slice
    from that device pointer.  No joins between slices: one slice's policy and kernel tail overlap another's step.
    Ways to issue a slice's tick: `eager` (three launches from Python), `graph` (the same three launches captured once per
    slice into a hipGraph and replayed: one host call per slice and step) — `value` is the faster of these two, both
    with one host iteration per env-step — and `graph16` (16 consecutive ticks per replay: the loop observation -> policy
    -> step still closes on the device every step, the host only comes by every 16 steps; reported beside, not as
    `value`)."""
    import torch
    from contracts_amd.engine import BatchedEnv
    kind, n = wl["kind"], wl["n"]
    A = 8 if kind == "cleanup" else 7
    env = BatchedEnv(kind, E, n, contract=wl["contract"], horizon=1000, auto_reset=True, device=device_index,
                     rng=wl.get("rng", "mt19937"))
    env.seed(seed0=SEED0)
    env.reset()
    obs = env.torch_tensors()["obs"]  # uint8 [E, n, 15, 15, 3], strided view of the engine's pitched buffer
    ahead = obs[:, :, 6, 7, 1]        # green channel of the cell in front of the agent: [E, n], stride-only view
    g = torch.Generator(device="cuda").manual_seed(1)
    noise = torch.randint(0, 256, (E, n), dtype=torch.uint8, device="cuda", generator=g)
    acts = torch.zeros((E, n), dtype=torch.uint8, device="cuda")
    S = max(1, num_slices)
    streams = [torch.cuda.Stream() for _ in range(S)]
    bounds = [(E * s // S, E * (s + 1) // S) for s in range(S)]
    pre = torch.empty((PREROLL, E, n), dtype=torch.uint8, device="cuda")
    env.synth_actions(SEED0 + 1, 0, PREROLL, pre.data_ptr())
    env.synchronize()  # the generator runs on the null stream, the slices' streams are non-blocking: without this the first
    # pre-roll launches can read planes that are not written yet (zeros in fresh memory, anything in a recycled torch block)
    env.rollout_device(pre.data_ptr(), PREROLL, [st.cuda_stream for st in streams])
    torch.cuda.synchronize()

    def slice_tick(st, b0, b1):  # every pointer is fixed, so the same calls can be captured into a graph
        # noise <- noise + pixel (uint8 wrap-around): stays uniform whatever the pixels were, and moves on every step
        torch.add(ahead[b0:b1], noise[b0:b1], out=noise[b0:b1])
        if A == 8:
            torch.bitwise_and(noise[b0:b1], 7, out=acts[b0:b1])
        else:
            torch.remainder(noise[b0:b1], A, out=acts[b0:b1])
        env.step_range_device(acts.data_ptr(), b0, b1 - b0, stream=st.cuda_stream)

    def slice_tick_policy(st, b0, b1):
        # the same policy with the action selection fused into the step kernel's action load (ce_step_policy,
        # CE_POLICY_BYTES_MOD: action = policy byte mod |A|): ONE policy kernel + one step launch per slice and tick
        torch.add(ahead[b0:b1], noise[b0:b1], out=noise[b0:b1])
        env.step_policy_device(noise.data_ptr(), "bytes", b0, b1 - b0, stream=st.cuda_stream)

    def slice_tick_inkernel(st, b0, b1):
        # CE_POLICY_AHEAD_NOISE: the same policy evaluated INSIDE the step kernel's action load (the noise byte moves on by the
        # green channel of the pixel ahead in the previous view, action = byte mod |A|): ONE launch per slice and tick
        env.step_policy_device(noise.data_ptr(), "ahead_noise", b0, b1 - b0, stream=st.cuda_stream)

    handles = [st.cuda_stream for st in streams]

    def sliced_tick_inkernel():  # all slices of a tick in ONE host call (the launch loop in C: ce_step_policy_sliced)
        env.step_policy_sliced(noise.data_ptr(), "ahead_noise", handles)

    def eager_tick(tick_fn=slice_tick):
        for st, (b0, b1) in zip(streams, bounds):
            with torch.cuda.stream(st):
                tick_fn(st, b0, b1)

    def timed(tick, K=480, per_tick=1):
        for _ in range(48 // per_tick):
            tick()
        torch.cuda.synchronize()
        elapsed = []
        while len(elapsed) < 3 or sum(elapsed) < min_seconds:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K // per_tick):
                tick()
            torch.cuda.synchronize()
            elapsed.append(time.perf_counter() - t0)
        env.check_faults()
        med = statistics.median(elapsed)
        return {"value": E * n * K / med, "ms_per_step": med / K * 1e3, "steps": K, "repeats": len(elapsed),
                "timed_seconds": sum(elapsed)}

    modes = {}
    if only is None or "eager" in only:
        modes["eager"] = dict(timed(eager_tick), host_calls_per_step=3 * S, host_iterations_per_step=1, launches_per_slice_and_step=3)
    if only is None or "policy_eager" in only:
        modes["policy_eager"] = dict(timed(lambda: eager_tick(slice_tick_policy)), host_calls_per_step=2 * S,
                                     host_iterations_per_step=1, launches_per_slice_and_step=2)
    if only is None or "inkernel_eager" in only:
        modes["inkernel_eager"] = dict(timed(lambda: eager_tick(slice_tick_inkernel)), host_calls_per_step=S,
                                       host_iterations_per_step=1, launches_per_slice_and_step=1)
    if only is None or "inkernel_sliced" in only:
        modes["inkernel_sliced"] = dict(timed(sliced_tick_inkernel), host_calls_per_step=1, host_iterations_per_step=1,
                                        launches_per_slice_and_step=1)
    # ONE graph over all slices (cross-stream fork / join inside the capture): one replay per tick whatever the slice count
    for name, tick_fn, per_slice in (("policy_graph_all", slice_tick_policy, 2), ("inkernel_graph_all", slice_tick_inkernel, 1)):
        if only is not None and name not in only:
            continue
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[0]):
                for st in streams[1:]:
                    st.wait_stream(streams[0])  # fork
                for st, (b0, b1) in zip(streams, bounds):
                    with torch.cuda.stream(st):
                        tick_fn(st, b0, b1)
                for st in streams[1:]:
                    streams[0].wait_stream(st)  # join
            torch.cuda.synchronize()

            def all_tick(gr=gr):
                with torch.cuda.stream(streams[0]):
                    gr.replay()

            modes[name] = dict(timed(all_tick), host_calls_per_step=1, host_iterations_per_step=1, launches_per_slice_and_step=per_slice)
        except Exception as exc:
            modes[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    for name, ticks, tick_fn in (("graph", 1, slice_tick), ("graph16", 16, slice_tick), ("policy_graph", 1, slice_tick_policy),
                                 ("policy_graph16", 16, slice_tick_policy)):
        if only is not None and name not in only:
            continue
        try:  # one hipGraph per slice, replayed on the slice's stream
            graphs = []
            for st, (b0, b1) in zip(streams, bounds):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(ticks):
                        tick_fn(st, b0, b1)
                graphs.append(gr)
            torch.cuda.synchronize()

            def graph_tick(graphs=graphs):
                for st, gr in zip(streams, graphs):
                    with torch.cuda.stream(st):
                        gr.replay()

            modes[name] = dict(timed(graph_tick, per_tick=ticks), host_calls_per_step=S / float(ticks),
                               host_iterations_per_step=1.0 / ticks, device_policy_evaluations_per_step=1,
                               launches_per_slice_and_step=2 if tick_fn is slice_tick_policy else 3)
        except Exception as exc:  # capture support differs between ROCm builds: the eager figure stands
            modes[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    one_iter = ("eager", "graph", "policy_eager", "policy_graph", "policy_graph_all", "inkernel_eager", "inkernel_sliced", "inkernel_graph_all")
    best_any = max((m for m in one_iter if "value" in modes.get(m, {})), key=lambda m: modes[m]["value"])
    separate = [m for m in one_iter if not m.startswith("inkernel") and "value" in modes.get(m, {})]
    inker = [m for m in one_iter if m.startswith("inkernel") and "value" in modes.get(m, {})]
    best_sep = max(separate, key=lambda m: modes[m]["value"]) if separate else None
    best_ink = max(inker, key=lambda m: modes[m]["value"]) if inker else None
    # `value` = the best row whose policy is a kernel of its OWN (what a policy network needs; comparable across rounds —
    # ADVICE r05).  The inkernel_* rows evaluate the benchmark's toy policy inside the step kernel's action load
    # (ce_step_policy CE_POLICY_AHEAD_NOISE): a separately named figure, never `value`.
    best = best_sep if best_sep is not None else best_any
    out = dict(modes[best], unit="agent-steps/s", issue=best, slices=S, modes=modes,
               best_with_separate_policy_kernel=None if best_sep is None else dict(
                   {k: modes[best_sep][k] for k in ("value", "ms_per_step", "host_calls_per_step")}, issue=best_sep),
               inkernel_policy=None if best_ink is None else dict(
                   {k: modes[best_ink][k] for k in ("value", "ms_per_step", "host_calls_per_step")}, issue=best_ink),
               policy="torch on device: action = (green(pixel ahead of the agent in the previous observation) + resident noise "
                      "byte, accumulated) mod %d — eager / graph: two elementwise kernels + the step per slice and tick; policy_*: "
                      "one elementwise kernel + ce_step_policy (the modulo happens in the step kernel's action load); inkernel_*: the whole "
                      "policy inside the step kernel (CE_POLICY_AHEAD_NOISE: it reads the pixel ahead from the previous view itself), one "
                      "launch per slice and tick; *_sliced = the launch loop in C (one host call per tick), *_graph_all = ONE hipGraph over "
                      "all slices (fork / join inside the capture)" % A,
               workload=wl["name"])
    env.close()
    return out


def counter_rng(group, wl, a, device_index, big_E=262144):
    """The engine's counter-RNG mode (CE_FLAG_RNG_COUNTER: Philox4x32-10 blocks, 16 bytes of generator state per env — NOT
    the reference's stream, so never the headline) beside the MT19937 mode, same protocol as the config rows: at the
    headline's batch, and at a batch far beyond the Infinity Cache where the MT19937 state round trip is real HBM traffic."""
    out = {"rng": "philox4x32-10 counter stream, 512-word generations, a fresh generation per env-step",
           "note": "same step logic and draw order; differs from the reference's np.random stream by construction "
                   "(parity: engine == oracle restatement of this stream, tests/test_counter_rng_gpu.py)"}
    for label, E, modes in (("headline_batch", wl["E"], ("counter",)), ("large_batch", big_E, ("mt19937", "counter"))):
        row = {"envs_per_gpu": E, "agents": wl["n"]}
        for mode in modes:
            fused_T = a.fused_steps if label == "headline_batch" else 0
            r = Runner(group, dict(wl, rng=mode), E, a.config_steps, min(a.warmup, 20), a.streams, device_index, fused_T)
            m = r.measure("per_step", min_repeats=3, min_seconds=a.config_seconds, max_repeats=100000)
            cell = {"value": m["value"], "ms_per_step": m["ms_per_step"], "steps": m["steps"], "repeats": m["repeats"],
                    "roofline_frac": r.roofline(m, kernel_names(wl["kind"], wl["n"])[0], "-")["frac"]}
            if fused_T:
                f = r.measure("fused", T=fused_T, min_repeats=3, min_seconds=a.config_seconds, max_repeats=100000)
                cell["fused"] = {"value": f["value"], "ms_per_step": f["ms_per_step"], "steps": f["steps"],
                                 "steps_per_launch": f["steps_per_launch"],
                                 "roofline_frac": r.roofline(f, kernel_names(wl["kind"], wl["n"])[1], "-")["frac"]}
            row[mode] = cell
            r.close()
        out[label] = row
    if not a.no_cpu_baseline:  # the counter-mode kernels checked in the run too: a short oracle sample on the same stream
        wc = dict(wl, rng="counter")
        out["parity_in_run"] = parity_in_run(wc, oracle_sample(wc, 2048, 1152), device_index, a.streams, a.fused_steps)
    cl = closed_loop(dict(wl, rng="counter"), wl["E"], device_index, a.streams)  # the per-step path is the closed loop's
    out["closed_loop"] = {"value": cl["value"], "ms_per_step": cl["ms_per_step"], "issue": cl["issue"],
                          "by_issue": {m: v.get("value", v.get("error")) for m, v in cl["modes"].items()}}
    return out


def boundary(wl, E, device_index):
    """The RLlib vector hook at the headline batch (contracts_amd.vector_env.BatchedBaseEnv), as a sampler on
    the host would drive it — PCIe- and Python-inclusive, never `value`:
      tensor path   send_actions_array(host uint8 [E, n]) + poll_tensors() (observations stay in HBM; rewards / dones read back)
      dict protocol send_actions({env: {agent: a}}) + poll() with every env's obs / reward / done / info dictionaries built"""
    import numpy as np
    from contracts_amd.vector_env import BatchedBaseEnv
    kind, n = wl["kind"], wl["n"]
    A = 8 if kind == "cleanup" else 7
    venv = BatchedBaseEnv(kind, E, n, contract=wl["contract"], seed0=SEED0, horizon=1000, device=device_index)
    venv.poll()
    rs = np.random.RandomState(0)
    planes = rs.randint(A, size=(8, E, n)).astype(np.uint8)
    out = {"workload": wl["name"], "envs": E}
    # tensor path
    for t in range(5):
        venv.send_actions_array(planes[t % 8])
        venv.poll_tensors()
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.5 or steps < 20:
        venv.send_actions_array(planes[steps % 8])
        tt = venv.poll_tensors()
        tt["reward"].sum().item()  # the sampler looks at the step's rewards (forces completion)
        steps += 1
    dt = time.perf_counter() - t0
    out["tensor_path"] = {"value": E * n * steps / dt, "unit": "agent-steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                          "what": "send_actions_array (host action plane over PCIe) + poll_tensors (zero-copy device views)"}
    # dict protocol: the policy side's per-env action dictionaries exist before the tick starts (RLlib's sampler builds them
    # from its policy's output; building them in this loop would time Python's dict constructor, not the hook)
    keys = ["a%d" % i for i in range(n)]
    t0 = time.perf_counter()
    action_dicts = [{e: dict(zip(keys, row)) for e, row in enumerate(planes[k].tolist())} for k in range(8)]
    ms_build = (time.perf_counter() - t0) / 8 * 1e3

    def dict_tick(t):
        venv.send_actions(action_dicts[t % 8])
        obs, rew, dones, infos, _ = venv.poll()
        k = 0
        # a sampler touches every env's dictionaries (RLlib walks `unfiltered_obs.items()` and looks the env's rewards / dones /
        # infos up beside it; the four mappings list the envs in the same order)
        for (e, o), r, d, i in zip(obs.items(), rew.values(), dones.values(), infos.values()):
            k += len(o)
        return k

    for t in range(3):  # the first two ticks build the two dictionary generations
        dict_tick(t)
    t0, steps, ticks = time.perf_counter(), 0, []
    while time.perf_counter() - t0 < 1.0 or steps < 4:
        ta = time.perf_counter()
        dict_tick(steps + 3)
        ticks.append(time.perf_counter() - ta)
        steps += 1
    dt = time.perf_counter() - t0
    # the same tick with the policy side's action dictionaries built INSIDE it (Python's dict constructor over E x n entries),
    # and with a consumer that copies every observation it is handed (what a collector must do with anything it keeps
    # beyond the next tick — vector_env.py's recycling contract; RLlib's Dict-space preprocessor does it by flattening)
    t1, k_incl = time.perf_counter(), 0
    while time.perf_counter() - t1 < 0.4 or k_incl < 3:
        pl = planes[k_incl % 8]
        venv.send_actions({e: dict(zip(keys, row)) for e, row in enumerate(pl.tolist())})
        obs, rew, dones, infos, _ = venv.poll()
        for (e, o), r, d, i in zip(obs.items(), rew.values(), dones.values(), infos.values()):
            pass
        k_incl += 1
    dt_incl = (time.perf_counter() - t1) / k_incl
    # ... and once more with everything alive so far moved out of the garbage collector's sight (BatchedBaseEnv.freeze_gc): the
    # tick above allocates 3 x 10^4 containers, i.e. a full collection every few ticks, each walking ~10^6 live objects
    import gc
    venv.freeze_gc()
    t1, k_frz = time.perf_counter(), 0
    while time.perf_counter() - t1 < 0.4 or k_frz < 3:
        pl = planes[k_frz % 8]
        venv.send_actions({e: dict(zip(keys, row)) for e, row in enumerate(pl.tolist())})
        obs, rew, dones, infos, _ = venv.poll()
        for (e, o), r, d, i in zip(obs.items(), rew.values(), dones.values(), infos.values()):
            pass
        k_frz += 1
    dt_frz = (time.perf_counter() - t1) / k_frz
    gc.unfreeze()
    t1 = time.perf_counter()
    for k in range(2):
        venv.send_actions(action_dicts[k % 8])
        obs, rew, dones, infos, _ = venv.poll()
        kept = [np.concatenate([v.ravel() for v in ao.values()]) for o in obs.values() for ao in o.values()]  # flatten = copy
    dt_copy = (time.perf_counter() - t1) / 2
    del kept
    ticks.sort()
    # `value` is the mean over the sample; this leg is host work (sixteen to thirty-two threads beside other tenants of the box), so
    # the spread of the single ticks is carried too
    out["dict_protocol"] = {"value": E * n * steps / dt, "unit": "agent-steps/s", "env_steps_per_s": E * steps / dt,
                            "ms_per_step": dt / steps * 1e3, "steps": steps, "host_threads": venv._host_threads(),
                            "ms_per_tick_min": ticks[0] * 1e3, "ms_per_tick_median": ticks[len(ticks) // 2] * 1e3,
                            "ms_per_tick_max": ticks[-1] * 1e3,
                            "ms_building_action_dicts_not_timed": ms_build, "last_tick_ms": dict(venv.tick_timing),
                            "value_incl_action_dicts": E * n / dt_incl, "ms_per_step_incl_action_dicts": dt_incl * 1e3,
                            "value_incl_action_dicts_gc_frozen": E * n / dt_frz, "ms_per_step_incl_action_dicts_gc_frozen": dt_frz * 1e3,
                            "value_with_consumer_copies": E * n / dt_copy, "ms_per_step_with_consumer_copies": dt_copy * 1e3,
                            "recycle_dicts": venv.recycle_dicts,
                            "contract": "what poll() returned at tick t is intact through t + 1 and rewritten in place by t + 2; "
                                        "`value` excludes building the action dictionaries (the policy side's product) and any copy the "
                                        "consumer makes; value_incl_action_dicts / value_with_consumer_copies (a Python-level flatten of "
                                        "every agent's observation, as RLlib's Dict preprocessor does) include them",
                            "what": "send_actions(pre-built {env: {agent: action}}) + poll with every env's obs (float64 image views) / "
                                    "reward / done / info dictionaries in hand and walked; recycled dictionary trees over page-locked "
                                    "snapshots (vector_env.py), observations converted to float64 on the host threads every tick"}
    # the same protocol with every dictionary rebuilt per tick (recycle_dicts=False: round 3's path), a few ticks
    venv.stop()
    venv = BatchedBaseEnv(kind, E, n, contract=wl["contract"], seed0=SEED0, horizon=1000, device=device_index, recycle_dicts=False)
    venv.poll()
    dict_tick(0)
    t0 = time.perf_counter()
    dict_tick(1)
    dict_tick(2)
    dt = (time.perf_counter() - t0) / 2
    out["dict_protocol_rebuilt"] = {"value": E * n / dt, "unit": "agent-steps/s", "ms_per_step": dt * 1e3,
                                    "what": "recycle_dicts=False: lazily built per-env dictionaries, every one looked up"}
    venv.stop()
    # the joint baseline (JointEnv semantics over ONE handle: vector_env.BatchedJointBaseEnv; reference two_stage_train.py:476-617,
    # experiment_configs/cleanup-joint-2agents.json): the centralised agent's [E, n] MultiDiscrete plane in, one launch per tick;
    # tensor path with the global colour map produced on the device (ce_global_view), and the dict protocol for a few ticks
    out["joint"] = joint_boundary(wl, E, device_index, planes)
    return out


def joint_boundary(wl, E, device_index, planes):
    import numpy as np
    from contracts_amd.vector_env import BatchedJointBaseEnv
    kind, n = wl["kind"], wl["n"]
    res = {"envs": E, "agents": n, "what": "JointEnv over one engine handle: summed rewards / infos, MultiDiscrete actions as one plane"}
    for mode in ("global", "concatenated"):
        venv = BatchedJointBaseEnv(kind, E, n, mode=mode, seed0=SEED0, horizon=1000, device=device_index)
        venv.poll()
        for t in range(5):
            venv.send_actions_array(planes[t % 8])
            venv.poll_tensors()
        t0, steps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.3 or steps < 20:
            venv.send_actions_array(planes[steps % 8])
            tt = venv.poll_tensors()
            tt["reward"].sum().item()  # the sampler looks at the step's rewards (forces completion, incl. the view launch)
            steps += 1
        dt = time.perf_counter() - t0
        row = {"tensor_value": E * n * steps / dt, "tensor_env_steps_per_s": E * steps / dt, "tensor_ms_per_step": dt / steps * 1e3}
        acts = [{e: {"a0": planes[k][e]} for e in range(E)} for k in range(2)]
        ticks = []
        for k in range(10):  # the first four are warm-up: both recycled image blocks get their pages there
            t0 = time.perf_counter()
            venv.send_actions(acts[k % 2])
            obs, rew, dones, infos, _ = venv.poll()
            for (e, o), r, d, i in zip(obs.items(), rew.values(), dones.values(), infos.values()):
                pass
            ticks.append(time.perf_counter() - t0)
        dt = sum(ticks[4:]) / len(ticks[4:])
        row.update({"dict_value": E * n / dt, "dict_env_steps_per_s": E / dt, "dict_ms_per_step": dt * 1e3, "dict_ticks": len(ticks) - 4,
                    "dict_ms_per_tick_max": max(ticks[4:]) * 1e3, "unit": "agent-steps/s"})
        del obs, rew, dones, infos
        res[mode] = row
        venv.stop()
    return res


KERNEL = {"cleanup": ("k_grid_step<cleanup>", "k_grid_rollout<cleanup>"), "harvest": ("k_grid_step<harvest>", "k_grid_rollout<harvest>"),
          "selfdrive": ("k_sd_step", "k_sd_rollout"), "harvest_features": ("k_feat_step<harvest>", "k_feat_rollout<harvest>"),
          "cleanup_features": ("k_feat_step<cleanup>", "k_feat_rollout<cleanup>")}


def kernel_names(kind, n):
    """(single-step, fused) kernel of a workload: HarvestFeatures with two agents runs the four-envs-per-wave pair"""
    if kind == "harvest_features" and n == 2 and os.environ.get("CE_FEAT_QUAD", "1")[:1] != "0":
        return ("k_feat_step_quad", "k_feat_rollout_quad")
    return KERNEL[kind]


def run_rank(a):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    rank, local_rank, world = parallel.rank_info()
    if world != max(a.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (a.gpus, world))
    # functional-test knobs (tests/test_bench_multirank_gpu.py): several ranks on ONE GPU with a gloo rendezvous —
    # exercises the launch / shard / reduction / report path where only a single-GPU box is available; never for numbers
    backend = os.environ.get("CONTRACTS_BENCH_BACKEND", "nccl")
    if os.environ.get("CONTRACTS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # CONTRACTS_BENCH_FORCE_DIST=1 (functional test, tests/test_bench_multirank_gpu.py): a one-rank run still builds its process
    # group, so the barrier and the reductions of this file go through RCCL on a single-GPU box
    group = parallel.Group(backend, "cuda:%d" % local_rank, force=os.environ.get("CONTRACTS_BENCH_FORCE_DIST") == "1")

    wl = dict(WORKLOADS["C4"])
    custom = bool(a.kind or a.agents)
    if custom:  # a hand-picked workload: its algorithmic bytes are only known when it coincides with a BASELINE config
        kind = a.kind or wl["kind"]
        n = a.agents or next((w["n"] for w in WORKLOADS.values() if w["kind"] == kind), wl["n"])
        match = next((w for w in WORKLOADS.values() if (w["kind"], w["n"]) == (kind, n)), None)
        contract = {"cleanup": "cleanup", "harvest": "harvest_local", "selfdrive": "selfdrive_distprop",
                    "harvest_features": "harvest_local", "cleanup_features": "cleanup"}[kind]
        wl = dict(match) if match else dict(kind=kind, n=n, E=wl["E"], contract=contract, algo=None,
                                            name="%s %d agents + contract (custom)" % (kind, n))
    E = a.envs_per_gpu or wl["E"]
    K, W = a.steps, a.warmup
    if a.rng == "counter":
        if wl["kind"] not in ("cleanup", "harvest"):
            raise SystemExit("bench.py: --rng counter belongs to the grid kinds")
        wl["rng"] = "counter"

    do_fused = bool(a.fused_steps) and wl["kind"] in FUSED_KINDS
    r = Runner(group, wl, E, K, W, a.streams, local_rank, a.fused_steps if do_fused else 0)
    head = r.measure("per_step", min_repeats=a.repeats, min_seconds=a.min_seconds, max_repeats=100000)
    fused = None
    if do_fused:
        fused = r.measure("fused", T=a.fused_steps, min_repeats=a.repeats, min_seconds=a.min_seconds, max_repeats=100000)
    stats = r.sanity()
    # proof of the job's width that does not rest on --gpus: every rank adds 1 over the process group (an all-reduce SUM
    # over RCCL under nccl); per-rank clocks of the timed repeats as min / max over ranks
    ranks = {"ranks_seen": int(round(group.sum(1.0))), "world_size": world,
             "backend": backend if group._dist is not None else "none (single process)",
             "device": torch.cuda.get_device_properties(local_rank).name, "envs_per_rank": E,
             "ms_per_step_rank_min": head.get("ms_per_step_rank_min", head["ms_per_step"]),
             "ms_per_step_rank_max": head.get("ms_per_step_rank_max", head["ms_per_step"])}
    out = None
    if rank == 0:
        kstep, kfused = kernel_names(wl["kind"], wl["n"])
        sfx = "_counter" if wl.get("rng") == "counter" else ""
        roof = r.roofline(head, kstep, "per_step" + sfx) if wl["algo"] else None
        if roof is not None:
            ceil = stream_ceiling()
            roof["measured_copy_GBs"], roof["measured_fill_GBs"] = ceil["copy"], ceil["fill"]
            if world == 1 and not a.no_live_traffic and a.rng == "mt19937":
                # the PMC bytes of the headline kernel from THIS run (two short child runs under rocprofv3 --pmc); the committed
                # constant (profiles/traffic.json) stays as `traffic_committed` and stands in when the passes cannot run
                bytes_live, note = live_traffic(wl, E, r.S)
                roof["traffic_committed"], roof["traffic_committed_source"] = roof["traffic"], roof["traffic_source"]
                if bytes_live is not None:
                    roof["traffic"] = int(round(bytes_live * E / float(r.S)))
                    roof["traffic_ratio"] = round(bytes_live / float(wl["algo"]), 3)
                    roof["traffic_source"] = note
                    roof["traffic_live"] = True
                else:
                    roof["traffic_live"] = False
                    roof["traffic_live_error"] = note
        out = {
            "metric": "agent-steps/sec", "value": head["value"], "unit": "agent-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[wl["kind"]], "data": "synthetic",
            "repeats": head["repeats"], "value_min": head["value_min"], "value_max": head["value_max"],
            "config": {"workload": wl["name"], "mode": "per_step: one launch per env-step and env slice", "envs_per_gpu": E,
                       "envs_per_launch": E // r.S, "agents": wl["n"], "global_envs": world * E,
                       "rng": "mt19937-numpy-compat" if a.rng == "mt19937" else "philox4x32-10 counter stream (NOT the reference's)",
                       "parallelism": "env-shard x%d, no collectives" % world, "streams_per_gpu": r.S,
                       "preroll_steps": PREROLL, "timed_seconds": head["timed_seconds"], "sanity": stats},
            "roofline": roof,
        }
        out["ranks"] = ranks
        if fused is not None:
            fr = r.roofline(fused, kfused, "fused" + sfx) if wl["algo"] else None
            out["fused"] = dict(fused, roofline=fr, note="ce_rollout_fused: %d steps per launch, env state resident on chip, every "
                                "step's obs / rewards / infos / features / done written to a %d-plane trajectory ring; whole launches "
                                "only (%d timed steps per repeat); results bit-identical to per_step (tests/test_fused_rollout_gpu.py)"
                                % (fused["steps_per_launch"], fused["trajectory_planes"], fused["steps"]))
    r.close()

    # the other BASELINE configs, same protocol: repeats of --config-steps steps (independent of --steps) until
    # --config-seconds of timed wall per mode (single-GPU workloads: rank 0's GPU only would idle the others, so every
    # rank runs its shard of them too and the line reports the whole-job value)
    custom = custom or a.rng != "mt19937"  # (a line on the engine's own stream carries the headline rows only)
    # --time-budget: the sections from here to the headline's CPU baseline are optional.  Once the run is `budget - reserve`
    # seconds old (reserve = what the required tail — the headline's oracle sample and its parity check — takes) the ones still
    # to come are skipped and listed, instead of letting a slow box push the whole line past whoever waits for it.  The clock is
    # MAX-reduced over the ranks, so every rank takes the same branch (the config rows hold barriers).
    reserve = 0.0 if (a.no_cpu_baseline or world > 1) else 2.0 * a.cpu_seconds + 30.0
    skipped = []

    def out_of_time(section, keep=None):
        if not a.time_budget:
            return False
        spent = group.max(time.monotonic() - T_START)
        if spent > a.time_budget - (reserve if keep is None else keep):
            skipped.append(section)
            return True
        return False

    if not a.no_configs and not custom:
        rows = []
        for key in ("C2", "C3", "C5", "C1"):
            if out_of_time("configs." + key):
                continue
            w = WORKLOADS[key]
            w_fused = bool(a.fused_steps) and w["kind"] in FUSED_KINDS
            rr = Runner(group, w, w["E"], a.config_steps, min(W, 20), a.streams, local_rank, a.fused_steps if w_fused else 0)
            m = rr.measure("per_step", min_repeats=3, min_seconds=a.config_seconds, max_repeats=100000)
            row = {"config": key, "workload": w["name"], "dtype": DTYPE[w["kind"]], "envs_per_gpu": w["E"], "agents": w["n"],
                   "value": m["value"], "value_min": m["value_min"], "value_max": m["value_max"], "unit": "agent-steps/s",
                   "ms_per_step": m["ms_per_step"], "repeats": m["repeats"], "steps": m["steps"],
                   "timed_seconds": m["timed_seconds"], "preroll_steps": rr.preroll,
                   "roofline": rr.roofline(m, kernel_names(w["kind"], w["n"])[0], "per_step_" + key)}
            if w_fused:
                f = rr.measure("fused", T=a.fused_steps, min_repeats=3, min_seconds=a.config_seconds, max_repeats=100000)
                froof = rr.roofline(f, kernel_names(w["kind"], w["n"])[1], "fused_" + key)
                row["fused"] = {"value": f["value"], "ms_per_step": f["ms_per_step"], "steps_per_launch": f["steps_per_launch"],
                                "steps": f["steps"], "repeats": f["repeats"], "launches_per_step": f["launches_per_step"],
                                "roofline_frac": froof["frac"], "traffic_ratio": froof["traffic_ratio"],
                                "roofline_frac_is": "algorithmic bytes / time: a throughput index for a resident-state kernel, which "
                                                    "moves traffic_ratio x those bytes (the state stays on chip between the steps of a launch)"}
            rows.append(row)
            rr.close()
        if out is not None:
            out["configs"] = rows
    def extra(fn, *args, **kw):
        """the sections beside the headline must never take the line down with them: a failure is reported in place"""
        try:
            return fn(*args, **kw)
        except Exception as exc:  # noqa: BLE001 (reported, not swallowed: in the line, with the traceback on stderr)
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}

    if out is not None and world == 1 and not custom:
        if not a.no_closed_loop and not out_of_time("closed_loop"):
            out["closed_loop"] = cl = extra(closed_loop, WORKLOADS["C4"], E, local_rank, a.streams)
            # the same loop cut into 2 / 4 slices (one graph replay per slice and tick: fewer slices = fewer host calls per
            # tick, more slices = more overlap on the device); `value` stays with the headline's slice count unless one of
            # these one-host-iteration-per-step rows beats it
            sweep = {}
            for S2 in (2, 4):
                if S2 == a.streams or "error" in cl or out_of_time("closed_loop.slices_sweep.%d" % S2):
                    continue
                r2 = extra(closed_loop, WORKLOADS["C4"], E, local_rank, S2, 0.5,
                           ("policy_eager", "policy_graph", "policy_graph16", "policy_graph_all", "inkernel_eager", "inkernel_sliced", "inkernel_graph_all"))
                sweep[str(S2)] = {m: {k: v[k] for k in ("value", "ms_per_step", "host_calls_per_step") if k in v}
                                  for m, v in r2.get("modes", {}).items()} if "error" not in r2 else r2
                if "error" not in r2 and r2["value"] > cl["value"]:
                    cl.update({k: r2[k] for k in ("value", "ms_per_step", "steps", "repeats", "timed_seconds", "host_calls_per_step",
                                                  "host_iterations_per_step", "issue", "slices")})
                if "error" not in r2 and (r2.get("best_with_separate_policy_kernel") or {}).get("value", 0) > \
                        (cl.get("best_with_separate_policy_kernel") or {}).get("value", 0):
                    cl["best_with_separate_policy_kernel"] = dict(r2["best_with_separate_policy_kernel"], slices=S2)
                if "error" not in r2 and (r2.get("inkernel_policy") or {}).get("value", 0) > (cl.get("inkernel_policy") or {}).get("value", 0):
                    cl["inkernel_policy"] = dict(r2["inkernel_policy"], slices=S2)
            cl["slices_sweep"] = sweep
        if not a.no_boundary and not out_of_time("boundary"):
            out["boundary"] = extra(boundary, WORKLOADS["C4"], E, local_rank)
        if not a.no_counter_rng and not out_of_time("counter_rng"):
            out["counter_rng"] = extra(counter_rng, group, WORKLOADS["C4"], a, local_rank)
    if out is not None:
        if not a.no_cpu_baseline and world == 1:
            def with_parity(target, w, seconds, **kw):
                cb = extra(cpu_baseline, w, seconds, **kw)
                st = cb.pop("_oracle_state", None)
                if st is not None and st["fields"] is not None:
                    target["parity_in_run"] = extra(parity_in_run, w, st, local_rank, a.streams, a.fused_steps)
                target["cpu_baseline"] = cb

            with_parity(out, dict(wl, E=E), a.cpu_seconds)
            for row in out.get("configs", []):  # the CPU path beside every GPU config, same E rule / seeds / actions
                if out_of_time("cpu_baseline." + row["config"], keep=0.0):
                    row["cpu_baseline"] = {"skipped": "--time-budget %g s" % a.time_budget}
                    continue
                with_parity(row, WORKLOADS[row["config"]], a.config_cpu_seconds, single_thread_s=0.0)
        out["time_budget"] = {"seconds": a.time_budget, "spent_s": round(time.monotonic() - T_START, 1), "skipped_sections": skipped}
        out["summary"] = summary(out)
        emit(out, a)
    group.close()


COMPACT_MAX_BYTES = 3072  # the driver parses the LAST stdout line; r05's 25 KB line came back `parsed: null`


def _sig(x, digits=5):
    """numbers of the compact line: 5 significant digits (floats), everything else untouched"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact(out, full_record=None):
    """The ONE stdout line: a standalone record of the headline row — the contract's keys, `roofline`, `cpu_baseline`,
    `parity_in_run`, and `summary` (one short row per other config) — without prose.  Everything else (per-mode tables, notes,
    samples, the config rows in full) is the full record, written to `full_record`."""
    cfg = out.get("config") or {}
    roof = out.get("roofline") or None
    cb = out.get("cpu_baseline") or None
    par = out.get("parity_in_run") or None
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "mode", "envs_per_gpu", "agents", "global_envs", "rng", "streams_per_gpu",
                                              "parallelism")}
    line["config"]["mode"] = "per_step"
    line["roofline"] = None if roof is None else {k: roof.get(k) for k in (
        "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "kernel", "algorithmic_bytes_per_env_step",
        "algorithmic_bytes_per_launch", "envs_per_launch", "launch_ms", "event_ms_per_step", "measured_copy_GBs", "measured_fill_GBs",
        "traffic_live")}
    if cb is not None and "error" not in cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model", "single_thread_value",
                                                       "python_reference_per_core")}
        ost = (par or {})
        line["cpu_baseline"]["sample"] = "envs 0..%s x %s steps of the same batch" % (
            (ost.get("envs") or 0) - 1 if ost.get("envs") else "?", ost.get("steps", "?"))
    else:
        line["cpu_baseline"] = cb
    if par is not None:
        line["parity_in_run"] = {k: par.get(k) for k in ("ok", "envs", "steps", "slices", "envs_per_launch", "error") if k in par}
    line["ranks_seen"] = (out.get("ranks") or {}).get("ranks_seen")
    line["summary"] = out.get("summary")
    line["full_record"] = full_record
    line = _sig(line)
    line["value"], line["ms_per_step"] = out.get("value"), out.get("ms_per_step")  # the two the driver cross-checks: unrounded
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_MAX_BYTES:  # never let the line outgrow the driver again: drop the digest before the record
        line["summary"] = {k: v for k, v in (line.get("summary") or {}).items() if k in ("C4", "parity_all_ok", "parity_legs_run", "skipped_for_time")}
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out, a):
    """full record -> a file (and stdout under --full, for the tools that read every row); the compact record is the LAST
    (by default the only) stdout line"""
    full = json.dumps(out)
    path = a.full_out
    try:
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as fh:
                fh.write(full + "\n")
    except OSError as exc:
        path = "unwritable (%s)" % type(exc).__name__
    if a.full:
        print(full, flush=True)
    print(compact(out, None if not path else os.path.relpath(path, ROOT) if not path.startswith("unwritable") else path), flush=True)


def summary(out):
    """compact digest of the line, emitted as its last key: [value in G agent-steps/s, roofline frac, fused frac] per config"""
    def g(x):
        return None if x is None else round(x / 1e9, 4)

    def f3(x):
        return None if x is None else round(x, 4)

    def row(r, fused):
        roof = r.get("roofline") or {}
        par = r.get("parity_in_run")
        return {"G": g(r.get("value")), "frac": f3(roof.get("frac")), "traffic_ratio": roof.get("traffic_ratio"),
                "fused_G": g((fused or {}).get("value")),
                "fused_frac": f3(((fused or {}).get("roofline") or {}).get("frac", (fused or {}).get("roofline_frac"))),
                "cpu_G": g((r.get("cpu_baseline") or {}).get("value")),
                "parity_ok": None if par is None else par.get("ok", False)}

    sm = {"C4": row(out, out.get("fused"))}
    for r in out.get("configs", []):
        sm[r["config"]] = row(r, r.get("fused"))
    cl, bd, cr = out.get("closed_loop") or {}, out.get("boundary") or {}, out.get("counter_rng") or {}
    sm["closed_loop_G"] = g(cl.get("value"))
    sm["closed_loop_issue"] = cl.get("issue")
    sm["closed_loop_host_calls_per_step"] = cl.get("host_calls_per_step")
    sm["closed_loop_inkernel_G"] = g((cl.get("inkernel_policy") or {}).get("value"))  # the toy policy inside the step kernel: NOT `value`
    sm["dict_M"] = None if "dict_protocol" not in bd else round(bd["dict_protocol"]["value"] / 1e6, 2)
    sm["dict_incl_action_dicts_M"] = None if "dict_protocol" not in bd else round(bd["dict_protocol"].get("value_incl_action_dicts", 0) / 1e6, 2)
    sm["tensor_G"] = g((bd.get("tensor_path") or {}).get("value"))
    sm["joint_global_tensor_G"] = g((((bd.get("joint") or {}).get("global")) or {}).get("tensor_value"))
    hb, lb = cr.get("headline_batch") or {}, cr.get("large_batch") or {}
    sm["counter_G"] = g((hb.get("counter") or {}).get("value"))
    sm["counter_frac"] = f3((hb.get("counter") or {}).get("roofline_frac"))
    sm["counter_fused_G"] = g(((hb.get("counter") or {}).get("fused") or {}).get("value"))
    sm["beyond_cache_frac"] = f3((lb.get("mt19937") or {}).get("roofline_frac"))
    sm["beyond_cache_counter_frac"] = f3((lb.get("counter") or {}).get("roofline_frac"))
    sm["counter_parity_ok"] = (cr.get("parity_in_run") or {}).get("ok")
    ran = [v["parity_ok"] for v in sm.values() if isinstance(v, dict) and v.get("parity_ok") is not None]
    if sm["counter_parity_ok"] is not None:
        ran.append(sm["counter_parity_ok"])
    sm["parity_legs_run"] = len(ran)
    sm["parity_all_ok"] = all(ran) if ran else None  # None = no parity leg ran (--no-cpu-baseline, world > 1): not a pass
    if (out.get("time_budget") or {}).get("skipped_sections"):  # a slow box: what the line does NOT carry, and why
        sm["skipped_for_time"] = out["time_budget"]["skipped_sections"]
    return sm


def main():
    a = parse()
    if a.gpus > 1 and not parallel.launched_by_torchrun():
        # `python bench.py --gpus N` by itself: N fresh rank processes on this node, one per GPU, started before this
        # process has touched the GPU (no exec over a HIP context); rank 0 prints the line
        sys.exit(parallel.spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus, timeout=a.launch_timeout))
    run_rank(a)


if __name__ == "__main__":
    main()
