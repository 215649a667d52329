#!/usr/bin/env python3
"""Where a closed-loop tick's time goes (bench.py closed_loop): per-tick wall time and the host time spent inside
the replay / launch calls, for
    step_only_graph     one hipGraph per slice holding just the step launch (fixed action plane): the floor of graph replays
    policy_graph        torch.add + ce_step_policy per slice and tick (bench.py's policy_graph)
    step_only_eager     ce_step_range per slice from Python, no graph
    policy_eager        torch.add + ce_step_policy per slice from Python
    python tools/closed_loop_probe.py [slices]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from contracts_amd.engine import BatchedEnv  # noqa: E402

E, n, S = 16384, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 3
env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
env.seed(seed0=73907)
env.reset()
obs = env.torch_tensors()["obs"]
ahead = obs[:, :, 6, 7, 1]
noise = torch.randint(0, 256, (E, n), dtype=torch.uint8, device="cuda")
acts = torch.randint(0, 8, (E, n), dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(S)]
bounds = [(E * s // S, E * (s + 1) // S) for s in range(S)]
pre = torch.empty((300, E, n), dtype=torch.uint8, device="cuda")
env.synth_actions(73908, 0, 300, pre.data_ptr())
env.rollout_device(pre.data_ptr(), 300, [st.cuda_stream for st in streams])
torch.cuda.synchronize()


def step_only(st, b0, b1):
    env.step_range_device(acts.data_ptr(), b0, b1 - b0, stream=st.cuda_stream)


def policy(st, b0, b1):
    torch.add(ahead[b0:b1], noise[b0:b1], out=noise[b0:b1])
    env.step_policy_device(noise.data_ptr(), "bytes", b0, b1 - b0, stream=st.cuda_stream)


def run(name, tick, K=600):
    for _ in range(60):
        tick(None)
    torch.cuda.synchronize()
    reps = []
    for _ in range(4):
        calls = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            tick(calls)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        reps.append(((t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6, statistics.median(calls) * 1e6 if calls else 0.0))
    w, h, c = sorted(reps)[len(reps) // 2]
    print("%-18s S=%d  %6.1f us/tick wall  %6.1f us/tick host issue  median call %5.1f us   %.2f G agent-steps/s" % (name, S, w, h, c, E * n / w / 1e3), flush=True)


for name, fn in (("step_only", step_only), ("policy", policy)):
    def eager(calls, fn=fn):
        for st, (b0, b1) in zip(streams, bounds):
            with torch.cuda.stream(st):
                t = time.perf_counter()
                fn(st, b0, b1)
                if calls is not None:
                    calls.append(time.perf_counter() - t)
    run(name + "_eager", eager)
    graphs = []
    for st, (b0, b1) in zip(streams, bounds):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn(st, b0, b1)
        graphs.append(g)
    torch.cuda.synchronize()

    def replay(calls, graphs=graphs):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                t = time.perf_counter()
                g.replay()
                if calls is not None:
                    calls.append(time.perf_counter() - t)
    run(name + "_graph", replay)
env.close()
