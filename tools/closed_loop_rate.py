#!/usr/bin/env python3
"""Closed-loop rate of the headline workload: every step's actions are computed ON THE DEVICE from that step's
observations (torch, same stream), then handed to ce_step / ce_step_range as a device pointer.  One Python iteration per
env-step and slice: what a GPU-resident sampler pays (kernel launch from Python + the policy's own kernels), beside
bench.py's pre-supplied action planes.  With S > 1 the batch is S independent env slices, each with its own stream and
its own policy batch (double-buffered sampling): no joins, so one slice's policy and tail overlap another's step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contracts_amd.engine import BatchedEnv

E, n, A = 16384, 8, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
env.seed(seed0=73907)
env.reset()
obs = env.torch_tensors()["obs"]  # uint8 [E, n, 15, 15, 3] view of the engine's pitched buffer
g = torch.Generator(device="cuda").manual_seed(0)
W = torch.randn(15 * 15 * 3, A, device="cuda", dtype=torch.float16, generator=g)
acts = torch.zeros((E, n), dtype=torch.uint8, device="cuda")


def random_policy(b0, b1):
    acts[b0:b1].random_(0, A)


def linear_policy(b0, b1):
    x = obs[b0:b1].reshape((b1 - b0) * n, -1).to(torch.float16)
    acts[b0:b1].copy_((x @ W).argmax(dim=1).to(torch.uint8).view(b1 - b0, n))


for name, fn in (("random actions (torch, on device)", random_policy), ("linear policy on the observation", linear_policy)):
    for S in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        bounds = [(E * s // S, E * (s + 1) // S) for s in range(S)]

        def tick():
            for st, (b0, b1) in zip(streams, bounds):
                with torch.cuda.stream(st):
                    fn(b0, b1)
                    env.step_range_device(acts.data_ptr(), b0, b1 - b0, stream=st.cuda_stream)

        for _ in range(30):
            tick()
        torch.cuda.synchronize()
        steps = 300
        t0 = time.perf_counter()
        for _ in range(steps):
            tick()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("closed loop, %-36s %d slice(s): %6.1f us per step = %5.2f G agent-steps/s" % (
            name, S, dt / steps * 1e6, steps * E * n / dt / 1e9))
env.check_faults()
env.close()
