#!/usr/bin/env python3
"""Closed-loop rate of the headline workload: every step's actions are computed ON THE DEVICE from that step's
observations (a small fixed linear policy over the 15 x 15 x 3 crop, argmax over the action logits — torch, same stream),
then handed to ce_step as a device pointer.  One Python iteration per env-step: what a GPU-resident sampler pays
(kernel launch from Python + the policy's own kernels), beside bench.py's pre-supplied action planes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contracts_amd.engine import BatchedEnv

E, n, A = 16384, 8, 8
env = BatchedEnv("cleanup", E, n, contract="cleanup", horizon=1000, auto_reset=True)
env.seed(seed0=73907)
env.reset()
t = env.torch_tensors()
obs = t["obs"]  # uint8 [E, n, 15, 15, 3] view of the engine's pitched buffer
g = torch.Generator(device="cuda").manual_seed(0)
W = torch.randn(15 * 15 * 3, A, device="cuda", dtype=torch.float16, generator=g)
acts = torch.zeros((E, n), dtype=torch.uint8, device="cuda")


def policy():
    x = obs.reshape(E * n, -1).to(torch.float16)
    acts.copy_((x @ W).argmax(dim=1).to(torch.uint8).view(E, n))


for name, fn in (("random actions (torch.randint on device)", lambda: acts.random_(0, A, generator=g)), ("linear policy on the observation", policy)):
    for _ in range(30):
        fn()
        env.step_device(acts.data_ptr())
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
        env.step_device(acts.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("closed loop, %-44s %6.1f us per step = %5.2f G agent-steps/s" % (name, dt / steps * 1e6, steps * E * n / dt / 1e9))
env.check_faults()
print("apples eaten per env so far: %.2f" % float(env.download("int_metrics")[:, 0].mean()))
env.close()
