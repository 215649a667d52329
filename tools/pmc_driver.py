#!/usr/bin/env python3
"""Workload for PMC passes (tools/pmc_run.sh): steps one handle either with per-step launches or fused rollouts.

    python3 tools/pmc_driver.py --mode step|fused [--kind cleanup] [--agents 8] [--envs 16384] [--steps 64] [--T 64]
                                [--preroll 300] [--streams 3] [--rng counter]
The pre-roll runs with the other mode's kernel (so the profiled kernel only sees the measured, steady-state steps).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="step")
    ap.add_argument("--kind", default="cleanup")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--preroll", type=int, default=300)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--rng", default="mt19937", help="mt19937 | counter (grid kinds)")
    a = ap.parse_args()
    import torch
    from contracts_amd.engine import BatchedEnv
    contract = {"cleanup": "cleanup", "harvest": "harvest_local", "selfdrive": "selfdrive_distprop",
                "harvest_features": "harvest_local", "cleanup_features": "cleanup"}[a.kind]
    env = BatchedEnv(a.kind, a.envs, a.agents, contract=contract, horizon=1000, auto_reset=True, rng=a.rng)
    env.seed(seed0=73907)
    env.reset()
    dt = torch.float32 if a.kind == "selfdrive" else torch.uint8
    acts = torch.empty((a.preroll + a.steps, a.envs, a.agents), dtype=dt, device="cuda")
    env.synth_actions(73908, 0, a.preroll + a.steps, acts.data_ptr())
    env.synchronize()
    plane = a.envs * a.agents * (4 if a.kind == "selfdrive" else 1)
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    handles = [s.cuda_stream for s in streams] if a.streams > 1 else None
    if a.preroll:  # with the OTHER mode's kernel, so that the profiled kernel only sees the measured steps
        if a.mode == "step":
            env.rollout_fused(acts.data_ptr(), a.preroll, 50, None, handles)
        else:
            env.rollout_device(acts.data_ptr(), a.preroll, handles)
    torch.cuda.synchronize()
    base = acts.data_ptr() + a.preroll * plane
    if a.mode == "fused":
        traj = env.alloc_trajectory(min(a.T, a.steps))
        env.rollout_fused(base, a.steps, a.T, traj, handles)
    else:
        env.rollout_device(base, a.steps, handles)
    torch.cuda.synchronize()
    env.check_faults()
    env.close()


if __name__ == "__main__":
    main()
