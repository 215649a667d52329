#!/usr/bin/env python3
"""Quick A/B of per-step launches vs fused rollouts (ce_rollout_fused) on one GPU.

    python tools/fused_bench.py [--kind cleanup] [--agents 8] [--envs 16384] [--steps 512] [--T 8,16,32,64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="cleanup")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--preroll", type=int, default=300)
    ap.add_argument("--T", default="8,16,32,64")
    ap.add_argument("--planes", type=int, default=0, help="trajectory planes (0 = T)")
    ap.add_argument("--no-traj", action="store_true")
    ap.add_argument("--slices", default="1,2,3,4")
    a = ap.parse_args()
    import torch
    from contracts_amd.engine import BatchedEnv
    kind, n, E, K = a.kind, a.agents, a.envs, a.steps
    contract = {"cleanup": "cleanup", "harvest": "harvest_local", "selfdrive": "selfdrive_distprop",
                "harvest_features": "harvest_local", "cleanup_features": "cleanup"}[kind]
    env = BatchedEnv(kind, E, n, contract=contract, horizon=1000, auto_reset=True)
    env.seed(seed0=73907)
    env.reset()
    dt = torch.float32 if kind == "selfdrive" else torch.uint8
    acts = torch.empty((a.preroll + K, E, n), dtype=dt, device="cuda")
    env.synth_actions(73908, 0, a.preroll + K, acts.data_ptr())
    env.synchronize()
    esz = 4 if kind == "selfdrive" else 1
    plane = E * n * esz
    streams = [torch.cuda.Stream() for _ in range(8)]
    handles = [s.cuda_stream for s in streams]
    env.rollout_device(acts.data_ptr(), a.preroll, handles[:3])
    torch.cuda.synchronize()
    state = env.state_dict()
    out = {}

    def timed(fn):
        env.load_state_dict(state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return E * n * K / (time.perf_counter() - t0)

    base = acts.data_ptr() + a.preroll * plane
    for S in (1, 3):
        out["per_step_%dstream" % S] = timed(lambda: env.rollout_device(base, K, handles[:S] if S > 1 else None))
    for T in [int(x) for x in a.T.split(",")]:
        traj = None if a.no_traj else env.alloc_trajectory(a.planes or T)
        for S in [int(x) for x in a.slices.split(",")]:
            out["fused_T%d_S%d" % (T, S)] = timed(lambda: env.rollout_fused(base, K, T, traj, handles[:S] if S > 1 else None))
        del traj
    print(json.dumps({k: round(v / 1e9, 4) for k, v in out.items()}))
    env.close()


if __name__ == "__main__":
    main()
