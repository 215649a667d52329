#!/usr/bin/env python3
"""Closed-loop rate of the headline workload (bench.py's `closed_loop` section on its own): every step's actions are
computed ON THE DEVICE from that step's observations (torch, same stream), then handed to ce_step_range as a device
pointer; three independent env slices on three streams; issued eagerly and as one hipGraph per slice.
    python tools/closed_loop_rate.py [slices]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = bench.closed_loop(bench.WORKLOADS["C4"], bench.WORKLOADS["C4"]["E"], 0, S)
print(json.dumps({k: v for k, v in out.items() if k != "policy"}, indent=1))
