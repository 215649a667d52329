#!/usr/bin/env python3
"""Runs ON the GPU box, over the raw kernel trace of one profiled bench.py run: per (kernel, launch size) statistics.

bench.py steps five workloads with the same few kernel names (k_grid_step<0> serves both C4 and C2, every pre-roll and
warm-up launch included), so rocprofv3's own `--stats` table pools them.  The launch size separates them: an env slice
of C4 is 5461 / 5462 workgroups, of C2 1365 / 1366, and so on.  Output: one CSV row per (kernel, grid) with calls,
total / mean / min / max duration — the per-config launch duration the roofline figure is checked against.

    python3 tools/kernel_trace_by_config.py <dir with *kernel_trace.csv> <out.csv>
"""
import csv
import glob
import os
import sys

CONFIGS = {  # workgroups of one env slice (3 slices per rank) -> BASELINE config
    ("k_grid_step<0, 8, 0>", 5461): "C4", ("k_grid_step<0, 8, 0>", 5462): "C4", ("k_grid_rollout<0, 8, 0>", 5461): "C4 fused", ("k_grid_rollout<0, 8, 0>", 5462): "C4 fused",
    ("k_grid_step<0, 4, 0>", 1365): "C2", ("k_grid_step<0, 4, 0>", 1366): "C2", ("k_grid_rollout<0, 4, 4>", 1365): "C2 fused", ("k_grid_rollout<0, 4, 4>", 1366): "C2 fused",
    ("k_grid_step<1, 8, 0>", 5461): "C3", ("k_grid_step<1, 8, 0>", 5462): "C3", ("k_grid_rollout<1, 7, 8>", 5461): "C3 fused", ("k_grid_rollout<1, 7, 8>", 5462): "C3 fused",
    ("k_sd_step<4>", 683): "C5", ("k_sd_rollout<4>", 683): "C5 fused",
    ("k_feat_step_quad", 1366): "C1", ("k_feat_step_quad", 1365): "C1", ("k_feat_rollout_quad", 1366): "C1 fused", ("k_feat_rollout_quad", 1365): "C1 fused",
    ("k_feat_step<1, 2>", 5461): "C1 (one wave per env)", ("k_feat_step<1, 2>", 5462): "C1 (one wave per env)",
    ("k_feat_rollout<1, 2>", 5461): "C1 fused (one wave per env)", ("k_feat_rollout<1, 2>", 5462): "C1 fused (one wave per env)",
}


def label(kernel, blocks):
    """BASELINE config of a (kernel, workgroups) pair; the grid kernels' last template argument (the custom-layout flag)
    is not part of the mapping's keys"""
    import re
    return CONFIGS.get((kernel, blocks)) or CONFIGS.get((re.sub(r", (false|true)>$", ">", kernel), blocks), "")


def short(name):
    n = name.replace("void ce::", "").replace("ce::", "")
    return n.split("(")[0].strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            wg = int(row.get("Workgroup_Size") or row.get("Workgroup_Size_X") or 64)
            grid = int(row.get("Grid_Size") or row.get("Grid_Size_X") or 0)
            blocks = grid // max(wg, 1)
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            a = acc.setdefault((k, blocks), [0, 0, 1 << 62, 0])
            a[0] += 1
            a[1] += d
            a[2] = min(a[2], d)
            a[3] = max(a[3], d)
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "workgroups", "config", "calls", "total_ns", "mean_ns", "min_ns", "max_ns"])
        for (k, blocks), (calls, tot, mn, mx) in rows:
            w.writerow([k, blocks, label(k, blocks), calls, tot, round(tot / calls, 1), mn, mx])
    print("kernel_trace_by_config: %d dispatches of %d (kernel, size) pairs from %d file(s) -> %s"
          % (sum(v[0] for v in acc.values()), len(acc), len(files), dst))


if __name__ == "__main__":
    main()
