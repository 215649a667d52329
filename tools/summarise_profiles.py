"""Turns gpurun_out/prof (tools/collect_profiles.sh) into the committed summaries under profiles/:
  r01_kernel_stats.csv   rocprofv3 --kernel-trace --stats kernel summary
  r01_pmc_summary.json   per-dispatch / per-wave counter means of the step kernel + HBM traffic
  traffic.json           HBM bytes per launch (read by bench.py into roofline.traffic)
Usage: python tools/summarise_profiles.py [gpurun_out/prof] [round-tag]"""
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
KERNEL = "k_grid_step"
ALGO = 7235


def find(sub, pat):
    got = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    if not got:
        raise SystemExit("missing %s/%s" % (sub, pat))
    return max(got, key=os.path.getmtime)  # gpurun merges into gpurun_out/: older runs' files may still be there


def counter_means(sub):
    acc, meta = {}, {}
    with open(find(sub, "*counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if KERNEL not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            s, k = acc.get(c, (0.0, 0))
            acc[c] = (s + float(row["Counter_Value"]), k + 1)
            if not meta:
                meta = {k2: row[k2] for k2 in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")
                        if k2 in row}
    return {c: s / k for c, (s, k) in acc.items()}, meta


per, meta = {}, {}
for sub in ("pmc1", "pmc2", "fetch", "write"):
    m, mm = counter_means(sub)
    per.update(m)
    meta = meta or mm
waves = per["SQ_WAVES"]  # mean over the dispatches of the step kernel (slices differ by one env)
envs = int(round(waves))
fetch_b = per["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE reports half of wide coalesced reads
write_b = per["WRITE_SIZE"] * 1024
traffic = int(fetch_b + write_b)

stats_path = find("kt", "*kernel_stats.csv")
shutil.copy(stats_path, "profiles/%s_kernel_stats.csv" % tag)
kt = {}
with open(stats_path) as f:
    for row in csv.DictReader(f):
        if KERNEL in row["Name"]:
            kt = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "percentage": float(row["Percentage"]),
                  "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
line = open(os.path.join(src, "kt.json")).read().strip().splitlines()[-1]
try:
    bench = json.loads(line)
    kt["bench_kernel_ms_hip_events"] = bench["roofline"]["kernel_ms"]
    kt["bench_value_under_profiler"] = bench["value"]
except Exception:
    pass

out = {
    "round": int(tag[1:]),
    "command": "tools/collect_profiles.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline ; "
               "PMC: separate `rocprofv3 --pmc <set>` passes of `python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline` "
               "(bench default: the rank's 16384 envs as 3 contiguous slices on 3 HIP streams)",
    "kernel": "ce::k_grid_step<0> (cleanup_new n=8 + CleanupContract, ~%d envs per launch, one 64-lane workgroup per env, 3 launches in flight)" % envs,
    "per_dispatch_mean": per,
    "dispatch_meta": meta,
    "per_wave": {c: round(v / waves, 1) for c, v in per.items() if c.startswith("SQ_")},
    "hbm_traffic": {"FETCH_SIZE_KB": per["FETCH_SIZE"], "WRITE_SIZE_KB": per["WRITE_SIZE"],
                    "note": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; "
                            "WRITE_SIZE taken as is; per launch of %d envs" % envs,
                    "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": ALGO * envs},
    "kernel_trace": kt,
}
json.dump(out, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
json.dump({"kind": "cleanup", "agents": 8, "hbm_bytes_per_env_step": traffic / envs, "measured_envs_per_launch": envs,
           "hbm_bytes_per_launch": traffic,
           "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH doubled per the "
                     "gfx950 correction)" % tag}, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({"traffic": traffic, "algo": ALGO * envs, "kt": kt, "per_wave": out["per_wave"]}, indent=1))
