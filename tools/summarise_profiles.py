"""Turns gpurun_out/prof_<tag> (tools/collect_profiles.sh) into the committed summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats kernel summary of the default bench command
  <tag>_kernel_stats_by_config.csv   the same trace split by launch size: one row per (kernel, BASELINE config)
  <tag>_pmc_summary.json   per env-step counter values of the per-step and the fused kernels + HBM traffic
  <tag>_bench_line.json    the bench line of the profiled run (slower than an unprofiled one)
  traffic.json             PMC HBM bytes per env-step, read by bench.py into roofline.traffic
Usage: python tools/summarise_profiles.py [gpurun_out/prof_r02] [r02]"""
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r03"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
ALGO = {"step": 7235, "fused": 7235, "sd_step": 863, "sd_fused": 863, "c3_step": 7313, "c3_fused": 7313, "c2_step": 4211,
        "c2_fused": 4211, "c1_step": 887, "c1_fused": 887}
KIND = {"step": ("cleanup", 8), "fused": ("cleanup", 8), "sd_step": ("selfdrive", 4), "sd_fused": ("selfdrive", 4),
        "c3_step": ("harvest", 8), "c3_fused": ("harvest", 8), "c2_step": ("cleanup", 4), "c2_fused": ("cleanup", 4),
        "c1_step": ("harvest_features", 2), "c1_fused": ("harvest_features", 2)}
TRAFFIC_KEY = {"step": "per_step", "fused": "fused", "sd_step": "per_step_C5", "sd_fused": "fused_C5", "c3_step": "per_step_C3",
               "c3_fused": "fused_C3", "c2_step": "per_step_C2", "c2_fused": "fused_C2", "c1_step": "per_step_C1", "c1_fused": "fused_C1"}

# provenance: the set was collected with a library built from committed sources (tools/collect_profiles.sh refuses otherwise);
# here, on the build box, refuse to write profiles/ if the kernel sources have moved since (a stale set is not evidence about HEAD)
import subprocess
prov = json.load(open(os.path.join(src, "provenance.json")))
changed = subprocess.run(["git", "diff", "--quiet", prov["git_head"], "--", "contracts_amd/csrc", "include"]).returncode != 0
if changed and "--allow-stale" not in sys.argv:
    sys.exit("summarise_profiles: refused — the kernel sources differ from %s, the commit this set was measured on "
             "(re-collect at HEAD, or pass --allow-stale to file it under that commit's name)" % prov["git_head"][:12])

stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(max(stats, key=os.path.getmtime), "profiles/%s_kernel_stats.csv" % tag)
by_cfg = os.path.join(src, "kernel_stats_by_config.csv")
if os.path.exists(by_cfg):  # the same trace split by launch size: one row per (kernel, config)
    import csv
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_trace_by_config import label  # (labels re-applied here: the mapping may be newer than the run's snapshot)
    rows = list(csv.reader(open(by_cfg)))
    for r in rows[1:]:
        r[2] = label(r[0], int(r[1]))
    with open("profiles/%s_kernel_stats_by_config.csv" % tag, "w", newline="") as fh:
        csv.writer(fh).writerows(rows)
full = os.path.join(src, "kt_bench_full.json")  # round 6: stdout carries the compact record, the full one goes to --full-out
line = [ln for ln in open(full if os.path.exists(full) else os.path.join(src, "kt_bench_line.json")).read().splitlines() if ln.startswith("{")][-1]
json.dump(json.loads(line), open("profiles/%s_bench_line.json" % tag, "w"), indent=1)

out = {"round": tag, "command": "tools/collect_profiles.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 400 --warmup 20 "
       "--no-cpu-baseline --no-closed-loop --no-boundary --no-counter-rng --min-seconds 0.2 ; PMC: separate `rocprofv3 --pmc <set>` passes of tools/pmc_driver.py (64 measured steps after "
       "a 300-step pre-roll run with the other mode's kernel), summed over the dispatches of the kernel and divided by envs x steps",
       "provenance": prov, "kernels": {}}
traffic = {}
for key in ALGO:
    path = os.path.join(src, "pmc_%s.json" % key)
    if not os.path.exists(path):
        continue
    s = json.load(open(path))
    per = s["per_env_step"]
    row = {"kernel": s["kernel"], "env_steps": s["env_steps"], "per_env_step": per, "dispatches": s["dispatches"]}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        # rocprofv3 reports KB; gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B -> doubled
        # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is
        fetch_b, write_b = per["FETCH_SIZE"] * 1024 * 2, per["WRITE_SIZE"] * 1024
        row["hbm"] = {"fetch_bytes_per_env_step": fetch_b, "write_bytes_per_env_step": write_b,
                      "hbm_bytes_per_env_step": fetch_b + write_b, "algorithmic_bytes_per_env_step": ALGO[key],
                      "ratio_to_algorithmic": (fetch_b + write_b) / ALGO[key]}
        tk = TRAFFIC_KEY[key]
        traffic[tk] = {"kind": KIND[key][0], "agents": KIND[key][1], "hbm_bytes_per_env_step": fetch_b + write_b,
                       "source": "profiles/%s_pmc_summary.json (%s; FETCH_SIZE doubled per the gfx950 correction)" % (tag, s["kernel"])}
    if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_ACTIVE_INST_LDS"):
        row["lds_conflict_share_of_lds_active"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_ACTIVE_INST_LDS"]
    if "SQ_WAVE_CYCLES" in per:
        row["wait_inst_any_share"] = per.get("SQ_WAIT_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]
    out["kernels"][key] = row
json.dump(out, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
try:  # the counter-RNG rows belong to tools/summarise_counter_profiles.py: kept
    traffic.update({k: v for k, v in json.load(open("profiles/traffic.json")).items() if k.endswith("_counter")})
except (OSError, ValueError):
    pass
traffic["_provenance"] = dict(prov, profile_set=tag)
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v.get("hbm", {}).get("hbm_bytes_per_env_step") for k, v in out["kernels"].items()}))
