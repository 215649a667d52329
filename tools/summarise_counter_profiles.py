#!/usr/bin/env python3
"""gpurun_out/prof_<TAG>_counter (tools/collect_counter_profiles.sh) -> profiles/<TAG>_counter_rng.json

    python tools/summarise_counter_profiles.py gpurun_out/prof_r03_counter r03
"""

import csv, json, subprocess, sys
out, tag = sys.argv[1], sys.argv[2]
prov = json.load(open(out + "/provenance.json"))  # written by the collector: the commit the library was built from
if subprocess.run(["git", "diff", "--quiet", prov["git_head"], "--", "contracts_amd/csrc", "include"]).returncode != 0 and "--allow-stale" not in sys.argv:
    sys.exit("summarise_counter_profiles: refused — the kernel sources differ from %s, the commit this set was measured on" % prov["git_head"][:12])
ALGO = {"c4_step": 7235, "c4_fused": 7235, "c3_step": 7313}
res = {"how": "tools/collect_counter_profiles.sh: separate rocprofv3 --pmc passes of tools/pmc_driver.py --rng counter (64 measured steps, 16 384 envs, "
              "three slices); FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md); kernel trace of a 400-step per-step run",
       "provenance": prov, "kernels": {}}
for k, algo in ALGO.items():
    s = json.load(open("%s/pmc_%s.json" % (out, k)))
    per = s["per_env_step"]
    row = {"per_env_step": per, "dispatches": s["dispatches"]}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        hb = per["FETCH_SIZE"] * 2048 + per["WRITE_SIZE"] * 1024
        row["hbm_bytes_per_env_step"] = hb
        row["algorithmic_bytes_per_env_step"] = algo
        row["ratio_to_algorithmic"] = hb / algo
    res["kernels"]["counter_" + k] = row
rows = [r for r in csv.DictReader(open(out + "/kernel_stats.csv")) if "k_grid_step" in r["Name"]]
res["kernel_trace"] = [{"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in rows]
json.dump(res, open("profiles/%s_counter_rng.json" % tag, "w"), indent=1)
# bench.py --rng counter looks its `roofline.traffic` up here
tj = json.load(open("profiles/traffic.json"))
for key, k in (("per_step_counter", "counter_c4_step"), ("fused_counter", "counter_c4_fused")):
    tj[key] = {"kind": "cleanup", "agents": 8, "hbm_bytes_per_env_step": res["kernels"][k]["hbm_bytes_per_env_step"],
               "source": "profiles/%s_counter_rng.json (%s; FETCH_SIZE doubled per the gfx950 correction)" % (tag, k)}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_env_step"), v["per_env_step"].get("SQ_INSTS_VALU"), v["per_env_step"].get("SQ_INSTS_SALU")) for k, v in res["kernels"].items()}))
print(res["kernel_trace"])
