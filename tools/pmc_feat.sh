#!/bin/bash
# tools/pmc_feat.sh : instruction counters (per wave) and durations of the feature-env step kernels, runs ON the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcfeat; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/tools/pmc_feat_run.py > $OUT/log.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/pmc_feat_run.py >> $OUT/log.txt 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)
acc = {}
for row in csv.DictReader(open(f[0])):
    if "k_feat_step" in row["Kernel_Name"]:
        key = (row["Kernel_Name"][:40], row["Counter_Name"])
        s, k = acc.get(key, (0.0, 0)); acc[key] = (s + float(row["Counter_Value"]), k + 1)
for (kn, c), (s, k) in sorted(acc.items()):
    w = acc[(kn, "SQ_WAVES")][0] / acc[(kn, "SQ_WAVES")][1]
    print("%-42s %-20s per-wave %10.2f" % (kn, c, s / k / w))
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_feat_step" in row["Name"]:
            print(row["Name"][:60], row["Calls"], row["AverageNs"])
PY
