#!/bin/bash
# tools/pmc_quick.sh NAME "COUNTER1 COUNTER2 ..." : one PMC pass of the default bench, per-wave means of the step kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; shift
OUT=$R/gpurun_out/pmcq/$NAME; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 > $OUT/log.txt 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = {}
for row in csv.DictReader(open(f[0])):
    if "k_grid_step" in row["Kernel_Name"]:
        s, k = acc.get(row["Counter_Name"], (0.0, 0)); acc[row["Counter_Name"]] = (s + float(row["Counter_Value"]), k + 1)
w = 8192.0
for c, (s, k) in sorted(acc.items()):
    print("%-32s per-dispatch %14.1f   per-wave %10.2f" % (c, s / k, s / k / w))
PY
