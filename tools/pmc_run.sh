#!/bin/bash
# tools/pmc_run.sh NAME KERNEL_SUBSTR ENV_STEPS "CTRS pass 1" ["CTRS pass 2" ...] -- driver args
# One rocprofv3 --pmc pass per counter group over tools/pmc_driver.py; prints, for the dispatches whose kernel name
# contains KERNEL_SUBSTR, each counter summed over those dispatches and divided by ENV_STEPS (= envs x measured steps),
# i.e. per env-step.  Runs ON the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; KSUB=$2; ENVSTEPS=$3; shift 3
CG=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do CG+=("$1"); shift; done
shift
OUT=$R/gpurun_out/pmc/$NAME; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${CG[@]}"; do
  rocprofv3 --pmc $g --output-format csv -d $OUT/p$i -- python3 $R/tools/pmc_driver.py "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $R
python3 - "$OUT" "$KSUB" "$ENVSTEPS" "$NAME" <<'PY'
import csv, glob, json, sys
out, ksub, envsteps, name = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
acc, disp, passes = {}, {}, {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            disp[c] = disp.get(c, 0) + 1
            seen.add(c)
    for c in seen:
        passes[c] = passes.get(c, 0) + 1
for c in acc:  # a counter collected in several passes: the mean of the passes
    acc[c] /= passes[c]
    disp[c] //= passes[c]
res = {c: acc[c] / envsteps for c in sorted(acc)}
for c in sorted(acc):
    print("%-28s dispatches %6d   per env-step %12.2f" % (c, disp[c], res[c]))
json.dump({"name": name, "kernel": ksub, "env_steps": envsteps, "per_env_step": res, "dispatches": disp},
          open(out + "/summary.json", "w"), indent=1)
PY
find $OUT -name '*_agent_info.csv' -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
