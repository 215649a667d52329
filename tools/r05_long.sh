#!/bin/bash
# write-through view stores against nontemporal in the LONG bench command (1000 steps per repeat: 177 MB of action planes
# resident beside the 175 MB handle) and in the closed loop; + the two new / fixed tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_long; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_write_through_gpu.py tests/test_bench_multirank_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for thr in 0 1000000000000; do
    CE_OBS_WT_MAX_BYTES=$thr timeout 600 python3 bench.py --no-configs --no-boundary --no-counter-rng --no-cpu-baseline > $OUT/b_${thr}_${i}.json 2>/dev/null
    python3 - $OUT/b_${thr}_${i}.json $thr <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cl = d["closed_loop"]
print("wt<=%s  per_step %.3f G  fused %.3f G  closed %.3f G (%s)  separate %.3f G  modes %s" % (
    sys.argv[2], d["value"] / 1e9, d["fused"]["value"] / 1e9, cl["value"] / 1e9, cl["issue"],
    cl["best_with_separate_policy_kernel"]["value"] / 1e9, {k: round(v.get("value", 0) / 1e9, 2) for k, v in cl["modes"].items()}))
PY
  done
done
