#!/bin/bash
# full GPU suite + the driver's bench command (+ write-through A/B at the headline shapes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_suite; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1
tail -8 $OUT/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
tail -c 3000 $OUT/bench_driver.json; tail -3 $OUT/bench_driver.err
for i in 1 2; do
  for thr in 0 100000000000; do
    CE_OBS_WT_MAX_BYTES=$thr timeout 600 python3 tools/rate.py C2 C4 C3 2>&1 | grep -v amdgpu.ids | sed "s/^/wt<=$thr /"
    CE_OBS_WT_MAX_BYTES=$thr RATE_PREROLL=12000 timeout 600 python3 tools/rate.py C5 2>&1 | grep -v amdgpu.ids | sed "s/^/wt<=$thr /"
  done
done > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
