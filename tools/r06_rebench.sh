#!/bin/bash
# re-record the two bench lines of the round (driver command, default command) without touching the kernel profile set: for changes
# above the C-ABI only (the library's kernels_sha16 must still be the profile set's, bench.py checks and says so in the record)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
cd $R
SECONDS=0; timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_driver_full.json > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "driver command: ${SECONDS}s"; wc -c $OUT/bench_driver.json; tail -c 2400 $OUT/bench_driver.json; echo
timeout 1200 python3 bench.py --full-out $OUT/bench_default_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; echo
