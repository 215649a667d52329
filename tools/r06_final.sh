#!/bin/bash
# round-6 evidence run on the GPU box: the full GPU suite, the driver's bench command (compact stdout line + full record) and the
# default one, the profile sets (tools/collect_profiles.sh / collect_counter_profiles.sh refuse unless the library was built from
# committed sources) and the one-launch scalar-pipe PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
cd $R
if [ "$2" != "notests" ]; then timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt; fi
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/bench_driver_full.json > $OUT/bench_driver.json 2> $OUT/bench_driver.err; wc -c $OUT/bench_driver.json; tail -c 2400 $OUT/bench_driver.json; echo
timeout 1200 python3 bench.py --full-out $OUT/bench_default_full.json > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; echo
timeout 2400 tools/collect_profiles.sh $TAG 2>&1 | tail -5
timeout 1200 tools/collect_counter_profiles.sh $TAG 2>&1 | tail -3
timeout 900 tools/r06_scalar_pipe.sh > $OUT/scalar_pipe.txt 2>&1; tail -45 $OUT/scalar_pipe.txt
