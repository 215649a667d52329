#!/bin/bash
# round-5 evidence run on the GPU box: the full GPU suite, the driver's bench command and the default one, then the profile sets
# (tools/collect_profiles.sh / collect_counter_profiles.sh refuse unless the library was built from committed sources)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}; OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1; tail -4 $OUT/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 1200 $OUT/bench_driver.json; echo
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; echo
timeout 2400 tools/collect_profiles.sh $TAG 2>&1 | tail -5
timeout 1200 tools/collect_counter_profiles.sh $TAG 2>&1 | tail -3
