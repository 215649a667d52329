#!/bin/bash
# tools/kt_run.sh NAME -- driver args : rocprofv3 --kernel-trace --stats over tools/pmc_driver.py; prints the kernel stats table
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; shift; shift
OUT=$R/gpurun_out/kt/$NAME; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/pmc_driver.py "$@" > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
head -12 $f | cut -c1-250
find $OUT -name '*_agent_info.csv' -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
