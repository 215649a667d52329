#!/bin/bash
# interleaved A/B on the GPU box: tools/r06_ab.sh NAME ROUNDS "SPECS" libA.so libB.so ... -> gpurun_out/r06_ab/NAME.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; NAME=$1; N=$2; SPECS=$3; shift 3
OUT=$R/gpurun_out/r06_ab; mkdir -p $OUT; cd $R
timeout 1500 tools/ab.sh $N "$SPECS" "$@" 2>&1 | grep -v amdgpu.ids > $OUT/$NAME.txt
cat $OUT/$NAME.txt
