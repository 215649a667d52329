#!/bin/bash
# write-through (sc1) vs nontemporal observation stores of single-step launches by batch size (CE_OBS_WT_MAX_BYTES=0 = never),
# after the parity tests of the rebuilt library
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_wt_sweep; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fused_rollout_gpu.py tests/test_vector_env_gpu.py tests/test_cabi.py -m gpu -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
for i in 1 2 3; do
  for thr in 0 100000000; do
    CE_OBS_WT_MAX_BYTES=$thr timeout 600 python3 tools/rate.py C2 C4 C3 cleanup,8,32768 cleanup,8,65536 cleanup,8,131072 harvest,8,65536 2>&1 | grep -v amdgpu.ids | sed "s/^/wt<=$thr /"
    CE_OBS_WT_MAX_BYTES=$thr RATE_PREROLL=12000 timeout 600 python3 tools/rate.py C5 selfdrive,4,131072 2>&1 | grep -v amdgpu.ids | sed "s/^/wt<=$thr /"
  done
done > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
