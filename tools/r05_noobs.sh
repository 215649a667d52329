#!/bin/bash
# what the observation STORES cost the resident / counter kernels (-DCE_ABLATE_OBSSTORE: pixels computed, not written; timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_noobs; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_vector_env_gpu.py -m gpu -x -q 2>&1 | tail -3
L=contracts_amd/csrc
timeout 900 tools/ab.sh 2 "C4:fused C4@counter C4:fused@counter C4 C3:fused C2 C2:fused" $L/libcontracts_engine.so $L/libcontracts_engine_noobs.so 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
