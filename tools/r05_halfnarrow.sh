#!/bin/bash
# bound of two-envs-per-wave packing in the grid kernels: -DCE_ABLATE_HALF_NARROW (every other env skips the <= 16-lane phases)
# against HEAD, on the VALU-bound rows: fused MT19937, counter per-step, counter fused; + the new vector-env test
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_halfnarrow; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_vector_env_gpu.py tests/test_policy_step_gpu.py -m gpu -x -q 2>&1 | tail -3
L=contracts_amd/csrc
timeout 900 tools/ab.sh 3 "C4:fused C4@counter C4:fused@counter C4 C3:fused C3@counter" $L/libcontracts_engine.so $L/libcontracts_engine_halfnarrow.so 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
