#!/bin/bash
# MT19937 row write-back through the L2 (sc1) vs plain, single-step launches; parity first
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_rngwt; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
L=contracts_amd/csrc
timeout 900 tools/ab.sh 3 "C4 C3 C2 C1" $L/libcontracts_engine.so $L/libcontracts_engine_rngwtoff.so 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
