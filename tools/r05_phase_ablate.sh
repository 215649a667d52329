#!/bin/bash
# which WIDE phase the resident / counter grid kernels wait for: builds that leave one phase out (timing only, wrong results):
# -DCE_ABLATE_FEATURES (feature scan), -DCE_ABLATE_SHUFFLE (the 119-entry waste-list shuffle), -DCE_ABLATE_MOVES (update_moves)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_phase_ablate; mkdir -p $OUT
cd $R
L=contracts_amd/csrc
timeout 1500 tools/ab.sh 2 "C4:fused C4@counter C4 C2 C2:fused" $L/libcontracts_engine.so $L/libcontracts_engine_ab_FEATURES.so $L/libcontracts_engine_ab_SHUFFLE.so $L/libcontracts_engine_ab_MOVES.so 2>&1 | grep -v amdgpu.ids > $OUT/ab.txt
cat $OUT/ab.txt
