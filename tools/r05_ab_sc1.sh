#!/bin/bash
# A/B: observation stores write-through (sc1, -DCE_OBS_SC1) vs nontemporal, per-step and fused, every config with views
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_ab_sc1; mkdir -p $OUT
cd $R
L=contracts_amd/csrc
RATE_PREROLL=12000 timeout 600 tools/ab.sh 3 "C5 C5:fused" $L/libcontracts_engine.so $L/libcontracts_engine_obssc1.so > $OUT/c5.txt 2>&1
grep -v amdgpu.ids $OUT/c5.txt
timeout 900 tools/ab.sh 3 "C2 C2:fused C4 C4:fused C3 cleanup,8,262144" $L/libcontracts_engine.so $L/libcontracts_engine_obssc1.so > $OUT/grid.txt 2>&1
grep -v amdgpu.ids $OUT/grid.txt
