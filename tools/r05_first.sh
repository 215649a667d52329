#!/bin/bash
# round 5, first GPU call: (a) floor of a resident per-step path (tools/microbench/doorbell_floor.hip), (b) A/B of the fused
# selfdrive kernel, HEAD vs -DCE_SD_ROLLOUT_PLAIN, at 32 768 and 262 144 envs in the steady state
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_first; mkdir -p $OUT
cd $R
timeout 300 tools/microbench/doorbell_floor 3000 > $OUT/doorbell_floor.txt 2>&1; echo "doorbell_floor exit $?" >> $OUT/doorbell_floor.txt
tail -40 $OUT/doorbell_floor.txt
L=contracts_amd/csrc
RATE_PREROLL=12000 timeout 900 tools/ab.sh 3 "C5:fused selfdrive,4,262144:fused C5" $L/libcontracts_engine.so $L/libcontracts_engine_sdplain.so > $OUT/sd_ab.txt 2>&1
cat $OUT/sd_ab.txt
