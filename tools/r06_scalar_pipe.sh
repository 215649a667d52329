#!/bin/bash
# Runs ON the GPU box: what the scalar pipe costs at REAL concurrency.
#   1. tools/microbench/issue_rates: SALU / VALU issue rates per CU by waves per CU (what the pipe can do)
#   2. PMC passes whose measured dispatch is ONE launch of all 16384 envs per step (--streams 1: no slice serialisation under
#      --pmc), per-step and fused kernels, MT19937 stream
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06_scalar; mkdir -p $OUT
cd $R
timeout 300 tools/microbench/issue_rates > $OUT/issue_rates.txt 2>&1; cat $OUT/issue_rates.txt
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters_available.txt; wc -l $OUT/sq_counters_available.txt
grep -c . $OUT/sq_counters_available.txt
for c in SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVES; do grep -qx $c $OUT/sq_counters_available.txt && echo "have $c" || echo "MISSING $c"; done > $OUT/have.txt; cat $OUT/have.txt
ES=$((64*16384))
G1="SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"
G2="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES"
G3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
G4="GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
G5="SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"
tools/pmc_run.sh r06s_step k_grid_step $ES "$G1" "$G2" "$G3" "$G4" "$G5" -- --mode step --steps 64 --streams 1 > $OUT/pmc_step_one_launch.txt 2>&1
tools/pmc_run.sh r06s_fused k_grid_rollout $ES "$G1" "$G2" "$G3" "$G4" "$G5" -- --mode fused --steps 64 --T 16 --streams 1 > $OUT/pmc_fused_one_launch.txt 2>&1
cat $OUT/pmc_step_one_launch.txt $OUT/pmc_fused_one_launch.txt
cp $R/gpurun_out/pmc/r06s_step/summary.json $OUT/pmc_step_one_launch.json; cp $R/gpurun_out/pmc/r06s_fused/summary.json $OUT/pmc_fused_one_launch.json
tail -5 $R/gpurun_out/pmc/r06s_step/p0.log
