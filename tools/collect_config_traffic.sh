#!/bin/bash
# Runs ON the GPU box: only the C1 / C2 / C3 traffic passes of tools/collect_profiles.sh (into the same gpurun_out/prof_$TAG)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r02}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd $R
ES3=$((64*16384)); ES2=$((64*4096))
tools/pmc_run.sh ${TAG}_c3_step k_grid_step $ES3 "FETCH_SIZE" "WRITE_SIZE" -- --kind harvest --agents 8 --envs 16384 --mode step --steps 64 > $OUT/pmc_c3_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c3_fused k_grid_rollout $ES3 "FETCH_SIZE" "WRITE_SIZE" -- --kind harvest --agents 8 --envs 16384 --mode fused --steps 64 --T 16 > $OUT/pmc_c3_fused.txt 2>&1
tools/pmc_run.sh ${TAG}_c2_step k_grid_step $ES2 "FETCH_SIZE" "WRITE_SIZE" -- --kind cleanup --agents 4 --envs 4096 --mode step --steps 64 > $OUT/pmc_c2_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c2_fused k_grid_rollout $ES2 "FETCH_SIZE" "WRITE_SIZE" -- --kind cleanup --agents 4 --envs 4096 --mode fused --steps 64 --T 16 > $OUT/pmc_c2_fused.txt 2>&1
tools/pmc_run.sh ${TAG}_c1_step k_feat_step $ES3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- --kind harvest_features --agents 2 --envs 16384 --mode step --steps 64 > $OUT/pmc_c1_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c1_fused k_feat_rollout $ES3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- --kind harvest_features --agents 2 --envs 16384 --mode fused --steps 64 --T 16 > $OUT/pmc_c1_fused.txt 2>&1
for k in c3_step c3_fused c2_step c2_fused c1_step c1_fused; do cp $R/gpurun_out/pmc/${TAG}_$k/summary.json $OUT/pmc_$k.json; done
for f in $OUT/pmc_c*_*.txt; do echo $f; tail -n 3 $f; done
