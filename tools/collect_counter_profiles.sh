#!/bin/bash
# Runs ON the GPU box: PMC passes of the counter-RNG kernels (CE_FLAG_RNG_COUNTER) for the headline and harvest shapes, per-step
# and fused, + a kernel trace of a per-step run -> gpurun_out/prof_${TAG}_counter/, summarised into profiles/${TAG}_counter_rng.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}; OUT=$R/gpurun_out/prof_${TAG}_counter; cd $R
PROV=$(python3 -m contracts_amd.build --provenance) || { echo "collect_counter_profiles: refused — $PROV"; exit 3; }  # see collect_profiles.sh
rm -rf $OUT; mkdir -p $OUT
echo "$PROV" > $OUT/provenance.json
ES=$((64*16384))
MIX="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES"
tools/pmc_run.sh ${TAG}_ctr_c4_step k_grid_step $ES "FETCH_SIZE" "WRITE_SIZE" "$MIX" -- --kind cleanup --agents 8 --envs 16384 --mode step --steps 64 --rng counter > $OUT/pmc_c4_step.txt 2>&1
tools/pmc_run.sh ${TAG}_ctr_c4_fused k_grid_rollout $ES "FETCH_SIZE" "WRITE_SIZE" "$MIX" -- --kind cleanup --agents 8 --envs 16384 --mode fused --steps 64 --T 16 --rng counter > $OUT/pmc_c4_fused.txt 2>&1
tools/pmc_run.sh ${TAG}_ctr_c3_step k_grid_step $ES "FETCH_SIZE" "WRITE_SIZE" "$MIX" -- --kind harvest --agents 8 --envs 16384 --mode step --steps 64 --rng counter > $OUT/pmc_c3_step.txt 2>&1
for k in c4_step c4_fused c3_step; do cp $R/gpurun_out/pmc/${TAG}_ctr_$k/summary.json $OUT/pmc_$k.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/pmc_driver.py --kind cleanup --agents 8 --envs 16384 --mode step --steps 400 --rng counter > $OUT/kt.log 2>&1
cd $R
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
find $OUT/kt -name '*.csv' -size +4M -delete
