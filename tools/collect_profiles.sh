#!/bin/bash
# Runs ON the GPU box (via gpurun): the round's profile set of the headline workload.
#   tools/collect_profiles.sh [TAG]        (TAG default r03)
# 1. rocprofv3 --kernel-trace --stats over the default bench command (every config of the line) -> kernel_stats, and the
#    raw trace split by launch size (tools/kernel_trace_by_config.py) -> one row per (kernel, config)
# 2. PMC passes (separate runs, --pmc only) over tools/pmc_driver.py for the per-step kernel and the fused kernel:
#    instruction mix / waits / LDS conflicts, then FETCH_SIZE and WRITE_SIZE each in its own pass
# Output under gpurun_out/prof_$TAG; summarise on the build box with tools/summarise_profiles.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03}
OUT=$R/gpurun_out/prof_$TAG
# A profile set is evidence about ONE commit: refuse to collect unless the in-tree library is a build of the sources that lie
# here AND those sources were a committed state when it was built (contracts_amd/build.py: provenance; the record travels with
# the library, the snapshot has no .git).  The summariser on the build box then refuses if HEAD has moved since.
PROV=$(cd $R && python3 -m contracts_amd.build --provenance) || { echo "collect_profiles: refused — $PROV"; exit 3; }
rm -rf $OUT; mkdir -p $OUT
echo "$PROV" > $OUT/provenance.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop --no-boundary --no-counter-rng --min-seconds 0.2 --full-out $OUT/kt_bench_full.json > $OUT/kt_bench_line.json 2> $OUT/kt.log
tail -1 $OUT/kt_bench_line.json | cut -c1-200
python3 $R/tools/kernel_trace_by_config.py $OUT/kt $OUT/kernel_stats_by_config.csv
cd $R
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
G3="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVES"
ES=$((64*16384))
tools/pmc_run.sh ${TAG}_step k_grid_step $ES "$G1" "$G2" "$G3" "FETCH_SIZE" "WRITE_SIZE" -- --mode step --steps 64 > $OUT/pmc_step.txt 2>&1
tools/pmc_run.sh ${TAG}_fused k_grid_rollout $ES "$G1" "$G2" "$G3" "FETCH_SIZE" "WRITE_SIZE" -- --mode fused --steps 64 --T 16 > $OUT/pmc_fused.txt 2>&1
ES5=$((64*32768))
tools/pmc_run.sh ${TAG}_sd_step k_sd_step $ES5 "$G1" "FETCH_SIZE" "WRITE_SIZE" -- --kind selfdrive --agents 4 --envs 32768 --mode step --steps 64 > $OUT/pmc_sd_step.txt 2>&1
tools/pmc_run.sh ${TAG}_sd_fused k_sd_rollout $ES5 "$G1" "FETCH_SIZE" "WRITE_SIZE" -- --kind selfdrive --agents 4 --envs 32768 --mode fused --steps 64 --T 16 > $OUT/pmc_sd_fused.txt 2>&1
# HBM traffic of the other BASELINE configs (FETCH_SIZE / WRITE_SIZE only)
ES3=$((64*16384)); ES2=$((64*4096))
tools/pmc_run.sh ${TAG}_c3_step k_grid_step $ES3 "FETCH_SIZE" "WRITE_SIZE" -- --kind harvest --agents 8 --envs 16384 --mode step --steps 64 > $OUT/pmc_c3_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c3_fused k_grid_rollout $ES3 "FETCH_SIZE" "WRITE_SIZE" -- --kind harvest --agents 8 --envs 16384 --mode fused --steps 64 --T 16 > $OUT/pmc_c3_fused.txt 2>&1
tools/pmc_run.sh ${TAG}_c2_step k_grid_step $ES2 "FETCH_SIZE" "WRITE_SIZE" -- --kind cleanup --agents 4 --envs 4096 --mode step --steps 64 > $OUT/pmc_c2_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c2_fused k_grid_rollout $ES2 "FETCH_SIZE" "WRITE_SIZE" -- --kind cleanup --agents 4 --envs 4096 --mode fused --steps 64 --T 16 > $OUT/pmc_c2_fused.txt 2>&1
tools/pmc_run.sh ${TAG}_c1_step k_feat_step $ES3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- --kind harvest_features --agents 2 --envs 16384 --mode step --steps 64 > $OUT/pmc_c1_step.txt 2>&1
tools/pmc_run.sh ${TAG}_c1_fused k_feat_rollout $ES3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- --kind harvest_features --agents 2 --envs 16384 --mode fused --steps 64 --T 16 > $OUT/pmc_c1_fused.txt 2>&1
for k in step fused sd_step sd_fused c3_step c3_fused c2_step c2_fused c1_step c1_fused; do cp $R/gpurun_out/pmc/${TAG}_$k/summary.json $OUT/pmc_$k.json; done
find $OUT -name '*_agent_info.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
du -sh $OUT
