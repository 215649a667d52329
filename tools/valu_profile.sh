#!/bin/bash
# Runs ON the GPU box: per-phase VALU/SALU/LDS instruction counts of the cleanup step kernel from truncated builds
R=${GRAFT_REPO_ROOT:-$(pwd)}; C=$R/contracts_amd/csrc
python3 $R/tools/valu_profile_run.py make $R/gpurun_out/vp_state.npz || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/vp; mkdir -p $R/gpurun_out/vp
for k in full 1 2 3 11 12 10 13 4 5 6 7 8; do
  L=$C/libce_trunc$k.so; [ $k = full ] && L=$C/libcontracts_engine.so
  CONTRACTS_AMD_LIB=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/vp/$k -- python3 $R/tools/valu_profile_run.py run $R/gpurun_out/vp_state.npz > $R/gpurun_out/vp/$k.log 2>&1
done
rm -f $R/gpurun_out/vp_state.npz
cd $R && python3 tools/valu_profile_report.py gpurun_out/vp
