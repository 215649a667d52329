#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel trace + separate PMC passes of the default bench workload.
# Output: gpurun_out/prof/{kt,pmc1,pmc2,fetch,write}/...  (summarise with tools/summarise_profiles.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 400 --warmup 50 > $OUT/kt.json 2> $OUT/kt.log
tail -1 $OUT/kt.json | cut -c1-300
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- $B --steps 60 --warmup 10 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- $B --steps 60 --warmup 10 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 60 --warmup 10 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 60 --warmup 10 > $OUT/write.log 2>&1
# keep only the small CSVs (the 64 MiB merge limit)
find $OUT -name '*_agent_info.csv' -delete
find $OUT -name '*kernel_trace.csv' -size +20M -delete
du -sh $OUT
