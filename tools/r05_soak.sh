#!/bin/bash
# randomised engine-vs-oracle soaks at HEAD (every field, every step / fused chunk): reference stream, counter stream, the packed
# feature kernels, selfdrive
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/soak; mkdir -p $OUT
cd $R
timeout 1000 python tools/soak.py ${SOAK_S:-300} ${SOAK_SEED:-501} 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/soak.py ${SOAK_S2:-150} $((${SOAK_SEED:-501}+1)) counter 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/soak.py ${SOAK_S2:-150} $((${SOAK_SEED:-501}+2)) quad 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/soak_selfdrive.py ${SOAK_S2:-150} $((${SOAK_SEED:-501}+3)) 2>&1 | grep -v amdgpu.ids | tail -2
