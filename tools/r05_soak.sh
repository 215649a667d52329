#!/bin/bash
# randomised engine-vs-oracle soaks at HEAD (every field, every step / fused chunk): reference stream, counter stream, the packed
# feature kernels, selfdrive
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05_soak; mkdir -p $OUT
cd $R
timeout 400 python tools/soak.py 300 501 2>&1 | grep -v amdgpu.ids | tail -2
timeout 250 python tools/soak.py 150 502 counter 2>&1 | grep -v amdgpu.ids | tail -2
timeout 250 python tools/soak.py 150 503 quad 2>&1 | grep -v amdgpu.ids | tail -2
timeout 250 python tools/soak_selfdrive.py 150 504 2>&1 | grep -v amdgpu.ids | tail -2
