#!/bin/bash
# builds the truncated profiling variants libce_trunc<k>.so (k = phase boundary) next to the engine library
cd "$(dirname "$0")/../contracts_amd/csrc"
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -mllvm -amdgpu-kernarg-preload-count=16"
build() { k=$1; /opt/rocm/bin/hipcc $F -DCE_DIAGNOSTIC -DCE_TRUNCATE=$k -c ce_grid_kernels.hip -o /tmp/trunc_$k.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libce_trunc$k.so ce_api.o /tmp/trunc_$k.o ce_grid_kernels_ctr.o ce_selfdrive_kernels.o -Wl,-z,defs -lamdhip64 -L/opt/rocm/lib; }
for k in 1 2 3 11; do build $k & done; wait
for k in 12 10 13 4; do build $k & done; wait
for k in 5 6 7 8; do build $k & done; wait
ls libce_trunc*.so | wc -l
