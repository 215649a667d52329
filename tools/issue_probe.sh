#!/bin/bash
# builds probe variants of the engine (per env-step in k_grid_step: 128 extra VALU no-ops / 128 SALU no-ops / 16 serialised
# scalar-load round trips / 16 serialised LDS round trips) for an A/B run:
#   tools/issue_probe.sh && gpurun -- 'bash tools/ab_many.sh 3 contracts_amd/csrc/libcontracts_engine.so contracts_amd/csrc/libce_probe_*.so'
cd "$(dirname "$0")/../contracts_amd/csrc"
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -mllvm -amdgpu-kernarg-preload-count=16"
build() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" -c ce_grid_kernels.hip -o /tmp/probe_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libce_probe_$name.so ce_api.o /tmp/probe_$name.o ce_grid_kernels_ctr.o ce_selfdrive_kernels.o -Wl,-z,defs -lamdhip64 -L/opt/rocm/lib; }
build valu -DCE_DIAGNOSTIC -DCE_PROBE_VALU & build salu -DCE_DIAGNOSTIC -DCE_PROBE_SALU & build smem -DCE_DIAGNOSTIC -DCE_PROBE_SMEM & build lds -DCE_DIAGNOSTIC -DCE_PROBE_LDS & wait
ls libce_probe_*.so
