#!/bin/bash
# Cost model of a counter-RNG (Philox4x32-10) build of the step kernel, measured instead of guessed:
#   philox4   the shipped kernel + 4 wave-wide Philox4x32-10 evaluations per env-step (1024 stream words: what the step's
#             rand(222) window, the 119-element shuffle and the two agent shuffles would draw on demand)
#   notwist   the shipped kernel without the MT19937 regeneration and without the key write-back (what the counter mode saves
#             on chip and in HBM; streams are wrong: cost model only)
#   both      the two together ~ the counter-RNG kernel
#   tools/philox_probe.sh && gpurun -- 'tools/ab.sh 3 "C4 C4:fused 'cleanup,8,262144'" contracts_amd/csrc/libcontracts_engine.so contracts_amd/csrc/libce_probe_{philox4,notwist,both}.so'
cd "$(dirname "$0")/../contracts_amd/csrc"
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-rdc -mllvm -amdgpu-kernarg-preload-count=16"
build() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" -c ce_grid_kernels.hip -o /tmp/probe_$name.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libce_probe_$name.so ce_api.o /tmp/probe_$name.o ce_grid_kernels_ctr.o ce_selfdrive_kernels.o -Wl,-z,defs -lamdhip64 -L/opt/rocm/lib; }
build philox4 -DCE_DIAGNOSTIC -DCE_PROBE_PHILOX=4 & build notwist -DCE_DIAGNOSTIC -DCE_ABLATE_TWIST & build both -DCE_DIAGNOSTIC -DCE_PROBE_PHILOX=4 -DCE_ABLATE_TWIST & wait
ls libce_probe_*.so
