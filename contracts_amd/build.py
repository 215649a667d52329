"""Builds the engine's shared library in-tree with hipcc for gfx950.

    python -m contracts_amd.build [--force]

Output: contracts_amd/csrc/libcontracts_engine.so (git-ignored; travels to the GPU box with
the gpurun snapshot).  hipcc cross-compiles gfx950 code objects without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libcontracts_engine.so")
SOURCES = ["ce_api.hip", "ce_grid_kernels.hip", "ce_grid_kernels_ctr.hip", "ce_selfdrive_kernels.hip"]
HEADERS = ["ce_device.h", "ce_grid_probe.inc", os.path.join("..", "..", "include", "contracts_engine.h")]
# -ffp-contract=off: float64 reward/transfer arithmetic must round exactly like the reference's
# separate multiply and add; no fast-math anywhere.
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's arguments arrive in SGPRs at wave launch (gfx950), see
# k_grid_step in ce_grid_kernels.hip.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine has no non-HIP build")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    stamps = os.environ.get("CE_PHASE_STAMPS") == "1"
    lib, flags, suffix = LIB, list(FLAGS), ""
    if stamps:  # diagnostic build: s_memtime stamps per phase, separate file, never benchmarked
        flags += ["-DCE_DIAGNOSTIC", "-DCE_PHASE_STAMPS"]
        lib, suffix, force = LIB.replace(".so", "_stamps.so"), "_stamps", True
    variant = os.environ.get("CE_VARIANT")  # experiment builds: extra flags, separate file (CONTRACTS_AMD_LIB selects it)
    if variant:
        flags += os.environ.get("CE_VARIANT_FLAGS", "").split()
        lib, suffix, force = LIB.replace(".so", "_%s.so" % variant), "_" + variant, True
    if not force and not needs_build():
        return LIB
    objs, procs = [], []
    for src in SOURCES:  # the translation units compile side by side
        obj = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [cmd for cmd, proc in procs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
