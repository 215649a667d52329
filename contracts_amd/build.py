"""Builds the engine's shared library in-tree with hipcc for gfx950.

    python -m contracts_amd.build [--force]

Output: contracts_amd/csrc/libcontracts_engine.so (git-ignored; travels to the GPU box with
the gpurun snapshot).  hipcc cross-compiles gfx950 code objects without a GPU.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

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


# CPython helper of the vector hook's dict protocol (host-side marshalling loops, plain C, system compiler)
PYDICT_SRC = "ce_pydict.c"


def pydict_path():
    import sysconfig
    return os.path.join(HERE, "_ce_pydict" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pydict(verbose=False):
    import sysconfig
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for contracts_amd/csrc/ce_pydict.c")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I", sysconfig.get_paths()["include"], os.path.join(CSRC, PYDICT_SRC), "-o", pydict_path()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return pydict_path()


INFO = os.path.join(CSRC, "build_info.json")  # record of the last default build (git-ignored, like the library)


def _sha16(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def fingerprint():
    """what the library is a function of: every source and header by content, and the flags"""
    files = {s: _sha16(os.path.join(CSRC, s)) for s in SOURCES + HEADERS + [PYDICT_SRC]}
    return {"files": files, "flags": FLAGS}


def kernels_sha16(fp=None):
    """one hash for "the kernel sources this library was built from": every source / header by content + the flags"""
    fp = fp or fingerprint()
    blob = json.dumps(fp, sort_keys=True).encode()
    return hashlib.sha256(blob).hexdigest()[:16]


def git_state():
    """(HEAD, sources differ from HEAD) of the repository the sources sit in; (None, None) outside a work tree (the GPU box
    receives a snapshot without .git: the record written at build time on the build box travels with the library)"""
    root = os.path.dirname(HERE)
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "contracts_amd/csrc", "include"],
                               capture_output=True, text=True, check=True).stdout.strip() != ""
        return head, dirty
    except (OSError, subprocess.CalledProcessError):
        return None, None


def last_build():
    try:
        with open(INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def needs_build():
    """content-based: the library is current iff it is the file the record describes and the record's fingerprint is today's"""
    rec = last_build()
    if not os.path.exists(LIB) or rec is None or not os.path.exists(pydict_path()):
        return True
    return rec.get("fingerprint") != fingerprint() or rec.get("lib_sha16") != _sha16(LIB)


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
    if os.environ.get("CE_FORCE_BUILD") == "1":
        force = True
    if not force and not needs_build():
        return LIB
    t0 = time.time()
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
    if lib == LIB:
        build_pydict(verbose)
        ver = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        with open(INFO, "w") as f:
            head, dirty = git_state()
            json.dump({"fingerprint": fingerprint(), "kernels_sha16": kernels_sha16(), "git_head": head, "git_dirty_sources": dirty,
                       "lib_sha16": _sha16(LIB), "compile_seconds": round(time.time() - t0, 1),
                       "hipcc": ver[0] if ver else "", "translation_units": SOURCES, "offload_arch": "gfx950",
                       "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}, f, indent=1)
    return lib


def provenance():
    """What a measurement taken with the in-tree library may be attributed to, or the reason it may not be attributed at all:
    the library must be the one the record describes, built from the sources that lie here, and those sources must have been
    a committed state (tools/collect_profiles.sh refuses to write profile sets otherwise; bench.py quotes it beside `traffic`)"""
    rec = last_build()
    if rec is None or needs_build():
        return None, "the in-tree library is not a build of the sources that lie here (python -m contracts_amd.build)"
    if rec.get("git_head") is None:
        return None, "the build record carries no git HEAD (built outside a work tree)"
    if rec.get("git_dirty_sources"):
        return None, "the library was built from kernel sources that differ from HEAD %s (commit, rebuild, collect)" % rec["git_head"][:12]
    return {"git_head": rec["git_head"], "kernels_sha16": rec.get("kernels_sha16"), "lib_sha16": rec["lib_sha16"], "built_at": rec["built_at"]}, None


if __name__ == "__main__":
    if "--provenance" in sys.argv:  # exit code 0 + JSON when measurements may be attributed to a commit, else 3 + the reason
        prov, why = provenance()
        print(json.dumps(prov) if prov else why)
        sys.exit(0 if prov else 3)
    print(build(force="--force" in sys.argv, verbose=True))
