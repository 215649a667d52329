"""The CPU restatement under the engine's own entry-point names (oracle.c's `ce_*` block, SURVEY 8b "identical ABI, device = cpu"):
a C harness written against include/contracts_engine.h links against liboracle.so unchanged (CPU, here) and against
libcontracts_engine.so (GPU test) and prints the same digests.  The oracle is still test infrastructure — this test is the only
thing that links it under those names."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
ENGINE = os.path.join(ROOT, "contracts_amd", "csrc", "libcontracts_engine.so")
SPECS = [(0, 24, 3, 1), (1, 16, 5, 2), (0, 8, 8, 0)]  # (kind, E, n, contract)


def build_harness(tmp_path, lib, tag):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    exe = str(tmp_path / ("harness_" + tag))
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_harness.c"), "-o", exe,
                           lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined"])
    return exe


def run(exe, spec, steps=60):
    out = subprocess.check_output([exe] + [str(x) for x in spec] + [str(steps)], text=True)
    return dict(line.split() for line in out.strip().splitlines())


def test_oracle_exports_the_core_abi_under_the_engines_names():
    from contracts_amd import _lib
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    L = C.CDLL(ORACLE)
    core = ["ce_abi_version", "ce_device_count", "ce_create", "ce_destroy", "ce_seed", "ce_reset", "ce_step", "ce_step_range",
            "ce_step_host", "ce_rollout", "ce_get_buffers", "ce_synchronize", "ce_set_contract", "ce_set_flags", "ce_download",
            "ce_upload", "ce_last_error"]
    for name in core:
        fn = getattr(L, name)
        fn.restype, fn.argtypes = _lib.EXPORTS[name]  # the engine's prototypes, verbatim
    assert L.ce_abi_version() == _lib.CE_ABI_VERSION
    # the same calls as oracle/pyoracle.py makes through orc_*: identical results, incl. an upload that must reach the working state
    from contracts_amd.engine import make_config  # the ENGINE's ctypes mirror of ce_config
    from oracle.pyoracle import Oracle
    E, n = 12, 4
    cfg = make_config("cleanup", E, n, contract="cleanup", horizon=15, auto_reset=True)
    h = C.c_void_p()
    assert L.ce_create(C.byref(cfg), C.byref(h)) == 0
    ref = Oracle("cleanup", E, n, contract="cleanup", horizon=15, auto_reset=True)
    assert L.ce_seed(h, None, 77, None, 3) == 0 and L.ce_reset(h, None, None) == 0
    assert L.ce_seed(h, None, 77, None, 0) == -22 and L.ce_step(h, None, None, None) == -22
    ref.seed(seed0=77)
    ref.reset()
    rs = np.random.RandomState(1)
    planes = rs.randint(8, size=(40, E, n)).astype(np.uint8)
    for t in range(20):
        assert L.ce_step(h, planes[t].ctypes.data, None, None) == 0
        ref.step(planes[t])
    assert L.ce_rollout(h, planes[20:].ctypes.data, 10, 1, None) == 0  # ten more, the launch loop's name
    assert L.ce_step_range(h, planes[30].ctypes.data, None, 0, 5, None) == 0 and L.ce_step_range(h, planes[30].ctypes.data, None, 5, E - 5, None) == 0
    assert L.ce_step_range(h, planes[30].ctypes.data, None, 5, E, None) == -22
    for t in range(20, 31):
        ref.step(planes[t])
    for f in ("agents", "rng", "reward", "features", "obs", "waste_perm", "int_metrics", "grid"):
        want = np.ascontiguousarray(getattr(ref, f))
        got = np.empty_like(want)
        assert L.ce_download(h, f.encode(), 0, E, got.ctypes.data, got.nbytes) == 0, f
        assert got.tobytes() == want.tobytes(), f
    assert L.ce_download(h, b"nope", 0, E, got.ctypes.data, got.nbytes) == -22 and b"unknown" in L.ce_last_error(h)
    # upload: move every agent of env 3 state-side and step both — the upload reaches the env's working state
    ag = np.ascontiguousarray(ref.agents).copy()
    th = np.full(E, 0.125)
    assert L.ce_upload(h, b"theta", 0, E, th.ctypes.data, th.nbytes) == 0
    ref.theta[:] = th
    ref.import_state()
    assert L.ce_upload(h, b"reward", 0, E, th.ctypes.data, th.nbytes) == -22  # an output is not state
    assert L.ce_step(h, planes[31].ctypes.data, None, None) == 0
    ref.step(planes[31])
    got = np.empty((E, n), np.float64)
    assert L.ce_download(h, b"reward", 0, E, got.ctypes.data, got.nbytes) == 0 and np.array_equal(got, ref.reward) and ag is not None
    assert L.ce_set_flags(h, 0x1, 0x1) == -22 and L.ce_set_flags(h, 0x2, 0) == 0 and L.ce_set_contract(h, 2, 0.0, 1.0, 0.0) == -22
    assert L.ce_destroy(h) == 0
    ref.close()


@pytest.mark.parametrize("spec", SPECS)
def test_c_harness_runs_on_the_cpu_library(tmp_path, spec):
    got = run(build_harness(tmp_path, ORACLE, "cpu"), spec)
    assert set(got) >= {"agents", "rng", "reward", "features", "obs", "f64_metrics"}
    # and it is the oracle that answered: the same rollout through pyoracle gives the same reward digest
    from oracle.pyoracle import Oracle
    kind, E, n, contract = spec
    kname, cname = {0: "cleanup", 1: "harvest"}[kind], {0: None, 1: "cleanup", 2: "harvest_local"}[contract]
    kw = dict(contract=cname, horizon=23, auto_reset=True)
    if cname == "cleanup":
        kw["contract_high"] = float(np.float32(0.2))
    elif cname == "harvest_local":
        kw["contract_high"] = 10.0
    o = Oracle(kname, E, n, **kw)
    o.seed(seed0=4242)
    o.reset()
    lcg, mask = 88172645463325252, (1 << 64) - 1
    for t in range(60):
        a = np.empty(E * n, np.uint8)
        for i in range(E * n):
            lcg = (lcg * 6364136223846793005 + 1442695040888963407) & mask
            a[i] = (lcg >> 33) % (8 if kind == 0 else 7)
        o.step(a.reshape(E, n))

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & mask
        return "%016x" % h

    assert got["reward"] == fnv(np.ascontiguousarray(o.reward).tobytes())
    assert got["agents"] == fnv(np.ascontiguousarray(o.agents).tobytes())
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("spec", SPECS)
def test_c_harness_swaps_libraries(tmp_path, spec):
    """the SAME C source, linked against the HIP engine and against the CPU restatement: identical digests for every field"""
    cpu = run(build_harness(tmp_path, ORACLE, "cpu"), spec)
    gpu = run(build_harness(tmp_path, ENGINE, "gpu"), spec)
    assert cpu == gpu
