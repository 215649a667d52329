"""CPU: the C-ABI library loads, exports every function include/contracts_engine.h declares, its
structs have the layout the ctypes mirrors assume, and it fails LOUDLY without a GPU (no CPU path)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "contracts_engine.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ce_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_exported():
    from contracts_amd import _lib
    L = _lib.load()
    names = declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(L, name), "library does not export %s" % name
        assert name in _lib.EXPORTS, "ctypes table misses %s" % name
    assert L.ce_abi_version() == _lib.CE_ABI_VERSION == 4


def test_struct_layout_matches_ctypes():
    from contracts_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "contracts_engine.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ce_config), offsetof(ce_config, env_index_base),
         offsetof(ce_config, contract_low), offsetof(ce_config, start_vel_ambulance), sizeof(ce_buffers),
         offsetof(ce_buffers, grid), offsetof(ce_buffers, error_flags));
  printf("%zu ", offsetof(ce_buffers, actions_taken));
  printf("%zu %zu %zu %zu %zu %d\n", offsetof(ce_buffers, sd_info), sizeof(ce_traj), offsetof(ce_traj, obs),
         offsetof(ce_traj, features), offsetof(ce_traj, sd_info), CE_ABI_VERSION);
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(ce_traj, num_envs), offsetof(ce_traj, num_agents), sizeof(ce_state_header),
         offsetof(ce_state_header, horizon), offsetof(ce_state_header, layout_hash), offsetof(ce_state_header, params),
         sizeof(ce_state_field), offsetof(ce_state_field, env_bytes));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    cfg, buf, traj = _lib.CeConfig, _lib.CeBuffers, _lib.CeTraj
    want = [C.sizeof(cfg), cfg.env_index_base.offset, cfg.contract_low.offset, cfg.start_vel_ambulance.offset,
            C.sizeof(buf), buf.grid.offset, buf.error_flags.offset,
            buf.actions_taken.offset, buf.sd_info.offset, C.sizeof(traj), traj.obs.offset, traj.features.offset, traj.sd_info.offset, _lib.CE_ABI_VERSION]
    sh, sf = _lib.CeStateHeader, _lib.CeStateField  # ABI 4: the ring's batch echo and the one-call snapshot's header / directory
    want += [traj.num_envs.offset, traj.num_agents.offset, C.sizeof(sh), sh.horizon.offset, sh.layout_hash.offset, sh.params.offset,
             C.sizeof(sf), sf.env_bytes.offset]
    assert got == want
    # the oracle's own ctypes mirrors (oracle/pyoracle.py) follow the same header: orc_get_buffers writes a whole ce_buffers
    from oracle import pyoracle as po
    assert C.sizeof(po.CeBuffers) == C.sizeof(buf) and C.sizeof(po.CeConfig) == C.sizeof(cfg)
    assert [f[0] for f in po.CeBuffers._fields_] == [f[0] for f in buf._fields_]
    assert po.make_config("cleanup", 1, 1).abi_version == _lib.CE_ABI_VERSION


def test_no_gpu_fails_loudly():
    """without a gfx950 device engine creation must raise — never fall back to a CPU path"""
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    if _lib.load().ce_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.EngineError) as ei:
        BatchedEnv("cleanup", 4, 2)
    assert "CE_ENODEV" in str(ei.value)


def test_bad_config_rejected():
    from contracts_amd import _lib
    from contracts_amd.engine import make_config
    L = _lib.load()
    h = C.c_void_p()
    for bad in (dict(kind="cleanup", num_envs=0, num_agents=2), dict(kind="cleanup", num_envs=4, num_agents=10),
                dict(kind="harvest", num_envs=4, num_agents=2, contract="cleanup"),
                dict(kind="selfdrive", num_envs=4, num_agents=11)):
        cfg = make_config(**bad)
        assert L.ce_create(C.byref(cfg), C.byref(h)) == -22, bad
    assert L.ce_create(None, C.byref(h)) == -22


def test_product_never_touches_oracle():
    """the product package must not import, link or mention oracle/ (the oracle is the checker)"""
    pkg = os.path.join(ROOT, "contracts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "orc_" not in txt, f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "csrc", "libcontracts_engine.so")]).decode()
    assert "oracle" not in out


def test_synth_action_host_is_deterministic_and_uniform():
    from contracts_amd import _lib
    L = _lib.load()
    import numpy as np
    a = [L.ce_synth_action_host(7, e, t, k, 8) for e in range(40) for t in range(20) for k in range(8)]
    b = [L.ce_synth_action_host(7, e, t, k, 8) for e in range(40) for t in range(20) for k in range(8)]
    assert a == b and min(a) == 0 and max(a) == 7
    hist = np.bincount(a, minlength=8) / len(a)
    assert np.all(np.abs(hist - 0.125) < 0.03)


def test_library_is_the_build_of_these_sources():
    """the in-tree library is checked against the build record by content, not by file times: every source and header hashed,
    the flags, and the library file itself (contracts_amd/build.py); a stale or foreign .so would make needs_build() true"""
    from contracts_amd import build as b
    rec = b.last_build()
    if rec is None:
        pytest.skip("no build record beside the library (built by other means): nothing to compare")
    assert rec["fingerprint"] == b.fingerprint(), "sources changed since the library was built"
    assert rec["lib_sha16"] == b._sha16(b.LIB)
    assert set(rec["translation_units"]) == {"ce_api.hip", "ce_grid_kernels.hip", "ce_grid_kernels_ctr.hip", "ce_selfdrive_kernels.hip"}
    assert not b.needs_build()


def test_custom_layout_rules_are_checked_before_the_device():
    """ce_config.ascii_map: a layout that breaks a rule of the header is refused with CE_EINVAL and a message naming the rule —
    on any box, before the device is looked for; a valid one gets as far as the device check (CE_ENODEV here, no GPU)"""
    from contracts_amd import _lib
    from contracts_amd.engine import make_config
    L = _lib.load()

    def create(kind, n, rows):
        cfg = make_config(kind, 4, n, ascii_map=rows)
        h = C.c_void_p()
        rc = L.ce_create(C.byref(cfg), C.byref(h))
        msg = (L.ce_last_error(h) or b"").decode() if h else ""
        if h:
            L.ce_destroy(h)
        return rc, msg

    ok = ["@@@@@@", "@HB P@", "@RB P@", "@@@@@@"]
    for kind, n, rows, word in (("cleanup", 3, ok, "spawn points"), ("cleanup", 1, ["@@@@@@", "@HB P ", "@RB P@", "@@@@@@"], "walled"),
                                ("cleanup", 1, ["@@@@@@", "@ B P@", "@@@@@@"], "at least one"), ("cleanup", 1, ["@@@@@@", "@HA P@", "@@@@@@"], "alphabet"),
                                ("harvest", 1, ["@" * 40] * 3, "frame"), ("harvest_features", 2, ["@@@@", "@AP@", "@@@@"], "grid kinds"),
                                ("cleanup", 1, ["@" * 18] + ["@" + "B" * 16 + "@"] * 8 + ["@HP" + "@" * 15] + ["@" * 18], "apple cells")):
        rc, msg = create(kind, n, rows)
        assert rc == -22 and word in msg, (kind, rows[:2], rc, msg)
    import torch
    if not torch.cuda.is_available():
        rc, msg = create("cleanup", 2, ok)
        assert rc == -19, (rc, msg)
    with pytest.raises(ValueError):
        make_config("cleanup", 1, 1, ascii_map=["@@@", "@@"])
    # a C caller with garbage dimensions: refused by the frame rule BEFORE anything is read through the pointer (it
    # used to copy rows * cols bytes first — an out-of-bounds read, or a length_error / bad_alloc through the C boundary)
    for rows_, cols_ in ((2 ** 31 - 1, 2 ** 31 - 1), (0, 5), (3, 0), (26, 18), (25, 19), (2 ** 20, 1)):
        cfg = make_config("cleanup", 4, 1, ascii_map=ok)
        cfg.map_rows, cfg.map_cols = rows_, cols_
        h = C.c_void_p()
        rc = L.ce_create(C.byref(cfg), C.byref(h))
        msg = (L.ce_last_error(h) or b"").decode() if h else ""
        if h:
            L.ce_destroy(h)
        assert rc == -22 and "frame" in msg, (rows_, cols_, rc, msg)
