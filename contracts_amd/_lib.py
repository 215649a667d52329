"""ctypes binding of libcontracts_engine.so (the C-ABI in include/contracts_engine.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable the
import of the library / creation of an engine fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONTRACTS_AMD_LIB") or os.path.join(HERE, "csrc", "libcontracts_engine.so")

CE_ABI_VERSION = 4
KIND = {"cleanup": 0, "harvest": 1, "selfdrive": 2, "harvest_features": 3, "cleanup_features": 4}
FEAT_KINDS = ("harvest_features", "cleanup_features")
FEAT_APPLE_SLOTS, FEAT_WASTE_SLOTS, FEAT_STATE_BYTES = 160, 120, 568  # CE_FEAT_* of the header
CONTRACT = {None: 0, "none": 0, "cleanup": 1, "harvest_local": 2, "selfdrive_distprop": 3}
FLAG_FIRING, FLAG_AUTO_RESET, FLAG_COLLECTIVE, FLAG_INEQUITY, FLAG_COLLISION, FLAG_EXTERNAL_THETA = 1, 2, 4, 8, 16, 32
FLAG_BEAM_TRACE = 64
FLAG_RNG_COUNTER = 128
BEAM_NONE, BEAM_FIRE, BEAM_CLEAN = 0, 1, 2
POLICY_BYTES_MOD, POLICY_ARGMAX_F32, POLICY_AHEAD_NOISE = 1, 2, 3
FAULT_BAD_ACTION, FAULT_NO_SPAWN, FAULT_STEP_AFTER_DONE = 1, 2, 4
ERRORS = {-22: "CE_EINVAL", -12: "CE_ENOMEM", -19: "CE_ENODEV", -5: "CE_EIO", -34: "CE_ERANGE"}


class CeConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("kind", C.c_uint32), ("num_envs", C.c_uint32), ("num_agents", C.c_uint32),
        ("horizon", C.c_uint32), ("contract", C.c_uint32), ("flags", C.c_uint32), ("device", C.c_int32),
        ("env_index_base", C.c_uint64),
        ("contract_low", C.c_double), ("contract_high", C.c_double), ("null_prob", C.c_double),
        ("alpha", C.c_double), ("beta", C.c_double),
        ("low_bound", C.c_double), ("high_bound", C.c_double), ("start_vel", C.c_double),
        ("start_vel_ambulance", C.c_double),
        ("ascii_map", C.c_char_p), ("map_rows", C.c_uint32), ("map_cols", C.c_uint32),
    ]


_P = C.c_void_p


class CeBuffers(C.Structure):
    _fields_ = [
        ("num_envs", C.c_uint32), ("num_agents", C.c_uint32), ("grid_h", C.c_uint32), ("grid_w", C.c_uint32),
        ("obs_agent_stride", C.c_uint32), ("num_features", C.c_uint32), ("num_int_metrics", C.c_uint32),
        ("num_f64_metrics", C.c_uint32), ("obs_env_stride", C.c_uint32), ("rng_words", C.c_uint32),
        ("grid_env_stride", C.c_uint32), ("grid_row_stride", C.c_uint32), ("grid_origin", C.c_uint32),
        ("obs_row_stride", C.c_uint32),
        ("grid", _P), ("agents", _P), ("spawn_perm", _P), ("waste_perm", _P), ("rng", _P), ("timestep", _P),
        ("theta", _P), ("sd_state", _P),
        ("obs", _P), ("obs_f64", _P), ("base_reward", _P), ("reward", _P), ("done", _P), ("done_agents", _P),
        ("info", _P), ("features", _P),
        ("int_metrics", _P), ("f64_metrics", _P), ("final_int_metrics", _P), ("final_f64_metrics", _P),
        ("error_flags", _P), ("beam_map", _P), ("sd_info", _P), ("actions_taken", _P),
    ]


class CeTraj(C.Structure):
    _fields_ = [("num_planes", C.c_uint32), ("first_plane", C.c_uint32), ("num_envs", C.c_uint32), ("num_agents", C.c_uint32),
                ("obs", _P), ("obs_f64", _P), ("base_reward", _P),
                ("reward", _P), ("done", _P), ("done_agents", _P), ("info", _P), ("features", _P), ("sd_info", _P)]


class CeStateHeader(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("abi_version", C.c_uint32), ("header_bytes", C.c_uint32), ("kind", C.c_uint32),
                ("num_envs", C.c_uint32), ("num_agents", C.c_uint32), ("contract", C.c_uint32), ("flags", C.c_uint32),
                ("horizon", C.c_uint32), ("what", C.c_uint32), ("num_fields", C.c_uint32), ("reserved", C.c_uint32),
                ("env_index_base", C.c_uint64), ("layout_hash", C.c_uint64), ("total_bytes", C.c_uint64),
                ("params", C.c_double * 9)]


class CeStateField(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("offset", C.c_uint64), ("env_bytes", C.c_uint64)]


STATE_MAGIC, STATE_OUTPUTS = 0x54534543, 1


class CeFieldReq(C.Structure):
    _fields_ = [("field", C.c_char_p), ("dst", C.c_void_p), ("dst_bytes", C.c_uint64)]


EXPORTS = {
    # name: (restype, argtypes)
    "ce_abi_version": (C.c_int, []),
    "ce_device_count": (C.c_int, []),
    "ce_create": (C.c_int, [C.POINTER(CeConfig), C.POINTER(C.c_void_p)]),
    "ce_destroy": (C.c_int, [C.c_void_p]),
    "ce_set_contract": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double]),
    "ce_set_flags": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "ce_seed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]),
    "ce_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ce_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ce_step_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ce_step_policy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ce_step_policy_sliced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ce_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ce_rollout_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(CeTraj), C.c_uint32, C.c_void_p]),
    "ce_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ce_synth_actions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ce_synth_action_host": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ce_synth_hash_host": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
    "ce_get_buffers": (C.c_int, [C.c_void_p, C.POINTER(CeBuffers)]),
    "ce_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ce_download": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ce_download_many": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(CeFieldReq), C.c_uint32]),
    "ce_upload": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ce_global_view": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "ce_state_bytes": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "ce_get_state": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ce_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "ce_set_cache_budget": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ce_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "ce_host_free": (C.c_int, [C.c_void_p]),
    "ce_download_async": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ce_step_host_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ce_obs_u8_to_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ce_download_obs_f64": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                      C.c_uint32, C.c_void_p]),
    "ce_i16_to_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    "ce_timing_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ce_timing_end": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "ce_selftest": (C.c_int, [C.c_int, C.POINTER(C.c_uint32)]),
    "ce_static_map": (C.c_int, [C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ce_last_error": (C.c_char_p, [C.c_void_p]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """Loads the HIP engine.  Raises if it has not been built (python -m contracts_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP engine library missing: %s — build it with `python -m contracts_amd.build` "
            "(needs hipcc; there is no CPU fallback)" % LIB_PATH)
    if os.environ.get("CONTRACTS_AMD_NO_TORCH") != "1":
        # torch ships its own libamdhip64 (same SONAME); importing it first makes the engine and
        # torch share ONE HIP runtime, so device pointers can be exchanged (torch = plumbing only)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.ce_abi_version() != CE_ABI_VERSION:
        raise EngineError("ABI version mismatch")
    _lib = L
    return L


def check(rc, handle=None, what=""):
    if rc != 0:
        msg = ""
        if handle:
            m = load().ce_last_error(handle)
            msg = m.decode() if m else ""
        raise EngineError("%s failed: %s (%d) %s" % (what, ERRORS.get(rc, "?"), rc, msg))


def static_map(kind):
    """the ASCII rows the engine's tables of a map kind are built from (ce_static_map) — a host-only call"""
    L = load()
    rows, cols = C.c_uint32(), C.c_uint32()
    if L.ce_static_map(KIND[kind], None, 0, C.byref(rows), C.byref(cols)) != 0:
        raise EngineError("%r has no map" % (kind,))
    buf = C.create_string_buffer(rows.value * cols.value)
    if L.ce_static_map(KIND[kind], buf, len(buf), None, None) != 0:
        raise EngineError("ce_static_map(%r) failed" % (kind,))
    text = buf.raw.decode("ascii")
    return [text[r * cols.value:(r + 1) * cols.value] for r in range(rows.value)]
