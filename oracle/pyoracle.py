"""ctypes front-end of the CPU oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg — never by
the product package `contracts_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONTRACTS_ORACLE_LIB") or os.path.join(HERE, "_build", "liboracle.so")  # (override: the sanitizer build)

KIND = {"cleanup": 0, "harvest": 1, "selfdrive": 2, "harvest_features": 3, "cleanup_features": 4}
FEAT_KINDS = ("harvest_features", "cleanup_features")
FEAT_APPLE_SLOTS, FEAT_WASTE_SLOTS, FEAT_STATE_BYTES = 160, 120, 568
CONTRACT = {None: 0, "none": 0, "cleanup": 1, "harvest_local": 2, "selfdrive_distprop": 3}
FLAG_FIRING, FLAG_AUTO_RESET, FLAG_COLLECTIVE, FLAG_INEQUITY, FLAG_COLLISION, FLAG_EXTERNAL_THETA = 1, 2, 4, 8, 16, 32


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


# default contract spaces of the reference (contract_list.py:19-20,42-43,66-67); the Box holds
# float32 bounds, the wrapper reads them back as float64 (two_stage_train.py:39-40)
CONTRACT_SPACE = {
    "cleanup": (0.0, float(np.float32(0.2))),
    "harvest_local": (0.0, 10.0),
    "selfdrive_distprop": (0.0, 100.0),
}


def make_config(kind, num_envs, num_agents, contract=None, horizon=1000, firing=False, auto_reset=False,
                collective=False, inequity=False, alpha=0.0, beta=0.0, collision_on=False, null_prob=0.0,
                env_index_base=0, device=0, contract_low=None, contract_high=None, external_theta=False,
                beam_trace=False, rng="mt19937", ascii_map=None):
    cfg = CeConfig()
    if ascii_map is not None:  # the reference's `ascii_map` argument: a list of equally long strings (grid kinds)
        rows = [str(r) for r in ascii_map]
        if not rows or any(len(r) != len(rows[0]) for r in rows):
            raise ValueError("ascii_map must be a non-empty list of equally long strings")
        cfg.ascii_map = "".join(rows).encode("ascii")  # (ctypes keeps the bytes object alive with the structure)
        cfg.map_rows, cfg.map_cols = len(rows), len(rows[0])
    cfg.abi_version = 4  # CE_ABI_VERSION of include/contracts_engine.h (the oracle refuses any other)
    cfg.kind = KIND[kind]
    cfg.num_envs = num_envs
    cfg.num_agents = num_agents
    cfg.horizon = horizon
    cfg.contract = CONTRACT[contract]
    cfg.flags = (FLAG_FIRING * bool(firing) | FLAG_AUTO_RESET * bool(auto_reset) | FLAG_COLLECTIVE * bool(collective)
                 | FLAG_INEQUITY * bool(inequity) | FLAG_COLLISION * bool(collision_on) | FLAG_EXTERNAL_THETA * bool(external_theta)
                 | 64 * bool(beam_trace) | 128 * (rng == "counter"))
    cfg.device = device
    cfg.env_index_base = env_index_base
    lo, hi = CONTRACT_SPACE.get(contract, (0.0, 0.0))
    cfg.contract_low = lo if contract_low is None else contract_low
    cfg.contract_high = hi if contract_high is None else contract_high
    cfg.null_prob = null_prob
    cfg.alpha, cfg.beta = alpha, beta
    cfg.low_bound, cfg.high_bound, cfg.start_vel, cfg.start_vel_ambulance = -10.0, 10.0, 0.2, 0.8
    return cfg


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "oracle.c")):
        target = [os.path.join("_build", os.path.basename(LIB_PATH))] if os.environ.get("CONTRACTS_ORACLE_LIB") else []
        subprocess.check_call(["make", "-C", HERE, "-s"] + target, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_create.argtypes = [C.POINTER(CeConfig), C.POINTER(C.c_void_p)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_buffers.argtypes = [C.c_void_p, C.POINTER(CeBuffers)]
        L.orc_import_state.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_rng_words.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        L.orc_rng_shuffle.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_int]
        L.orc_rng_double.argtypes = [C.c_uint32, C.c_int, C.c_int]
        L.orc_rng_double.restype = C.c_double
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_counter_stream.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
        for f in ("orc_create", "orc_destroy", "orc_seed", "orc_reset", "orc_step", "orc_get_buffers", "orc_import_state"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _view(ptr, dtype, shape):
    n = int(np.prod(shape))
    if not ptr or n == 0:
        return None
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def obs_view(raw, n, agent_stride=675, row_stride=45):
    """[E][obs_env_stride] raw bytes -> zero-copy [E][n][15][15][3] view"""
    E, stride = raw.shape
    return np.lib.stride_tricks.as_strided(raw, shape=(E, n, 15, 15, 3), strides=(stride, agent_stride, row_stride, 3, 1),
                                           writeable=False)


def buffer_views(b, kind):
    """numpy views (host memory) with the shapes documented in include/contracts_engine.h"""
    E, n = b.num_envs, b.num_agents
    v = {}
    if kind in FEAT_KINDS:
        st = _view(b.grid, np.uint8, (E, FEAT_STATE_BYTES))
        v["grid"] = st  # raw state block; typed views below
        v["apple_stamp"] = st[:, :2 * FEAT_APPLE_SLOTS].view(np.uint16)
        v["waste_stamp"] = st[:, 2 * FEAT_APPLE_SLOTS:2 * (FEAT_APPLE_SLOTS + FEAT_WASTE_SLOTS)].view(np.uint16)
        v["next_stamp"] = st[:, 2 * (FEAT_APPLE_SLOTS + FEAT_WASTE_SLOTS):].view(np.uint32)
        v["agents"] = _view(b.agents, np.uint8, (E, n, 4))
        v["rng"] = _view(b.rng, np.uint32, (E, b.rng_words))
        v["features"] = _view(b.features, np.int16, (E, n, b.num_features))
    elif kind != "selfdrive":
        cells = b.grid_h * b.grid_w
        v["grid"] = _view(b.grid, np.uint8, (E, b.grid_env_stride))[:, :cells].reshape(E, b.grid_h, b.grid_w)  # dense in the oracle
        v["agents"] = _view(b.agents, np.uint8, (E, n, 4))
        v["spawn_perm"] = _view(b.spawn_perm, np.uint8, (E, 20))
        v["waste_perm"] = _view(b.waste_perm, np.uint8, (E, 119))
        v["rng"] = _view(b.rng, np.uint32, (E, b.rng_words))
        v["obs"] = obs_view(_view(b.obs, np.uint8, (E, b.obs_env_stride)), n, b.obs_agent_stride, b.obs_row_stride)
        v["features"] = _view(b.features, np.int16, (E, n, b.num_features))
        v["beam_map"] = _view(b.beam_map, np.uint8, (E, b.grid_h, b.grid_w))
    else:
        v["rng"] = _view(b.rng, np.uint32, (E, b.rng_words))
        v["sd_state"] = _view(b.sd_state, np.float64, (E, 5 * n + 3))
        v["obs_f64"] = _view(b.obs_f64, np.float64, (E, n, 2 * n + 7))
        v["done_agents"] = _view(b.done_agents, np.uint8, (E, n))
        v["sd_info"] = _view(b.sd_info, np.float64, (E, 2))
    v["timestep"] = _view(b.timestep, np.int32, (E,))
    v["theta"] = _view(b.theta, np.float64, (E,))
    v["base_reward"] = _view(b.base_reward, np.int32, (E, n))
    v["reward"] = _view(b.reward, np.float64, (E, n))
    v["done"] = _view(b.done, np.uint8, (E,))
    v["info"] = _view(b.info, np.uint8, (E, n, 2))
    v["int_metrics"] = _view(b.int_metrics, np.int64, (E, b.num_int_metrics))
    v["f64_metrics"] = _view(b.f64_metrics, np.float64, (E, b.num_f64_metrics))
    v["final_int_metrics"] = _view(b.final_int_metrics, np.int64, (E, b.num_int_metrics))
    v["final_f64_metrics"] = _view(b.final_f64_metrics, np.float64, (E, b.num_f64_metrics))
    v["error_flags"] = _view(b.error_flags, np.uint32, (E,))
    return v


class Oracle:
    """E independent env replicas stepped on the CPU by the C restatement."""

    def __init__(self, kind, num_envs, num_agents, **kw):
        self.kind = kind
        self.cfg = make_config(kind, num_envs, num_agents, **kw)
        self.E, self.n = num_envs, num_agents
        self._h = C.c_void_p()
        rc = lib().orc_create(C.byref(self.cfg), C.byref(self._h))
        if rc:
            raise RuntimeError("orc_create failed: %d" % rc)
        self._b = CeBuffers()
        lib().orc_get_buffers(self._h, C.byref(self._b))
        self.buf = buffer_views(self._b, kind)

    def __getattr__(self, k):
        buf = self.__dict__.get("buf")
        if buf is not None and k in buf:
            return buf[k]
        raise AttributeError(k)

    def seed(self, seeds=None, seed0=0, mask=None, replay_constructor=True):
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = lib().orc_seed(self._h, None if s is None else s.ctypes.data, seed0, None if m is None else m.ctypes.data,
                            3 if replay_constructor else 1)
        assert rc == 0, rc

    def construct(self, mask=None):
        """constructor RNG replay on the current generator state (no re-seed)"""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        assert lib().orc_seed(self._h, None, 0, None if m is None else m.ctypes.data, 2) == 0

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = lib().orc_reset(self._h, None if m is None else m.ctypes.data)
        assert rc == 0, rc

    def step(self, actions, active=None):
        if self.kind == "selfdrive":
            a = np.ascontiguousarray(actions, np.float32).reshape(self.E, self.n)
        else:
            a = np.ascontiguousarray(actions, np.uint8).reshape(self.E, self.n)
        act = None if active is None else np.ascontiguousarray(active, np.uint8).reshape(self.E, self.n)
        rc = lib().orc_step(self._h, a.ctypes.data, None if act is None else act.ctypes.data)
        assert rc == 0, rc

    def import_state(self, env=None):
        """push edits made to the exported state arrays (grid/agents/perm/rng/theta/…) back in"""
        for ei in (range(self.E) if env is None else [env]):
            assert lib().orc_import_state(self._h, ei) == 0

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
