"""BatchedEnv — tensor-level front end of the HIP engine (one handle = E env replicas on one GPU).

This is the new batched API the reference has no equivalent of (it steps one env per process);
the per-env RLlib-shaped adapters in `contracts_amd.envs` sit on top of it.  All heavy state
lives in HBM and is owned by the C library; this class only moves pointers around.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CeBuffers, CeConfig, check

# contract spaces of the reference (contract/contract_list.py:19-20,42-43,66-67).  The gym Box holds
# float32 bounds and SeparateContractEnv reads them back (two_stage_train.py:39-40), hence float32(0.2).
CONTRACT_SPACE = {
    "cleanup": (0.0, float(np.float32(0.2))),
    "harvest_local": (0.0, 10.0),
    "selfdrive_distprop": (0.0, 100.0),
}
NUM_ACTIONS = {("cleanup", False): 8, ("cleanup", True): 9, ("harvest", False): 7, ("harvest", True): 8,
               ("harvest_features", False): 7, ("cleanup_features", False): 8}

_FIELD_DTYPES = {
    "grid": np.uint8, "agents": np.uint8, "spawn_perm": np.uint8, "waste_perm": np.uint8, "rng": np.uint32,
    "timestep": np.int32, "theta": np.float64, "sd_state": np.float64, "obs": np.uint8, "obs_f64": np.float64,
    "base_reward": np.int32, "reward": np.float64, "done": np.uint8, "done_agents": np.uint8, "info": np.uint8,
    "features": np.int16, "int_metrics": np.int64, "f64_metrics": np.float64, "final_int_metrics": np.int64,
    "final_f64_metrics": np.float64, "error_flags": np.uint32, "debug": np.uint64,
    "beam_map": np.uint8, "sd_info": np.float64, "actions_taken": np.uint8,
}


def make_config(kind, num_envs, num_agents, contract=None, horizon=1000, firing=False, auto_reset=False,
                collective=False, inequity=False, alpha=0.0, beta=0.0, collision_on=False, null_prob=0.0,
                env_index_base=0, device=0, contract_low=None, contract_high=None, external_theta=False,
                beam_trace=False, low_bound=-10.0, high_bound=10.0, start_vel=0.2, start_vel_ambulance=0.8, rng="mt19937",
                ascii_map=None):
    """rng: "mt19937" = the reference's numpy stream (every parity claim is about this mode); "counter" = the engine's own
    Philox4x32-10 stream (CE_FLAG_RNG_COUNTER, grid kinds only): 16 bytes of generator state per env, no np.random.seed trace.
    ascii_map: the reference's constructor argument (a list of equally long strings; grid kinds) — None = the shipped layout;
    a custom one must be walled in and within the shipped layout's frame and cell counts (ce_config.ascii_map, the header)"""
    if rng not in ("mt19937", "counter"):
        raise ValueError("rng must be 'mt19937' or 'counter', got %r" % (rng,))
    cfg = CeConfig()
    if ascii_map is not None:
        rows = [str(r) for r in ascii_map]
        if not rows or any(len(r) != len(rows[0]) for r in rows):
            raise ValueError("ascii_map must be a non-empty list of equally long strings")
        cfg.ascii_map = "".join(rows).encode("ascii")  # (the structure keeps the bytes object alive; ce_create copies the text)
        cfg.map_rows, cfg.map_cols = len(rows), len(rows[0])
    cfg.abi_version = _lib.CE_ABI_VERSION
    cfg.kind = _lib.KIND[kind]
    cfg.num_envs, cfg.num_agents, cfg.horizon = num_envs, num_agents, horizon
    cfg.contract = _lib.CONTRACT[contract]
    cfg.flags = (_lib.FLAG_FIRING * bool(firing) | _lib.FLAG_AUTO_RESET * bool(auto_reset)
                 | _lib.FLAG_COLLECTIVE * bool(collective) | _lib.FLAG_INEQUITY * bool(inequity)
                 | _lib.FLAG_COLLISION * bool(collision_on) | _lib.FLAG_EXTERNAL_THETA * bool(external_theta)
                 | _lib.FLAG_BEAM_TRACE * bool(beam_trace) | _lib.FLAG_RNG_COUNTER * (rng == "counter"))
    cfg.device = device
    cfg.env_index_base = env_index_base
    lo, hi = CONTRACT_SPACE.get(contract, (0.0, 0.0))
    cfg.contract_low = lo if contract_low is None else contract_low
    cfg.contract_high = hi if contract_high is None else contract_high
    cfg.null_prob = null_prob
    cfg.alpha, cfg.beta = alpha, beta
    cfg.low_bound, cfg.high_bound = low_bound, high_bound
    cfg.start_vel, cfg.start_vel_ambulance = start_vel, start_vel_ambulance
    return cfg


class _DevArray:
    """minimal __cuda_array_interface__ carrier so torch/cupy can wrap an engine buffer zero-copy"""

    def __init__(self, ptr, shape, dtype, strides=None, owner=None):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (int(ptr), False), "version": 2,
            "strides": None if strides is None else tuple(strides),
        }
        self._owner = owner


class Trajectory:
    """Per-step outputs of a fused rollout as [P, E, ...] device tensors; step s of a call lands in plane
    (first_plane + s) % P.  `fields` selects which outputs are kept as trajectories (the others go to the engine's
    per-step buffers, overwritten every step)."""

    def __init__(self, env, num_planes, fields=None):
        import torch
        b, E, n, P = env.b, env.E, env.n, int(num_planes)
        dev = "cuda:%d" % env.cfg.device
        if env.kind == "selfdrive":
            shapes = {"obs_f64": ((E, n, 2 * n + 7), torch.float64), "done_agents": ((E, n), torch.uint8),
                      "sd_info": ((E, 2), torch.float64)}
        elif env.kind in _lib.FEAT_KINDS:
            shapes = {"features": ((E, n, b.num_features), torch.int16)}
        else:
            shapes = {"obs": ((E, b.obs_env_stride), torch.uint8), "features": ((E, n, b.num_features), torch.int16)}
        shapes.update({"base_reward": ((E, n), torch.int32), "reward": ((E, n), torch.float64), "done": ((E,), torch.uint8),
                       "info": ((E, n, 2), torch.uint8)})
        self.env, self.P, self.E, self.n = env, P, E, n
        self.tensors = {}
        self.c = _lib.CeTraj()
        self.c.num_planes, self.c.first_plane = P, 0
        self.c.num_envs, self.c.num_agents = E, n  # ABI 4: ce_rollout_fused refuses a ring sized for another batch
        for f, (shape, dt) in shapes.items():
            if fields is not None and f not in fields:
                continue
            t = torch.zeros((P,) + shape, dtype=dt, device=dev)
            self.tensors[f] = t
            setattr(self.c, f, t.data_ptr())

    def host(self, field):
        """host copy [P, E, ...]; obs comes back dense [P, E, n, 15, 15, 3]"""
        a = self.tensors[field].cpu().numpy()
        if field == "obs":
            b, n = self.env.b, self.env.n
            a = a.reshape(self.P, self.env.E, n, b.obs_agent_stride)[..., : 15 * b.obs_row_stride]
            a = np.ascontiguousarray(a.reshape(self.P, self.env.E, n, 15, b.obs_row_stride)[..., :45]).reshape(
                self.P, self.env.E, n, 15, 15, 3)
        return a


class BatchedEnv:
    def __init__(self, kind, num_envs, num_agents, **kw):
        self.kind, self.E, self.n = kind, int(num_envs), int(num_agents)
        self._L = _lib.load()
        self.cfg = make_config(kind, num_envs, num_agents, **kw)
        self.firing = bool(kw.get("firing", False))
        self._h = C.c_void_p()
        rc = self._L.ce_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            msg = self._L.ce_last_error(self._h) if self._h else b""
            if self._h:
                self._L.ce_destroy(self._h)
                self._h = C.c_void_p()
            raise _lib.EngineError("ce_create failed: %s (%d) %s" % (_lib.ERRORS.get(rc, "?"), rc, (msg or b"").decode()))
        self.b = CeBuffers()
        check(self._L.ce_get_buffers(self._h, C.byref(self.b)), self._h, "ce_get_buffers")
        self.num_actions = NUM_ACTIONS.get((kind, self.firing))
        self._tensors = None
        self._pref = None  # prefetch(): host copies of several fields from ONE call, served by download() until the next launch

    # ---- lifecycle -------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.ce_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_contract(self, contract, low=None, high=None, null_prob=0.0):
        self._dirty()
        lo, hi = CONTRACT_SPACE.get(contract, (0.0, 0.0))
        lo = lo if low is None else float(low)
        hi = hi if high is None else float(high)
        check(self._L.ce_set_contract(self._h, _lib.CONTRACT[contract], lo, hi, float(null_prob)), self._h,
              "ce_set_contract")
        self.cfg.contract, self.cfg.contract_low, self.cfg.contract_high = _lib.CONTRACT[contract], lo, hi
        self.cfg.null_prob = float(null_prob)

    def set_flags(self, auto_reset=None, external_theta=None, beam_trace=None):
        """flip the run-time flags of the live handle (see ce_set_flags)"""
        self._dirty()
        mask = value = 0
        for flag, v in ((_lib.FLAG_AUTO_RESET, auto_reset), (_lib.FLAG_EXTERNAL_THETA, external_theta),
                        (_lib.FLAG_BEAM_TRACE, beam_trace)):
            if v is not None:
                mask |= flag
                value |= flag * bool(v)
        check(self._L.ce_set_flags(self._h, mask, value), self._h, "ce_set_flags")
        self.cfg.flags = (self.cfg.flags & ~mask) | value

    # ---- reference-protocol entry points ---------------------------------------------
    def seed(self, seeds=None, seed0=0, mask=None, replay_constructor=True):
        """np.random.seed(s) (+ random.seed(s)) then construct the env (see ce_seed)."""
        self._dirty()
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        m = self._mask(mask)
        if s is not None and s.size != self.E:  # ce_seed reads E entries from the raw pointer
            raise ValueError("seeds must hold one entry per env (%d), got %d" % (self.E, s.size))
        check(self._L.ce_seed(self._h, None if s is None else s.ctypes.data, int(seed0),
                              None if m is None else m.ctypes.data, 3 if replay_constructor else 1), self._h, "ce_seed")

    def _mask(self, mask):
        """host env mask for the C-ABI (which reads exactly E bytes from the pointer)"""
        if mask is None:
            return None
        m = np.ascontiguousarray(mask, np.uint8)
        if m.size != self.E:
            raise ValueError("mask must hold one entry per env (%d), got %d" % (self.E, m.size))
        return m

    def construct(self, mask=None):
        """replay the constructor's RNG use on the CURRENT generator state (no re-seed)"""
        self._dirty()
        m = self._mask(mask)
        check(self._L.ce_seed(self._h, None, 0, None if m is None else m.ctypes.data, 2), self._h, "ce_seed")

    def reset(self, mask=None, stream=None):
        self._dirty()
        m = self._mask(mask)
        check(self._L.ce_reset(self._h, None if m is None else m.ctypes.data, stream), self._h, "ce_reset")

    def step(self, actions, active=None, stream=None):
        """actions: host numpy array [E, n] (uint8 ids / float32 accelerations) — staged by the library."""
        self._dirty()
        dt = np.float32 if self.kind == "selfdrive" else np.uint8
        a = np.ascontiguousarray(actions, dt).reshape(self.E, self.n)
        act = None if active is None else np.ascontiguousarray(active, np.uint8).reshape(self.E, self.n)
        check(self._L.ce_step_host(self._h, a.ctypes.data, None if act is None else act.ctypes.data, stream),
              self._h, "ce_step_host")

    def step_device(self, actions_ptr, active_ptr=None, stream=None):
        """actions_ptr: raw DEVICE pointer (int) to [E, n] actions already resident in HBM."""
        self._dirty()
        check(self._L.ce_step(self._h, actions_ptr, active_ptr, stream), self._h, "ce_step")

    def step_range_device(self, actions_ptr, env_begin, env_count, active_ptr=None, stream=None):
        """step only envs [env_begin, env_begin+env_count); pointers address the full [E, n] planes"""
        self._dirty()
        check(self._L.ce_step_range(self._h, actions_ptr, active_ptr, int(env_begin), int(env_count), stream), self._h,
              "ce_step_range")

    def step_policy_device(self, policy_ptr, mode, env_begin=0, env_count=None, stream=None):
        """ce_step_policy: step envs [env_begin, env_begin + env_count) with the action selection fused into the step kernel —
        mode "bytes" (uint8 [E, n], action = byte mod |A|), "argmax" (float32 [E, n, |A|] scores, first maximum) or "ahead_noise"
        (uint8 [E, n] noise plane, read and written: += green of the pixel ahead in the previous view, action = byte mod |A|);
        the ids taken land in `actions_taken`"""
        self._dirty()
        m = self._POLICY_MODES[mode]
        cnt = self.E - env_begin if env_count is None else env_count
        check(self._L.ce_step_policy(self._h, policy_ptr, m, int(env_begin), int(cnt), stream), self._h, "ce_step_policy")

    _POLICY_MODES = {"bytes": _lib.POLICY_BYTES_MOD, "argmax": _lib.POLICY_ARGMAX_F32, "ahead_noise": _lib.POLICY_AHEAD_NOISE}

    def step_policy_sliced(self, policy_ptr, mode, streams=None, num_slices=None):
        """ce_step_policy_sliced: one policy step of ALL envs as env slices on `streams` (raw hipStream_t handles), the launch
        loop in C — one host call per sampler tick.  mode "ahead_noise": `policy_ptr` is a uint8 [E, n] noise plane the
        launch reads AND writes (the benchmark policy evaluated inside the step kernel, contracts_engine.h)"""
        self._dirty()
        S = len(streams) if streams else (num_slices or 1)
        arr = (C.c_void_p * S)(*streams) if streams else None
        check(self._L.ce_step_policy_sliced(self._h, policy_ptr, self._POLICY_MODES[mode], S, arr), self._h, "ce_step_policy_sliced")

    # ---- host boundary helpers (page-locked staging, asynchronous copies, threaded format conversion) ------------
    def host_alloc(self, shape, dtype):
        """page-locked host array (ce_host_alloc): the DMA target of download_async / source of step_host_async; freed
        when the array is garbage collected"""
        import weakref
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        check(self._L.ce_host_alloc(max(nbytes, 1), C.byref(p)), self._h, "ce_host_alloc")
        raw = (C.c_ubyte * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(raw, np.uint8, nbytes).view(dt).reshape(shape)
        weakref.finalize(raw, self._L.ce_host_free, p.value)  # `arr` keeps `raw` alive through its base chain
        return arr

    def download_async(self, field, dst, env_begin=0, env_count=None, stream=None):
        """field slice -> dst (page-locked array) on `stream`, no host synchronization; synchronize(stream) completes it"""
        cnt = self.E - env_begin if env_count is None else env_count
        check(self._L.ce_download_async(self._h, field.encode(), int(env_begin), int(cnt), dst.ctypes.data, dst.nbytes, stream),
              self._h, "ce_download_async(%s)" % field)

    def step_host_async(self, actions, env_begin=0, env_count=None, stream=None):
        """ce_step_host_async: `actions` is a page-locked uint8 [E, n] plane that stays untouched until the stream has
        passed the copy"""
        self._dirty()
        cnt = self.E - env_begin if env_count is None else env_count
        check(self._L.ce_step_host_async(self._h, actions.ctypes.data, int(env_begin), int(cnt), stream), self._h,
              "ce_step_host_async")

    def obs_u8_to_f64(self, pitched, out, threads):
        """pitched uint8 observation rows (as downloaded raw) -> float64 [envs, n, 15, 15, 3] = value / 255 on `threads`
        host threads (the GIL is released for the duration)"""
        b = self.b
        envs = pitched.size // b.obs_env_stride
        check(self._L.ce_obs_u8_to_f64(pitched.ctypes.data, out.ctypes.data, envs, self.n, b.obs_env_stride, b.obs_agent_stride,
                                       b.obs_row_stride, int(threads)), self._h, "ce_obs_u8_to_f64")

    def download_obs_f64(self, staging, out, threads, parts=4, env_begin=0, env_count=None, stream=None):
        """ce_download_obs_f64: the slice's image observations -> `out` (float64 [envs, n, 15, 15, 3]) through the page-locked
        `staging` block, copy and conversion overlapped part by part; returns with `out` complete (GIL released throughout)"""
        cnt = self.E - env_begin if env_count is None else env_count
        check(self._L.ce_download_obs_f64(self._h, int(env_begin), int(cnt), staging.ctypes.data, staging.nbytes, out.ctypes.data,
                                          out.nbytes, int(parts), int(threads), stream), self._h, "ce_download_obs_f64")

    def i16_to_f64(self, src, out, threads):
        check(self._L.ce_i16_to_f64(src.ctypes.data, out.ctypes.data, src.size, int(threads)), self._h, "ce_i16_to_f64")

    def rollout_device(self, actions_ptr, num_steps, stream_handles=None):
        """num_steps consecutive steps from pre-supplied device action planes [T, E, n]; the launch loop runs
        in C.  stream_handles: list of raw HIP stream handles, one env slice per stream (None = null stream)."""
        self._dirty()
        if stream_handles:
            arr = (C.c_void_p * len(stream_handles))(*stream_handles)
            check(self._L.ce_rollout(self._h, actions_ptr, int(num_steps), len(stream_handles), arr), self._h, "ce_rollout")
        else:
            check(self._L.ce_rollout(self._h, actions_ptr, int(num_steps), 1, None), self._h, "ce_rollout")

    def rollout_fused(self, actions_ptr, num_steps, steps_per_launch=0, traj=None, stream_handles=None):
        """num_steps consecutive steps from device action planes [T, E, n] with ONE launch per steps_per_launch steps
        (0 = a single launch): the env state stays on chip between the steps of a launch, every step still writes all
        of its outputs — to its plane of `traj` (a Trajectory) or, without one, to the per-step buffers.  Bit-identical
        to num_steps step_device() calls (see ce_rollout_fused).  stream_handles: raw HIP stream handles, one contiguous
        env slice per stream (None = one slice on the null stream)."""
        self._dirty()
        if traj is not None and (traj.E, traj.n, traj.env.kind) != (self.E, self.n, self.kind):
            # (ce_rollout_fused checks E and n itself since ABI 4 — ce_traj.num_envs / num_agents; the kind only exists here)
            raise ValueError("trajectory was allocated for %s E=%d n=%d, this handle is %s E=%d n=%d"
                             % (traj.env.kind, traj.E, traj.n, self.kind, self.E, self.n))
        t = None if traj is None else C.byref(traj.c)
        ns, arr = 1, None
        if stream_handles:
            ns, arr = len(stream_handles), (C.c_void_p * len(stream_handles))(*stream_handles)
        check(self._L.ce_rollout_fused(self._h, actions_ptr, int(num_steps), int(steps_per_launch), t, ns, arr), self._h,
              "ce_rollout_fused")
        if traj is not None:
            traj.c.first_plane = (traj.c.first_plane + int(num_steps)) % traj.c.num_planes

    def global_view(self, out_ptr, env_begin=0, env_count=None, stream=None):
        """MapEnv.global_view of the env slice into a DEVICE buffer uint8 [count, grid_h, grid_w, 3] (ce_global_view)"""
        cnt = self.E - env_begin if env_count is None else int(env_count)
        check(self._L.ce_global_view(self._h, int(env_begin), cnt, out_ptr, stream), self._h, "ce_global_view")

    def alloc_trajectory(self, num_planes, fields=None):
        """device-resident trajectory arrays [P, E, ...] for rollout_fused (torch owns the memory: plumbing only)"""
        return Trajectory(self, num_planes, fields)

    def synth_actions(self, key, t0, T, out_ptr, stream=None):
        check(self._L.ce_synth_actions(self._h, int(key), int(t0), int(T), out_ptr, stream), self._h, "ce_synth_actions")

    def synchronize(self, stream=None):
        check(self._L.ce_synchronize(self._h, stream), self._h, "ce_synchronize")

    def timing_begin(self, stream=None):
        check(self._L.ce_timing_begin(self._h, stream), self._h, "ce_timing_begin")

    def timing_end(self, stream=None):
        ms, cnt = C.c_double(), C.c_uint32()
        check(self._L.ce_timing_end(self._h, stream, C.byref(ms), C.byref(cnt)), self._h, "ce_timing_end")
        return ms.value, cnt.value

    # ---- host copies -----------------------------------------------------------------
    def _env_shape(self, field):
        b, n = self.b, self.n
        return {
            "grid": (self._grid_image_stride(),), "agents": (n, 4), "spawn_perm": (20,), "waste_perm": (119,),
            "rng": (b.rng_words,), "timestep": (), "theta": (), "sd_state": (5 * n + 3,),
            "obs": (b.obs_env_stride,), "obs_f64": (n, 2 * n + 7), "base_reward": (n,), "reward": (n,), "done": (),
            "done_agents": (n,), "info": (n, 2), "features": (n, b.num_features), "int_metrics": (b.num_int_metrics,),
            "f64_metrics": (b.num_f64_metrics,), "final_int_metrics": (b.num_int_metrics,),
            "final_f64_metrics": (b.num_f64_metrics,), "error_flags": (), "debug": (16,),
            "beam_map": (b.grid_h, b.grid_w), "sd_info": (2,), "actions_taken": (n,),
        }[field]

    def prefetch(self, fields):
        """fetch several fields in one call (ce_download_many); download() serves them until the next launch / upload.
        The cached arrays are handed out read-only (a caller that wants to edit one copies it): mutating a shared cache
        entry would silently corrupt every later download() of that field."""
        self._pref = self.download_many([f for f in fields if f in self._fields_present()])
        for a in self._pref.values():
            a.setflags(write=False)

    def _fields_present(self):
        b = self.b
        return {f for f in _FIELD_DTYPES if f != "debug" and getattr(b, f, None)}

    def _dirty(self):
        self._pref = None

    def download(self, field, env_begin=0, env_count=None, raw=False):
        """host copy of a field; grid/obs are returned as [E,H,W] / [E,n,15,15,3] unless raw.  After `prefetch()` a
        whole-batch request is served from the prefetched snapshot, which is shared and READ-ONLY: copy it before editing in
        place (e.g. `rng = env.download("rng").copy()` ahead of an `upload`)."""
        cnt = self.E - env_begin if env_count is None else env_count
        if self._pref is not None and field in self._pref and env_begin == 0 and cnt == self.E:
            out = self._pref[field]
            return self._dense_obs(out) if (field == "obs" and not raw) else out
        out = np.empty((cnt,) + self._env_shape(field), _FIELD_DTYPES[field])
        check(self._L.ce_download(self._h, field.encode(), env_begin, cnt, out.ctypes.data, out.nbytes), self._h,
              "ce_download(%s)" % field)
        if raw or (field == "grid" and self.kind in _lib.FEAT_KINDS):  # feature kinds: the raw list-stamp block
            return out
        if field == "grid":  # bordered image -> dense [cnt, H, W]
            return np.ascontiguousarray(self._grid_interior(out))
        if field == "obs":  # pitched rows (obs_row_stride bytes) -> dense [cnt, n, 15, 15, 3]
            b = self.b
            return np.ascontiguousarray(out.reshape(cnt, self.n, b.obs_agent_stride)[:, :, : 15 * b.obs_row_stride]
                                        .reshape(cnt, self.n, 15, b.obs_row_stride)[:, :, :, :45]).reshape(cnt, self.n, 15, 15, 3)
        return out

    def download_many(self, fields, env_begin=0, env_count=None):
        """{field: host array} for several fields of one env slice in one call (one device synchronize, one staged copy
        for small slices — what the per-env adapters use for a whole step result).  Arrays come back RAW (obs pitched
        as in HBM, see _dense_obs); "grid" is not accepted (download() expands it)."""
        cnt = self.E - env_begin if env_count is None else env_count
        out = {f: np.empty((cnt,) + self._env_shape(f), _FIELD_DTYPES[f]) for f in fields}
        reqs = (_lib.CeFieldReq * len(fields))()
        for i, f in enumerate(fields):
            reqs[i].field, reqs[i].dst, reqs[i].dst_bytes = f.encode(), out[f].ctypes.data, out[f].nbytes
        check(self._L.ce_download_many(self._h, env_begin, cnt, reqs, len(fields)), self._h, "ce_download_many")
        return out

    def _dense_obs(self, raw):
        """pitched observation rows [cnt, obs_env_stride] -> dense uint8 [cnt, n, 15, 15, 3]"""
        b, cnt = self.b, raw.shape[0]
        return np.ascontiguousarray(raw.reshape(cnt, self.n, b.obs_agent_stride)[:, :, : 15 * b.obs_row_stride]
                                    .reshape(cnt, self.n, 15, b.obs_row_stride)[:, :, :, :45]).reshape(cnt, self.n, 15, 15, 3)

    def _grid_image_stride(self):
        """bytes per env of the "grid" download / upload format: the padded map image for the grid kinds (the device
        keeps 32 B of presence bits per env, ce_download expands them), the raw list-stamp block for the feature kinds"""
        b = self.b
        if self.kind in _lib.FEAT_KINDS or self.kind == "selfdrive":
            return b.grid_env_stride
        frame_rows = {"cleanup": 25, "harvest": 16}[self.kind]  # (a custom layout sits at the origin of the kind's frame)
        return ((frame_rows + 14) * b.grid_row_stride + 15) // 16 * 16

    def _grid_interior(self, raw):
        """strided [cnt, H, W] view of the interior of raw bordered grid slices [cnt, grid_env_stride]"""
        b = self.b
        return np.lib.stride_tricks.as_strided(raw[:, b.grid_origin:], shape=(raw.shape[0], b.grid_h, b.grid_w),
                                               strides=(raw.strides[0], b.grid_row_stride, 1))

    def upload(self, field, array, env_begin=0):
        self._dirty()
        arr = np.asarray(array)
        cnt = arr.shape[0]
        if field == "grid" and arr.ndim == 3:
            raw = np.zeros((cnt, self._grid_image_stride()), np.uint8)
            self._grid_interior(raw)[...] = arr
            arr = raw
        arr = np.ascontiguousarray(arr, _FIELD_DTYPES[field]).reshape((cnt,) + self._env_shape(field))
        check(self._L.ce_upload(self._h, field.encode(), env_begin, cnt, arr.ctypes.data, arr.nbytes), self._h,
              "ce_upload(%s)" % field)

    # ---- env-state checkpoint (the reference never checkpoints env state) -----------------
    _STATE_GRID = ("grid", "agents", "spawn_perm", "waste_perm", "rng", "timestep", "theta", "int_metrics",
                   "f64_metrics", "final_int_metrics", "final_f64_metrics", "done", "error_flags")
    _STATE_SD = ("sd_state", "rng", "theta", "f64_metrics", "final_f64_metrics", "int_metrics", "final_int_metrics",
                 "done", "done_agents", "error_flags")

    def state_dict(self):
        """host copy of every persistent field: stepping from a restored state is bit-identical"""
        if self.kind in _lib.FEAT_KINDS:
            fields = [f for f in self._STATE_GRID if f not in ("spawn_perm", "waste_perm")]
        else:
            fields = self._STATE_SD if self.kind == "selfdrive" else [
                f for f in self._STATE_GRID if not (f == "waste_perm" and self.kind != "cleanup")]
        out = {f: np.array(self.download(f, raw=True)) for f in fields}
        out["_meta"], out["_meta_f64"] = self._meta()
        return out

    def _meta(self):
        """what a checkpoint must agree on for the continuation to be bit-identical: the layout (ABI version: field
        shapes such as the metric rows and the grid pitch follow it), the batch, and every parameter that enters a step"""
        c = self.cfg
        return (np.array([_lib.KIND[self.kind], self.E, self.n, c.contract, c.flags, c.horizon, _lib.CE_ABI_VERSION,
                          c.env_index_base], np.int64),
                np.array([c.contract_low, c.contract_high, c.null_prob, c.alpha, c.beta, c.low_bound, c.high_bound,
                          c.start_vel, c.start_vel_ambulance], np.float64))

    _META_NAMES = ("kind", "num_envs", "num_agents", "contract", "flags", "horizon", "abi_version", "env_index_base")
    _META_F64_NAMES = ("contract_low", "contract_high", "null_prob", "alpha", "beta", "low_bound", "high_bound", "start_vel",
                       "start_vel_ambulance")

    def load_state_dict(self, state):
        meta = [int(x) for x in state["_meta"]]
        mine_i, mine_f = self._meta()
        mine = [int(x) for x in mine_i]
        if len(meta) < len(mine) or "_meta_f64" not in state:
            raise ValueError("checkpoint predates the versioned format (no ABI version / float parameters in _meta): it was "
                             "written by an older engine build whose field layout this one (ABI v%d) cannot verify"
                             % _lib.CE_ABI_VERSION)
        if meta[6] != mine[6]:
            raise ValueError("checkpoint was written by engine ABI v%d, this build is ABI v%d (field layouts differ)"
                             % (meta[6], mine[6]))
        if meta[:3] != mine[:3]:
            raise ValueError("checkpoint is for kind/E/n %s, engine has %s" % (meta[:3], mine[:3]))
        bad = [n for n, a, b in zip(self._META_NAMES, meta, mine) if a != b]
        theirs_f = [float(x) for x in state["_meta_f64"]]
        bad += [n for n, a, b in zip(self._META_F64_NAMES, theirs_f, mine_f) if a != float(b)]
        if bad:  # stepping on would silently not be bit-identical to the run that was saved
            raise ValueError("checkpoint and engine disagree on %s: continuing would not reproduce the saved run" % ", ".join(bad))
        # error_flags first: the validated grid upload may raise CE_FAULT_BAD_GRID, which must survive the restore
        for f in sorted((f for f in state if not f.startswith("_meta")), key=lambda f: f != "error_flags"):
            self.upload(f, state[f])
        self.check_faults()

    # ---- the C-ABI's one-call snapshot (ce_state_bytes / ce_get_state / ce_set_state): save / load are thin wrappers ----
    def get_state(self, outputs=False):
        """the handle's whole persistent state (with outputs=True also the last step's outputs) as ONE self-describing blob
        (uint8 array: ce_state_header + directory + fields) — what a C caller gets from ce_get_state"""
        what = _lib.STATE_OUTPUTS if outputs else 0
        nbytes = C.c_uint64()
        check(self._L.ce_state_bytes(self._h, what, C.byref(nbytes)), self._h, "ce_state_bytes")
        blob = np.empty(nbytes.value, np.uint8)
        check(self._L.ce_get_state(self._h, what, blob.ctypes.data, blob.nbytes), self._h, "ce_get_state")
        return blob

    def set_state(self, blob):
        """restores a blob of get_state(); raises (CE_EINVAL + the reason) when blob and handle disagree on anything that
        enters a step — the handle is untouched then"""
        blob = np.ascontiguousarray(blob, np.uint8)
        self._dirty()
        check(self._L.ce_set_state(self._h, blob.ctypes.data, blob.nbytes), self._h, "ce_set_state")
        self.check_faults()

    @staticmethod
    def state_fields(blob):
        """{field: array [E, row bytes]} views into a blob, by its directory (no field list hard-coded on this side)"""
        blob = np.ascontiguousarray(blob, np.uint8)
        hd = _lib.CeStateHeader.from_buffer_copy(blob[:C.sizeof(_lib.CeStateHeader)].tobytes())
        if hd.magic != _lib.STATE_MAGIC:
            raise ValueError("not a ce_get_state blob")
        out, off = {}, C.sizeof(_lib.CeStateHeader)
        for i in range(hd.num_fields):
            d = _lib.CeStateField.from_buffer_copy(blob[off + i * C.sizeof(_lib.CeStateField):off + (i + 1) * C.sizeof(_lib.CeStateField)].tobytes())
            out[d.name.decode()] = blob[d.offset:d.offset + d.env_bytes * hd.num_envs].reshape(hd.num_envs, d.env_bytes)
        return out

    def save(self, path, outputs=False):
        """checkpoint file = the ce_get_state blob (compressed .npz with one member)"""
        np.savez_compressed(path, ce_state=self.get_state(outputs))

    def load(self, path):
        with np.load(path) as z:
            if "ce_state" in z.files:
                self.set_state(z["ce_state"])
            else:  # a state_dict() written field by field (rounds 1-5)
                self.load_state_dict({k: z[k] for k in z.files})

    def set_cache_budget(self, nbytes):
        """share of the GPU's last-level cache this handle may assume (ce_set_cache_budget): 0 = never write observations through"""
        check(self._L.ce_set_cache_budget(self._h, int(nbytes)), self._h, "ce_set_cache_budget")

    def feature_state(self):
        """feature kinds: (apple_stamp u16 [E, 160], waste_stamp u16 [E, 120], next_stamp u32 [E, 2]) — the rank of each
        present cell in the reference's current_apple_points / current_waste_points list, 0xffff = absent"""
        st = self.download("grid", raw=True)
        a, w = 2 * _lib.FEAT_APPLE_SLOTS, 2 * _lib.FEAT_WASTE_SLOTS
        return (np.ascontiguousarray(st[:, :a]).view(np.uint16), np.ascontiguousarray(st[:, a:a + w]).view(np.uint16),
                np.ascontiguousarray(st[:, a + w:]).view(np.uint32))

    def __getattr__(self, name):  # oracle-compatible attribute access = fresh host copy
        if name in ("apple_stamp", "waste_stamp", "next_stamp") and self.__dict__.get("kind") in _lib.FEAT_KINDS:
            return self.feature_state()[("apple_stamp", "waste_stamp", "next_stamp").index(name)]
        if name in _FIELD_DTYPES and "b" in self.__dict__:
            return self.download(name)
        raise AttributeError(name)

    def check_faults(self):
        f = self.download("error_flags")
        if f.any():
            bad = np.nonzero(f)[0]
            raise _lib.EngineError("env faults: %s" % {int(i): int(f[i]) for i in bad[:8]})

    # ---- zero-copy device views ------------------------------------------------------
    def device_arrays(self):
        """{field: object with __cuda_array_interface__} over the engine's HBM buffers"""
        b, E, n = self.b, self.E, self.n
        out = {}
        if self.kind in _lib.FEAT_KINDS:
            out["features"] = _DevArray(b.features, (E, n, b.num_features), np.int16, None, self)
            out["base_reward"] = _DevArray(b.base_reward, (E, n), np.int32, None, self)
        elif self.kind != "selfdrive":
            out["obs"] = _DevArray(b.obs, (E, n, 15, 15, 3), np.uint8, (b.obs_env_stride, b.obs_agent_stride, b.obs_row_stride, 3, 1), self)
            out["grid_bits"] = _DevArray(b.grid, (E, 8), np.uint32, None, self)  # packed presence bits (see the header)
            out["features"] = _DevArray(b.features, (E, n, b.num_features), np.int16, None, self)
            out["base_reward"] = _DevArray(b.base_reward, (E, n), np.int32, None, self)
        else:
            out["obs_f64"] = _DevArray(b.obs_f64, (E, n, 2 * n + 7), np.float64, None, self)
            out["done_agents"] = _DevArray(b.done_agents, (E, n), np.uint8, None, self)
            out["sd_info"] = _DevArray(b.sd_info, (E, 2), np.float64, None, self)
        out["reward"] = _DevArray(b.reward, (E, n), np.float64, None, self)
        out["done"] = _DevArray(b.done, (E,), np.uint8, None, self)
        out["info"] = _DevArray(b.info, (E, n, 2), np.uint8, None, self)
        out["theta"] = _DevArray(b.theta, (E,), np.float64, None, self)
        return out

    def torch_tensors(self):
        """torch views (no copies) of the output buffers; torch is plumbing only"""
        if self._tensors is None:
            import torch
            dev = "cuda:%d" % self.cfg.device
            self._tensors = {k: torch.as_tensor(v, device=dev) for k, v in self.device_arrays().items()}
        return self._tensors
