"""BatchedBaseEnv — RLlib `BaseEnv`-shaped vector hook over one engine handle.

RLlib samples through `BaseEnv.poll() / send_actions() / try_reset()`; a `MultiAgentEnv` is normally wrapped one
Python object per sub-env (`to_base_env`, call sites utils/ray_config_utils.py:126-214 with `num_envs_per_worker`).
This class serves the same protocol for E sub-envs from ONE engine handle: `send_actions` is one kernel launch for all
E envs, `poll` hands back the per-env / per-agent dictionaries RLlib expects, `try_reset(env_id)` a masked `ce_reset`.
Observations, rewards, dones and infos have exactly the shapes of the single-env adapters (`environments/*.py`),
including the contract wrapper's extra observation entry.

Built to stay usable at E = 16 384:
  * one device -> host copy per field and tick (a snapshot), never per env;
  * the per-env dictionaries are LAZY: `poll()` returns mappings that build an env's entry when it is looked up, straight
    from the snapshot arrays (an env nobody looks at costs nothing);
  * resets are batched (`batch_done_resets=True`, the default): the first `try_reset` after a tick resets EVERY env that
    reported done in ONE masked launch and one copy; the following `try_reset(e)` calls are served from that batch — a
    synchronized horizon (all E envs done in the same tick) costs O(E), not O(E^2).  SIDE EFFECT: a done env the
    caller never asks about is reset as well (RLlib's sampler resets every done sub-env, so it never notices);
    `batch_done_resets=False` resets exactly the env asked for;
  * `poll_tensors()` / `send_actions_array()` skip Python containers altogether (observations stay in HBM);
  * `recycle_dicts` (grid and feature-vector kinds): the dictionaries of a tick are not rebuilt but RECYCLED.  Two generations
    of complete dictionary trees ({env: {agent: {"image": ..}}}, rewards, dones, infos) are kept over page-locked snapshot
    buffers; a tick copies the step's results into the older generation asynchronously (one DMA per field), converts the
    observations to the reference's float64 on worker threads, and rewrites only the dictionary entries whose values
    changed (C loops, `csrc/ce_pydict.c`) — ~8 ms per tick at E = 16 384 instead of ~290 ms.  CONTRACT: everything `poll()`
    returned at tick t — observation arrays, the info dictionaries and their `feature_obs` arrays, reward and done
    dictionaries — stays intact through tick t + 1 and is REWRITTEN IN PLACE by tick t + 2.  A consumer may therefore keep a
    reference for one tick at most; anything it wants longer it must copy.
      - `"auto"` (the default): recycled for the grid kinds, whose observation space is a Dict — RLlib's
        DictFlatteningPreprocessor copies every observation into a fresh flat array before its collectors see it.  The
        feature-vector kinds, whose Box observations RLlib's NoPreprocessor / NoFilter pass through by reference into the
        rollout fragment (a recycled buffer would alias every row of it), keep the recycled machinery but get FRESH
        observation and `feature_obs` rows every tick (`recycle_dicts` reads "fresh_obs": one [E, n, F] copy, per-env
        observation dictionaries over its rows) and per-tick copies of the info dictionaries.  Under "auto" the GRID kinds' info
        dictionaries are still recycled: a consumer that keeps `infos` beyond one tick (SampleBatch.INFOS does) and reads their
        scalar entries later wants `False`.
      - `True` / `False`: force either path.
      - `"checked"` (debug): the recycled data path with the contract ENFORCED — the generation about to be rewritten is
        poisoned first (NaN fill of its observation / feature blocks), and every array / info dictionary handed out is
        stamped with its generation's epoch and raises `StaleDictError` when it is read after that generation moved on.  Run
        a new sampler under it once; it costs the per-tick construction of the stamped wrappers (as slow as `False`).

When `ray` is importable the class derives from `ray.rllib.env.BaseEnv`, otherwise it is duck-typed; nothing else in it
depends on RLlib.  Each sub-env keeps a private RNG stream seeded `seed0 + env_index_base + i` (the batched API's
convention), not the process-global generator.
"""
from collections.abc import Mapping

import numpy as np

from . import _lib
from .engine import BatchedEnv
from .environments.metrics import episode_metrics

try:  # pragma: no cover - RLlib is absent in the build image
    from ray.rllib.env import BaseEnv as _RLlibBaseEnv
except Exception:
    _RLlibBaseEnv = object

_GRID = ("cleanup", "harvest")
_SECOND_INFO = {"cleanup": "cleaned_squares", "harvest": "eaten_close_apples", "harvest_features": "eaten_close_apples",
                "cleanup_features": "cleaned_squares"}


_POOL = None
_CHUNK = 256  # envs per conversion job


def _pool():
    """worker threads for the uint8 -> float64 observation conversion (numpy releases the GIL inside the ufunc)"""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) - 1)), thread_name_prefix="ce-obs")
    return _POOL


class _gc_paused:
    """Holds the cyclic garbage collector off while a tick's dictionaries are built.  A 16 384-env tick allocates ~10^5
    containers in a few milliseconds; with the collector's default thresholds that is one full collection per tick, and a full
    collection walks every tracked object of the process (torch and numpy included) — 45-70 ms landing on whichever allocation
    crosses the count, several times the tick itself (tools/joint_dict_rate.py).  None of these objects is cyclic: reference
    counts free them.  The collector's state is restored on exit; nothing is frozen or exempted from later collections."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self._was:
            import gc
            gc.enable()
        return False


class _ImageChunks:
    """the tick's observations as float64 (`uint8 / 255`, cleanup_new.py:258 / harvest_new.py:229), converted in chunks
    of _CHUNK envs by background threads while the caller walks the envs; small batches convert on demand"""

    def __init__(self, raw, out=None):
        """`out`: a float64 block of raw's shape to convert into (a caller's recycled buffer) instead of fresh arrays"""
        self.raw = raw
        self.jobs = None
        if out is not None:
            pool = _pool()
            self.jobs = [pool.submit(np.true_divide, raw[c:c + _CHUNK], 255, out[c:c + _CHUNK]) for c in range(0, raw.shape[0], _CHUNK)]
        elif raw.shape[0] >= 4 * _CHUNK:
            pool = _pool()
            self.jobs = [pool.submit(np.true_divide, raw[c:c + _CHUNK], 255) for c in range(0, raw.shape[0], _CHUNK)]

    def env(self, j):
        if self.jobs is None:
            return self.raw[j] / 255
        return self.jobs[j // _CHUNK].result()[j % _CHUNK]

    def rows(self):
        """every env's image in env order (waits for the conversion chunk by chunk)"""
        if self.jobs is None:
            return list(self.raw / 255)
        out = []
        for job in self.jobs:
            out.extend(job.result())
        return out


class _LazyEnvMap(Mapping):
    """{env_id: value} over a fixed id list; a value is built by `build(env_id)` when it is looked up (once)"""

    def __init__(self, ids, build):
        self._ids = list(ids)
        self._set = set(self._ids)
        self._build = build
        self._cache = {}

    def __getitem__(self, e):
        try:
            return self._cache[e]
        except KeyError:
            pass
        if e not in self._set:
            raise KeyError(e)
        v = self._cache[e] = self._build(e)
        return v

    def __iter__(self):
        return iter(self._ids)

    def __len__(self):
        return len(self._ids)


class _SubEnvView:
    """one sub-env as callbacks see it: `.metrics` (computed on access), `.base_env` (the reference's wrapper chain)"""

    def __init__(self, venv, env_id):
        self._venv, self.env_id = venv, env_id

    @property
    def metrics(self):
        return self._venv.env_metrics(self.env_id)

    @property
    def base_env(self):
        return self

    @property
    def num_agents(self):
        return self._venv.num_agents


class _SubEnvs:
    """list-like over the E sub-env views, built on access (nobody wants 16 384 objects per worker)"""

    def __init__(self, venv):
        self._venv = venv

    def __len__(self):
        return self._venv.num_envs

    def __getitem__(self, e):
        if isinstance(e, slice):
            return [self[i] for i in range(*e.indices(len(self)))]
        if e < 0:
            e += len(self)
        if not 0 <= e < len(self):
            raise IndexError(e)
        return _SubEnvView(self._venv, e)

    def __iter__(self):
        return (self[e] for e in range(len(self)))


class StaleDictError(RuntimeError):
    """recycle_dicts="checked": something poll() handed out two or more ticks ago was read after its buffers were rewritten"""


def _stale(owner, epoch):
    if owner is not None and owner.epoch != epoch:
        raise StaleDictError("an object handed out by poll() %d tick(s) of its generation ago was read after that generation's "
                             "buffers were rewritten (recycle_dicts contract: keep references for one tick at most, copy what "
                             "must live longer, or pass recycle_dicts=False)" % (owner.epoch - epoch))


class _EpochArray(np.ndarray):
    """recycle_dicts="checked": a view of a recycled buffer that remembers the epoch of its generation and refuses to be read
    once the generation has moved on (indexing, ufuncs, numpy functions, copies, conversions; np.asarray() of it bypasses every
    hook and reads the poison instead)"""
    _ce_owner, _ce_epoch = None, 0

    def __array_finalize__(self, obj):
        if obj is not None:
            self._ce_owner, self._ce_epoch = getattr(obj, "_ce_owner", None), getattr(obj, "_ce_epoch", 0)

    def _ce_plain(self):
        _stale(self._ce_owner, self._ce_epoch)
        return self.view(np.ndarray)

    def __getitem__(self, k):
        return self._ce_plain()[k]

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        ins = tuple(x._ce_plain() if isinstance(x, _EpochArray) else x for x in inputs)
        if "out" in kw:
            kw["out"] = tuple(x._ce_plain() if isinstance(x, _EpochArray) else x for x in kw["out"])
        return getattr(ufunc, method)(*ins, **kw)

    def __array_function__(self, func, types, args, kwargs):
        def plain(x):
            if isinstance(x, _EpochArray):
                return x._ce_plain()
            if isinstance(x, (list, tuple)):
                return type(x)(plain(y) for y in x)
            return x
        return func(*plain(args), **{k: plain(v) for k, v in kwargs.items()})

    def copy(self, *a, **kw):
        return self._ce_plain().copy(*a, **kw)

    def astype(self, *a, **kw):
        return self._ce_plain().astype(*a, **kw)

    def tolist(self):
        return self._ce_plain().tolist()

    def __iter__(self):
        return iter(self._ce_plain())

    def __repr__(self):
        return repr(self._ce_plain())

    __str__ = __repr__

    # copies and pickles are plain arrays of the values (never the generation the stamp points at)
    def __deepcopy__(self, memo):
        return self._ce_plain().copy()

    __copy__ = lambda self: self._ce_plain().copy()  # noqa: E731

    def __reduce__(self):
        a = self._ce_plain()
        return (np.array, (a.tolist(), a.dtype.str))  # (values AND dtype: tolist() alone would come back float64 / int64)


def _stamp(arr, owner, epoch=None):
    """`epoch` = the generation's epoch at the poll() the array belongs to (None: now) — a wrapper built LATER from a mapping that
    poll() returned earlier must carry the epoch of that poll, not of the access"""
    v = arr.view(_EpochArray)
    v._ce_owner, v._ce_epoch = owner, owner.epoch if epoch is None else epoch
    return v


class _EpochDict(dict):
    """recycle_dicts="checked": an info / observation dictionary stamped like _EpochArray (reads raise once stale)"""
    __slots__ = ("_ce_owner", "_ce_epoch")

    def __init__(self, owner, items, epoch=None):
        dict.__init__(self, items)
        self._ce_owner, self._ce_epoch = owner, owner.epoch if epoch is None else epoch

    def _ok(self):
        _stale(self._ce_owner, self._ce_epoch)

    def __getitem__(self, k):
        self._ok()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._ok()
        return dict.get(self, k, d)

    def items(self):
        self._ok()
        return dict.items(self)

    def values(self):
        self._ok()
        return dict.values(self)

    def keys(self):
        self._ok()
        return dict.keys(self)

    def __iter__(self):
        self._ok()
        return dict.__iter__(self)

    def __contains__(self, k):
        self._ok()
        return dict.__contains__(self, k)

    def __eq__(self, other):
        self._ok()
        return dict.__eq__(self, other)

    __hash__ = None

    # copies and pickles are plain dictionaries of (copies of) the values
    def __deepcopy__(self, memo):
        import copy
        self._ok()
        return {k: copy.deepcopy(v, memo) for k, v in dict.items(self)}

    def __copy__(self):
        self._ok()
        return dict(dict.items(self))

    def __reduce__(self):
        self._ok()
        return (dict, (list(dict.items(self)),))


def _pydict():
    """the C loops of the recycled dict protocol (built by contracts_amd.build next to the engine library)"""
    from . import _ce_pydict
    return _ce_pydict


class _DictGeneration:
    """one generation of a tick's results on the host: page-locked snapshot buffers, the float64 arrays the dictionaries
    hand out views of, and the dictionary trees themselves (built once, refreshed in place every second tick)"""

    def __init__(self, venv, build_trees=True):
        eng, E, n, keys = venv.engine, venv.num_envs, venv.num_agents, venv._keys
        b = eng.b
        F = b.num_features
        self.epoch = 0  # recycle_dicts="checked": bumped every time this generation's buffers are about to be rewritten
        self.grid = venv.kind in _GRID
        self.float_rewards = venv._float_rewards
        self.rew = eng.host_alloc((E, n), np.float64 if self.float_rewards else np.int32)
        self.done = eng.host_alloc((E,), np.uint8)
        self.info = eng.host_alloc((E, n, 2), np.uint8)
        self.feat_i16 = eng.host_alloc((E, n, F), np.int16)
        self.feat_f64 = np.zeros((E, n, F), np.float64)
        self.err = eng.host_alloc((E,), np.uint32)
        self.theta = eng.host_alloc((E,), np.float64) if venv.contract else None
        self.cobs = np.zeros((E, 2), np.float64)    # [theta, 0]: the wrapper's 'contract' observation (two_stage_train.py:104-117)
        self.cparam = np.zeros((E, 1), np.float64)  # infos[k]['contract_param']
        # shadows: what the dictionaries currently say
        self.rew_shadow = np.zeros_like(self.rew)
        self.done_shadow = np.zeros((E,), np.uint8)
        self.info_shadow = np.zeros((E, n, 2), np.uint8)
        second = _SECOND_INFO[venv.kind]
        contract = bool(venv.contract)
        zero = 0.0 if self.float_rewards else 0
        feat_f64, cobs, cparam = self.feat_f64, self.cobs, self.cparam
        if self.grid:
            self.obs_u8 = eng.host_alloc((E, b.obs_env_stride), np.uint8)
            self.obs_f64 = np.empty((E, n, 15, 15, 3), np.float64)
            self.obs_vec = None
        else:
            # feature-vector kinds: the observation IS the feature row, with the wrapper's [theta, 0] appended under a contract
            # (two_stage_train.py:113-117) — a second array then, the infos' feature_obs rows stay F long
            self.obs_u8 = self.obs_f64 = None
            self.obs_vec = np.zeros((E, n, F + 2), np.float64) if contract else feat_f64
        self.key0 = None if venv.kind == "cleanup_features" else "eaten_apples"  # CleanupFeatures' infos carry the second counter only
        self.obs, self.rewards, self.dones, self.infos = {}, {}, {}, {}
        self.agent_infos = []
        self.second = second
        if not build_trees:  # "checked": stamped per-tick wrappers instead of persistent trees (BatchedBaseEnv._checked_maps)
            return
        for e in range(E):
            fe = feat_f64[e]
            c, cp = cobs[e], cparam[e]
            if self.grid:
                img = self.obs_f64[e]
                if contract:
                    self.obs[e] = {k: {"image": img[i], "contract": c} for i, k in enumerate(keys)}
                else:
                    self.obs[e] = {k: {"image": img[i]} for i, k in enumerate(keys)}
            else:
                ov = self.obs_vec[e]
                self.obs[e] = {k: ov[i] for i, k in enumerate(keys)}
            if self.key0 is None:
                inf = {k: ({second: 0, "contract_param": cp} if contract else {second: 0}) for k in keys}
            elif contract:
                inf = {k: {second: 0, "eaten_apples": 0, "feature_obs": fe[i], "contract_param": cp} for i, k in enumerate(keys)}
            else:
                inf = {k: {second: 0, "eaten_apples": 0, "feature_obs": fe[i]} for i, k in enumerate(keys)}
            self.infos[e] = inf
            self.agent_infos.extend(inf[k] for k in keys)
            self.rewards[e] = dict.fromkeys(keys, zero)
            self.dones[e] = {"__all__": False, "a0": False, "a1": False}  # the reference's dones dict (cleanup_new.py:242)
        self.reward_list = [self.rewards[e] for e in range(E)]
        self.done_list = [self.dones[e] for e in range(E)]

    def poison(self):
        """"checked": what a holder of this generation's previous hand-out would read from now on is NaN, not plausible data"""
        self.epoch += 1
        for a in (self.obs_f64, self.feat_f64, self.obs_vec, self.cobs, self.cparam):
            if a is not None:
                a.fill(np.nan)


class BatchedBaseEnv(_RLlibBaseEnv):
    def __init__(self, kind, num_envs, num_agents, contract=None, seed0=73907, convolutional=True, batch_done_resets=True,
                 recycle_dicts="auto", **engine_kwargs):
        self.kind, self.num_envs, self.num_agents = kind, int(num_envs), int(num_agents)
        self.contract = contract
        self.convolutional = convolutional
        self.batch_done_resets = bool(batch_done_resets)
        engine_kwargs.setdefault("auto_reset", False)  # RLlib resets through try_reset
        self.engine = BatchedEnv(kind, num_envs, num_agents, contract=contract, **engine_kwargs)
        self._keys = ["a%d" % i for i in range(self.num_agents)]
        self._float_rewards = bool(contract) or bool(self.engine.cfg.flags & _lib.FLAG_INEQUITY) or kind == "selfdrive"
        self.engine.seed(seed0=seed0)
        self.engine.reset()
        self._fresh = list(range(self.num_envs))  # envs whose next poll returns a reset observation
        self._pending = None                       # env ids stepped by the last send_actions
        self._done_ids = set()                     # envs that reported done and have not been reset yet
        self._episode_over = set()                 # envs whose last step ended an episode (metrics: the final rows)
        self._reset_obs = {}                       # env_id -> reset observation (or the lazy map of its reset batch)
        self._acted = None                         # selfdrive: [E, n] which agents acted in the last step
        # recycled dict protocol (grid kinds): two generations of dictionary trees over page-locked snapshots
        # (selfdrive's dictionaries change their key sets with the acting cars: always rebuilt)
        if recycle_dicts not in ("auto", "checked", True, False):
            raise ValueError("recycle_dicts must be 'auto', 'checked', True or False, not %r" % (recycle_dicts,))
        # "auto" (see the module docstring): Dict observation spaces are copied by RLlib's preprocessor, so the grid kinds recycle
        # everything; the Box-space feature kinds keep the recycled machinery (C action parser, asynchronous copies, refreshed
        # reward / done / info trees) but hand out FRESH observation / feature_obs rows every tick — the only objects RLlib's
        # NoPreprocessor keeps by reference (dropping the whole fast path cost ~290 ms instead of ~8 ms per tick)
        self._fresh_obs = recycle_dicts == "auto" and kind in ("harvest_features", "cleanup_features")
        if recycle_dicts == "auto":
            recycle_dicts = kind in _GRID or self._fresh_obs
        self._checked = recycle_dicts == "checked" and kind != "selfdrive"
        self._recycle = bool(recycle_dicts) and kind != "selfdrive"
        self.recycle_dicts = "checked" if self._checked else ("fresh_obs" if self._fresh_obs else self._recycle)
        self._gens, self._gen = [None, None], 0
        self._act_planes, self._act_turn = None, 0
        self._keys_t = tuple(self._keys)
        self._env_keys = list(range(self.num_envs))
        self._unchecked_steps = False  # steps issued without a fault check (the fast path checks at poll)
        self._threads = None
        self.tick_timing = {}  # where the last recycled tick's host time went (ms), for bench.py's boundary section

    # ---- snapshots: one device -> host copy per field -------------------------------------------------------------
    def _obs_fields(self, env_begin=0, env_count=None):
        eng = self.engine
        theta = eng.download("theta", env_begin, env_count) if self.contract else None
        snap = {"theta": theta, "base": env_begin}
        if theta is not None:  # per-env rows handed out as views: no small-array construction per env / agent
            snap["contract_obs"] = np.stack([theta, np.zeros_like(theta)], axis=1)  # [theta, 0] (two_stage_train.py:104-117)
            snap["contract_param"] = theta.reshape(-1, 1)
        if self.kind in _GRID:
            snap["obs"] = _ImageChunks(eng.download("obs", env_begin, env_count))  # uint8 [cnt, n, 15, 15, 3]
        elif self.kind == "selfdrive":
            snap["obs_f64"] = eng.download("obs_f64", env_begin, env_count)
        else:
            snap["features"] = eng.download("features", env_begin, env_count)
        return snap

    def _obs_of(self, snap, e, acting=None):
        """the observation dict of env e from a snapshot — same containers as the single-env adapters"""
        j = e - snap["base"]
        theta = snap["theta"]
        if self.kind in _GRID:
            img = snap["obs"].env(j)  # float64 [n, 15, 15, 3]
            if self.contract:
                c = snap["contract_obs"][j]
                return {k: {"image": im, "contract": c} for k, im in zip(self._keys, img)}
            return {k: {"image": im} for k, im in zip(self._keys, img)}
        if self.kind == "selfdrive":
            n = self.num_agents
            rows = snap["obs_f64"][j]
            width = 2 * n + 7 if self.contract else 2 * n + 5  # the wrapper appends [theta, 0] (two_stage_train.py:113-117)
            who = range(n) if acting is None else np.nonzero(acting)[0]
            return {self._keys[i]: rows[i, :width].copy() for i in who}
        f = self._features_f64(snap)[j]
        if self.contract:
            tail = snap["contract_obs"][j]
            return {k: np.concatenate((f[i], tail)) for i, k in enumerate(self._keys)}
        return {k: f[i] for i, k in enumerate(self._keys)}

    @staticmethod
    def _features_f64(snap):
        """the tick's int16 feature rows as float64 (feature_obs entries are floats in the reference), converted once"""
        f = snap.get("features_f64")
        if f is None:
            f = snap["features_f64"] = snap["features"].astype(np.float64)
        return f

    def _infos_of(self, snap, e):
        info = snap["info"][e]
        if self.kind == "selfdrive":
            acting = np.nonzero(self._acted[e])[0]
            rank, dtf = (float(x) for x in snap["sd_info"][e])
            out = {}
            for pos, i in enumerate(acting):  # ambulance stats / is_crashed ride on the first acting key (…accelerate.py:183-189)
                first = pos == 0
                out[self._keys[i]] = {"just_passed": bool(info[i, 0]), "is_crashed": int(info[i, 1]) if first else 0,
                                      "ambulance_rank": int(rank) if first else 0.0,
                                      "ambulance_dist_to_front": dtf if first else 0.0}
                if self.contract:
                    out[self._keys[i]]["contract_param"] = snap["contract_param"][e]
            return out
        second = _SECOND_INFO[self.kind]
        inf = info.tolist()  # python ints in one go
        cp = snap["contract_param"][e] if self.contract else None
        if self.kind == "cleanup_features":
            return {k: ({second: inf[i][1], "contract_param": cp} if self.contract else {second: inf[i][1]})
                    for i, k in enumerate(self._keys)}
        feats = self._features_f64(snap)[e]
        if self.contract:
            return {k: {second: inf[i][1], "eaten_apples": inf[i][0], "feature_obs": feats[i], "contract_param": cp}
                    for i, k in enumerate(self._keys)}
        return {k: {second: inf[i][1], "eaten_apples": inf[i][0], "feature_obs": feats[i]} for i, k in enumerate(self._keys)}

    # ---- BaseEnv protocol -----------------------------------------------------------------------------------------
    def poll(self):
        """-> (obs, rewards, dones, infos, off_policy_actions), each a mapping {env_id: {agent_id: value}}"""
        if self._pending is None:  # nothing stepped yet: reset observations only
            ids, self._fresh = self._fresh, []
            snap = self._obs_fields() if ids else None
            zero_r = {k: 0.0 for k in self._keys}
            return (_LazyEnvMap(ids, lambda e: self._obs_of(snap, e)), _LazyEnvMap(ids, lambda e: dict(zero_r)),
                    _LazyEnvMap(ids, lambda e: {"__all__": False}), _LazyEnvMap(ids, lambda e: {k: {} for k in self._keys}), {})
        if self._recycle:
            with _gc_paused():
                return self._poll_recycled()
        ids, self._pending = self._pending, None
        eng = self.engine
        self._settle_faults()
        snap = self._obs_fields()
        snap["rew"] = eng.download("reward") if self._float_rewards else eng.download("base_reward")
        snap["done"] = eng.download("done")
        snap["info"] = eng.download("info")
        if self.kind == "selfdrive":
            snap["done_agents"] = eng.download("done_agents")
            snap["sd_info"] = eng.download("sd_info")
        elif self.kind in _GRID:
            snap["features"] = eng.download("features")
        self._done_ids = {int(e) for e in np.nonzero(snap["done"])[0]}
        self._episode_over = set(self._done_ids)
        self._reset_obs = {}
        sd = self.kind == "selfdrive"

        def rewards(e):
            r = snap["rew"][e]
            if sd:
                return {self._keys[i]: float(r[i]) for i in np.nonzero(self._acted[e])[0]}
            return dict(zip(self._keys, r.tolist()))  # python floats (contract / inequity) or ints

        def dones(e):
            d = bool(snap["done"][e])
            if sd:
                out = {k: bool(snap["done_agents"][e, i]) for i, k in enumerate(self._keys)}
                out["__all__"] = d
                return out
            return {"__all__": d, "a0": d, "a1": d}  # the reference's dones dict (cleanup_new.py:242)

        return (_LazyEnvMap(ids, lambda e: self._obs_of(snap, e, self._acted[e] if sd else None)), _LazyEnvMap(ids, rewards),
                _LazyEnvMap(ids, dones), _LazyEnvMap(ids, lambda e: self._infos_of(snap, e)), {})

    def send_actions(self, action_dict):
        """{env_id: {agent_id: action}} for every env (one kernel launch steps them all).  Selfdrive: an env's dict
        holds the agents that act (RLlib stops sending actions for agents that are done)."""
        if len(action_dict) != self.num_envs or any(e not in action_dict for e in range(self.num_envs)):
            raise KeyError("send_actions needs actions for all %d sub-envs in one call" % self.num_envs)
        E, n, keys = self.num_envs, self.num_agents, self._keys
        if self._recycle and type(action_dict) is dict:
            # the walk over the E x n action entries in C, into a page-locked plane that the step's stream copies from
            import time
            t0 = time.perf_counter()
            plane = self._action_plane()
            _pydict().parse_actions(action_dict, self._env_keys, self._keys_t, plane)
            self.engine.step_host_async(plane)
            self.tick_timing["send_actions_ms"] = (time.perf_counter() - t0) * 1e3
            self._unchecked_steps = True
            self._episode_over = set()
            self._pending = self._env_keys
            return
        if self.kind == "selfdrive":
            a = np.zeros((E, n), np.float32)
            act = np.zeros((E, n), np.uint8)
            for e in range(E):
                for k, v in action_dict[e].items():
                    i = int(k[1:])
                    a[e, i] = np.float32(np.asarray(v).reshape(-1)[0])
                    act[e, i] = 1
            self._acted = act
            self.engine.step(a, act)
        else:
            a = np.fromiter((action_dict[e][k] for e in range(E) for k in keys), np.int64, E * n).astype(np.uint8)
            self.engine.step(a.reshape(E, n))
        self.engine.check_faults()
        self._episode_over = set()  # every env has stepped again: env_metrics() serves the running episode until poll()
        self._pending = list(range(E))

    # ---- recycled dict protocol (see the module docstring) ---------------------------------------------------------
    def _action_plane(self):
        """page-locked uint8 [E, n] planes, used in turn: the previous tick's asynchronous upload may still be reading its own"""
        if self._act_planes is None:
            self._act_planes = [self.engine.host_alloc((self.num_envs, self.num_agents), np.uint8) for _ in range(3)]
        self._act_turn = (self._act_turn + 1) % len(self._act_planes)
        return self._act_planes[self._act_turn]

    def _settle_faults(self):
        if self._unchecked_steps:
            self._unchecked_steps = False
            self.engine.check_faults()

    def _host_threads(self):
        if self._threads is None:
            import os
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 2
            self._threads = max(1, min(32 if cores >= 128 else 16, cores - 1))
        return self._threads

    def _poll_recycled(self):
        import time
        eng, pd = self.engine, _pydict()
        tm, t0 = self.tick_timing, time.perf_counter()
        self._pending = None
        g = self._gen = self._gen ^ 1
        G = self._gens[g]
        if G is None:
            G = self._gens[g] = _DictGeneration(self, build_trees=not self._checked)
        if self._checked:
            G.poison()
        # small fields first: one asynchronous copy each behind the step on its stream, ONE synchronize for all of them
        eng.download_async("reward" if G.float_rewards else "base_reward", G.rew)
        eng.download_async("done", G.done)
        eng.download_async("info", G.info)
        eng.download_async("features", G.feat_i16)
        eng.download_async("error_flags", G.err)
        if G.theta is not None:
            eng.download_async("theta", G.theta)
        eng.synchronize()
        t1 = time.perf_counter()
        # the observations (most of the bytes) travel and are converted while this thread refreshes the dictionaries: one
        # library call (no GIL hand-offs with the refresh loops) that copies into the page-locked block in parts and converts
        # part i (value / 255 -> float64 across the host cores) while the later parts are still on the wire
        T = self._host_threads()

        def finish_obs():
            ta = time.perf_counter()
            if G.grid:
                eng.download_obs_f64(G.obs_u8, G.obs_f64, T, parts=4 if self.num_envs >= 1024 else 1)
            eng.i16_to_f64(G.feat_i16, G.feat_f64, T)  # feature rows -> the float64 arrays infos[..]['feature_obs'] are views of
            if G.obs_vec is not None and G.obs_vec is not G.feat_f64:  # feature kinds under a contract: row = features + [theta, 0]
                F = G.feat_f64.shape[2]
                G.obs_vec[:, :, :F] = G.feat_f64
                G.obs_vec[:, :, F] = G.theta[:, None]
                G.obs_vec[:, :, F + 1] = 0.0  # (constant, but "checked" poisons the whole block before every rewrite)
            tm["obs_copy_and_convert_ms"] = (time.perf_counter() - ta) * 1e3

        job = _pool().submit(finish_obs)
        try:
            self._unchecked_steps = False
            if G.err.any():
                bad = np.nonzero(G.err)[0]
                raise _lib.EngineError("env faults: %s" % {int(i): int(G.err[i]) for i in bad[:8]})
            if G.theta is not None:
                G.cobs[:, 0] = G.theta
                G.cobs[:, 1] = 0.0  # (constant, but "checked" poisons the block before every rewrite)
                G.cparam[:, 0] = G.theta
            if self._checked:
                pass  # the stamped wrappers are built per env on access (_checked_maps)
            elif G.float_rewards:
                pd.refresh_floats(G.reward_list, self._keys_t, G.rew, G.rew_shadow)
            else:
                pd.refresh_ints(G.reward_list, self._keys_t, G.rew, G.rew_shadow)
            if not self._checked:
                pd.refresh_infos(G.agent_infos, G.key0, G.second, G.info, G.info_shadow)
                pd.refresh_dones(G.done_list, ("__all__", "a0", "a1"), G.done, G.done_shadow)
            self._done_ids = {int(e) for e in np.nonzero(G.done)[0]}
            self._episode_over = set(self._done_ids)
            self._reset_obs = {}
            t2 = time.perf_counter()
        finally:
            job.result()
        if self._fresh_obs:  # Box-space kinds under "auto": this tick's rows live in arrays of their own (nothing rewrites them)
            fresh = G.obs_vec.copy()
            rows = list(fresh.reshape(self.num_envs * self.num_agents, fresh.shape[2]))
            n, keys = self.num_agents, self._keys
            fresh_obs = {e: dict(zip(keys, rows[e * n:(e + 1) * n])) for e in self._env_keys}  # (per-env dictionaries of their own too)
            if G.key0 is not None:  # infos[..]['feature_obs'] (kept by SampleBatch.INFOS): the feature part of the same rows
                F = G.feat_f64.shape[2]
                pd.assign_key(G.agent_infos, "feature_obs", rows if fresh.shape[2] == F else [r[:F] for r in rows])
            # ... and info dictionaries of their own too (SampleBatch.INFOS keeps the dictionary objects): copies of the refreshed ones
            flat = [d.copy() for d in G.agent_infos]
            fresh_infos = {e: dict(zip(keys, flat[e * n:(e + 1) * n])) for e in self._env_keys}
        t3 = time.perf_counter()
        tm["step_and_small_fields_ms"], tm["refresh_dicts_ms"], tm["wait_for_obs_ms"] = (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
        tm["poll_ms"] = (t3 - t0) * 1e3
        if self._checked:
            return self._checked_maps(G)
        if self._fresh_obs:
            return fresh_obs, G.rewards, G.dones, fresh_infos, {}
        return G.obs, G.rewards, G.dones, G.infos, {}

    def _checked_maps(self, G):
        """recycle_dicts="checked": this tick's results as per-env dictionaries built on access from generation G's buffers, every
        array an _EpochArray view and every observation / info dictionary an _EpochDict stamped with G's current epoch — equal,
        value for value, to what the recycled trees hold, but a read after G has moved on raises StaleDictError"""
        keys, contract, second, key0 = self._keys, bool(self.contract), G.second, G.key0
        ids = self._env_keys
        rew, info, done = G.rew.tolist(), G.info.tolist(), G.done.tolist()
        epoch = G.epoch  # of THIS poll: a sampler that keeps the top-level mappings and first indexes them two ticks later gets
        #                  StaleDictError from the builder, never wrappers stamped with the newer epoch over the newer data

        def obs(e):
            _stale(G, epoch)
            if G.grid:
                img = G.obs_f64[e]
                if contract:
                    c = _stamp(G.cobs[e], G, epoch)
                    return _EpochDict(G, ((k, _EpochDict(G, (("image", _stamp(img[i], G, epoch)), ("contract", c)), epoch)) for i, k in enumerate(keys)), epoch)
                return _EpochDict(G, ((k, _EpochDict(G, (("image", _stamp(img[i], G, epoch)),), epoch)) for i, k in enumerate(keys)), epoch)
            ov = G.obs_vec[e]
            return _EpochDict(G, ((k, _stamp(ov[i], G, epoch)) for i, k in enumerate(keys)), epoch)

        def infos(e):
            _stale(G, epoch)
            fe = G.feat_f64[e]
            out = {}
            for i, k in enumerate(keys):
                it = [(second, info[e][i][1])]
                if key0 is not None:
                    it += [("eaten_apples", info[e][i][0]), ("feature_obs", _stamp(fe[i], G, epoch))]
                if contract:
                    it.append(("contract_param", _stamp(G.cparam[e], G, epoch)))
                out[k] = _EpochDict(G, it, epoch)
            return out

        def dones(e):
            d = bool(done[e])
            return {"__all__": d, "a0": d, "a1": d}

        return (_LazyEnvMap(ids, obs), _LazyEnvMap(ids, lambda e: dict(zip(keys, rew[e]))), _LazyEnvMap(ids, dones),
                _LazyEnvMap(ids, infos), {})

    def send_actions_array(self, actions, active=None):
        """the same tick from a dense [E, n] array (uint8 action ids / float32 accelerations): no per-env containers"""
        if self.kind == "selfdrive":
            self._acted = (1 - self.engine.download("done_agents")) if active is None else np.asarray(active, np.uint8)
            self.engine.step(actions, self._acted)
            self.engine.check_faults()
        else:
            # staged through a page-locked plane and issued asynchronously: no host synchronization in the tick (a fault —
            # an action id outside the table — is reported by the next poll() / check_faults())
            plane = self._action_plane()
            np.copyto(plane, np.asarray(actions).reshape(plane.shape), casting="unsafe")
            self.engine.step_host_async(plane)
            self._unchecked_steps = True
        self._episode_over = set()
        self._pending = self._env_keys

    def poll_tensors(self):
        """zero-copy torch views of the engine's output buffers (observations stay in HBM) — what a GPU-resident sampler
        reads instead of poll(); valid until the next send_actions / try_reset.  Issues no synchronization: work queued on the
        caller's stream after this (torch ops on the default stream) is ordered behind the step; `check_faults()` on demand"""
        self._pending = None
        return self.engine.torch_tensors()

    def check_faults(self):
        """raise if any sub-env reported a fault (bad action id, ...) since the last check; synchronizes the device"""
        self._unchecked_steps = False
        self.engine.check_faults()

    def try_reset(self, env_id=None):
        """reset observation(s) {env_id: obs}.  The first call after a tick resets every env that reported done (plus the
        one asked for) in one masked launch; later calls for envs of that batch are answered from it."""
        if env_id is not None and env_id in self._reset_obs:
            return {env_id: self._take_reset(env_id)}
        E = self.num_envs
        ids = set(range(E)) if env_id is None else ((self._done_ids | {env_id}) if self.batch_done_resets else {env_id})
        mask = np.zeros((E,), np.uint8)
        mask[list(ids)] = 1
        self.engine.reset(mask=mask)
        self._done_ids -= ids
        if len(ids) <= 4:  # a few scattered envs: per-env slices instead of the whole batch
            for e in ids:
                self._reset_obs[e] = self._obs_of(self._obs_fields(e, 1), e)
        else:
            snap = self._obs_fields()
            lazy = _LazyEnvMap(sorted(ids), lambda e: self._obs_of(snap, e))
            for e in ids:  # entries of an earlier batch that were not collected yet stay valid
                self._reset_obs[e] = lazy
        if env_id is None:
            return {e: self._take_reset(e) for e in sorted(ids)}
        return {env_id: self._take_reset(env_id)}

    def _take_reset(self, e):
        v = self._reset_obs.pop(e)
        return v[e] if isinstance(v, _LazyEnvMap) else v

    # ---- episode metrics: what the reference's MetricsCallback (utils/logger_utils.py:126-150) reads per episode ----
    def env_metrics(self, env_id):
        """the `metrics` dictionary of sub-env `env_id`, with the single-env adapters' keys: the finished episode's
        (equality / sustainability included) from the poll() that reported the done until the next send_actions — a
        reset in between does not clear them — the running episode's otherwise"""
        eng, e = self.engine, int(env_id)
        if not 0 <= e < self.num_envs:
            raise IndexError(env_id)
        final = e in self._episode_over
        mi = eng.download("final_int_metrics" if final else "int_metrics", e, 1)[0]
        mf = eng.download("final_f64_metrics" if final else "f64_metrics", e, 1)[0]
        return episode_metrics(self.kind, self.num_agents, mi, mf, final and self.kind != "selfdrive",
                               contract=bool(self.contract), inequity=bool(eng.cfg.flags & _lib.FLAG_INEQUITY))

    def get_sub_environments(self, as_dict=False):
        """per-env views for callbacks (`base_env.get_sub_environments()[env_index].metrics`); `.base_env` is the view
        itself, so the reference's wrapper.base_env.metrics chain resolves too"""
        views = _SubEnvs(self)
        return {e: views[e] for e in range(self.num_envs)} if as_dict else views

    @staticmethod
    def freeze_gc():
        """Opt-in, process-wide: `gc.collect(); gc.freeze()` — every object alive now (torch, numpy, this env's recycled
        dictionary trees: ~10^6 containers at E = 16 384) moves to the collector's permanent generation and is no longer walked
        by full collections.  A sampler that allocates per tick (action dictionaries, batches) otherwise triggers one full
        collection every few ticks, and each walks all of that: ≈ 85 ms against a 14 ms tick (bench.py: boundary.dict_protocol
        value_incl_action_dicts vs ..._gc_frozen).  Call it once the loop is warm (after the second poll(): both generations
        of trees exist then); `gc.unfreeze()` undoes it.  Reference counting is unaffected; only cycles among the frozen
        objects are never reclaimed."""
        import gc
        gc.collect()
        gc.freeze()

    def stop(self):
        self.engine.close()


class BatchedJointBaseEnv(BatchedBaseEnv):
    """`JointEnv` semantics (reference environments/two_stage_train.py:476-617 — the joint baseline of
    experiment_configs/cleanup-joint-2agents.json, built at utils/ray_config_utils.py:160-183) for E sub-envs from ONE engine
    handle: RLlib's `to_base_env(num_envs=E)` of a JointEnv over a pixel grid env is one kernel launch per sampler tick
    instead of E Python objects.  The centralised agent 'a0'
      * sends one MultiDiscrete action per base agent (`global_action_space`): the [E, n] ids go to the device as one plane;
      * receives the SUM of the agents' rewards (Python's left-to-right `sum`) and of every info entry (`feature_obs` rows
        add elementwise), `dones = {'a0': d, '__all__': d}`;
      * observes `{'image': ...}`: under `mode="global"` the whole colour map with the agents painted (`MapEnv.global_view`
        / 255 — produced on the device by ce_global_view, [E, H, W, 3] uint8, one launch and one copy per tick), under
        `mode="concatenated"` the n egocentric views stacked on the channel axis ([15, 15, 3 n]; the engine's `obs` buffer
        read as [E, 15, 15, n, 3], i.e. a strided view of what the step wrote).
    The joint baseline has one agent per env — E dictionaries of one entry each — so a tick's four mappings are plain dicts
    built in four passes over the batch, new every tick; the stacked views are transposed on the device, the float64 images
    converted chunk by chunk on the worker threads into one of two recycled host blocks (`recycle_images`, see __init__: an
    image array is valid through the next tick; pass False where a consumer keeps observations without copying them).
    `poll_tensors()` additionally carries `global_view` (device uint8) under mode="global"."""

    def __init__(self, kind, num_envs, num_agents, mode="concatenated", seed0=73907, recycle_images="auto", **engine_kwargs):
        if kind not in _GRID:
            raise ValueError("BatchedJointBaseEnv serves the pixel grid kinds (cleanup_new / harvest_new), not %r" % (kind,))
        if mode not in ("global", "concatenated"):
            raise ValueError("mode must be 'global' or 'concatenated', not %r" % (mode,))
        if recycle_images not in ("auto", True, False):
            raise ValueError("recycle_images must be 'auto', True or False, not %r" % (recycle_images,))
        engine_kwargs.pop("contract", None)  # the joint baseline runs the bare base env (utils/ray_config_utils.py:159-176)
        BatchedBaseEnv.__init__(self, kind, num_envs, num_agents, contract=None, seed0=seed0, recycle_dicts=False, **engine_kwargs)
        self.mode = mode
        self._gv = None  # device buffer of the global views (torch owns it: plumbing)
        # The float64 images of a stepped tick are converted into one of TWO host blocks used in turn — BatchedBaseEnv's
        # recycling contract (what poll() returned at tick t is intact through t + 1 and rewritten in place by t + 2), for
        # the image arrays only; the dictionaries around them are new every tick.  'auto' = on: the observation space is a
        # Dict, which RLlib flattens (copies) as it receives it.  Fresh pages for 16 384 stacked views are 0.7 GB per tick:
        # faulting them in and handing them back is three quarters of such a tick.  False: new arrays every tick.
        self._recycle_images = recycle_images in ("auto", True)
        self._img_ring, self._img_turn = [None, None], 0

    # ---- observations ----------------------------------------------------------------------------------------------
    def global_view_device(self, env_begin=0, env_count=None):
        """uint8 [E, H, W, 3] on the device: MapEnv.global_view of every sub-env; the rows of the env slice are refreshed by one
        ce_global_view launch"""
        import torch
        b = self.engine.b
        if self._gv is None:
            self._gv = torch.zeros((self.num_envs, b.grid_h, b.grid_w, 3), dtype=torch.uint8, device="cuda:%d" % self.engine.cfg.device)
        cnt = self.num_envs - env_begin if env_count is None else env_count
        self.engine.global_view(self._gv[env_begin:].data_ptr(), env_begin, cnt)
        return self._gv

    def _obs_fields(self, env_begin=0, env_count=None, ring=False):
        snap = {"base": env_begin, "theta": None}
        cnt = self.num_envs - env_begin if env_count is None else env_count
        if self.mode == "global":
            gv = self.global_view_device(env_begin, cnt)
            self.engine.synchronize()
            u8 = gv[env_begin:env_begin + cnt].cpu().numpy()
        else:
            # [cnt, n, 15, 15, 3] -> [cnt, 15, 15, n, 3] on the device (a strided read of what the step wrote, one dense copy
            # out): the same transpose of 3-byte pixels on the host costs more than the tick's every other part together
            self.engine.synchronize()
            o = self.engine.torch_tensors()["obs"][env_begin:env_begin + cnt]
            u8 = o.permute(0, 2, 3, 1, 4).contiguous().cpu().numpy().reshape(cnt, 15, 15, 3 * self.num_agents)
        out = None
        if ring and self._recycle_images and cnt == self.num_envs:  # (reset observations never take a turn of the ring)
            self._img_turn ^= 1
            out = self._img_ring[self._img_turn]
            if out is None:
                out = self._img_ring[self._img_turn] = np.empty(u8.shape, np.float64)
        snap["image"] = _ImageChunks(u8, out)  # value / 255 -> float64 (cleanup_new.py:300), chunks converted on the worker threads
        return snap

    def _obs_of(self, snap, e, acting=None):
        return {"a0": {"image": snap["image"].env(e - snap["base"])}}

    # ---- BaseEnv protocol ------------------------------------------------------------------------------------------
    def poll(self):
        if self._pending is None:
            ids, self._fresh = self._fresh, []
            snap = self._obs_fields() if ids else None
            return (_LazyEnvMap(ids, lambda e: self._obs_of(snap, e)), _LazyEnvMap(ids, lambda e: {"a0": 0.0}),
                    _LazyEnvMap(ids, lambda e: {"__all__": False}), _LazyEnvMap(ids, lambda e: {"a0": {}}), {})
        with _gc_paused():
            return self._poll_stepped()

    def _poll_stepped(self):
        ids, self._pending = self._pending, None
        eng, n = self.engine, self.num_agents
        self._settle_faults()
        snap = self._obs_fields(ring=True)  # first: the worker threads convert the images while the rest of the tick is put together
        rew = eng.download("reward") if self._float_rewards else eng.download("base_reward")
        total = 0  # Python's sum(): 0 + r_a0 + r_a1 + ... left to right (ints stay ints, inequity-averse floats add in that order)
        for a in range(n):
            total = total + rew[:, a]
        done = eng.download("done")
        info = eng.download("info").astype(np.int64).sum(axis=1)  # [E, 2]: eaten_apples, cleaned_squares | eaten_close_apples
        feats = eng.download("features").astype(np.float64)
        fsum = 0
        for a in range(n):
            fsum = fsum + feats[:, a]
        self._done_ids = {int(e) for e in np.nonzero(done)[0]}
        self._episode_over = set(self._done_ids)
        self._reset_obs = {}
        second = _SECOND_INFO[self.kind]
        tot, inf, dn = total.tolist(), info.tolist(), done.astype(bool).tolist()
        if list(ids) != list(range(self.num_envs)):  # (never from send_actions / send_actions_array: every env steps every tick)
            return (_LazyEnvMap(ids, lambda e: self._obs_of(snap, e)), _LazyEnvMap(ids, lambda e: {"a0": tot[e]}),
                    _LazyEnvMap(ids, lambda e: {"a0": dn[e], "__all__": dn[e]}),
                    _LazyEnvMap(ids, lambda e: {"a0": {second: inf[e][1], "eaten_apples": inf[e][0], "feature_obs": fsum[e]}}), {})
        # One centralised agent per env and a sampler that walks every env: the four mappings are built in four passes over
        # the batch (a lazily built entry costs a Python call per env and mapping — most of a 16 384-env tick)
        rews = {e: {"a0": t} for e, t in enumerate(tot)}
        dones = {e: {"a0": d, "__all__": d} for e, d in enumerate(dn)}
        infos = {e: {"a0": {second: i[1], "eaten_apples": i[0], "feature_obs": f}} for e, (i, f) in enumerate(zip(inf, fsum))}
        return {e: {"a0": {"image": im}} for e, im in enumerate(snap["image"].rows())}, rews, dones, infos, {}

    def send_actions(self, action_dict):
        """{env_id: {'a0': [action of a0, action of a1, ...]}} for every env: one [E, n] plane, one launch"""
        E, n = self.num_envs, self.num_agents
        if len(action_dict) != E:
            raise KeyError("send_actions needs actions for all %d sub-envs in one call" % E)
        try:
            rows = [action_dict[e]["a0"] for e in range(E)]
        except KeyError:
            raise KeyError("send_actions needs actions for all %d sub-envs in one call" % E) from None
        try:
            a = np.asarray(rows, np.int64)
        except (ValueError, TypeError):  # ragged or nested rows: one by one
            a = None
        if a is None or a.shape != (E, n):
            a = np.empty((E, n), np.int64)
            for e in range(E):
                a[e] = np.asarray(rows[e]).reshape(-1)[:n]
        self.send_actions_array(a)

    def poll_tensors(self):
        t = dict(BatchedBaseEnv.poll_tensors(self))
        if self.mode == "global":
            t["global_view"] = self.global_view_device()
        return t
