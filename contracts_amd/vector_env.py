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
  * `poll_tensors()` / `send_actions_array()` skip Python containers altogether (observations stay in HBM).

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
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) - 1)), thread_name_prefix="ce-obs")
    return _POOL


class _ImageChunks:
    """the tick's observations as float64 (`uint8 / 255`, cleanup_new.py:258 / harvest_new.py:229), converted in chunks
    of _CHUNK envs by background threads while the caller walks the envs; small batches convert on demand"""

    def __init__(self, raw):
        self.raw = raw
        self.jobs = None
        if raw.shape[0] >= 4 * _CHUNK:
            pool = _pool()
            self.jobs = [pool.submit(np.true_divide, raw[c:c + _CHUNK], 255) for c in range(0, raw.shape[0], _CHUNK)]

    def env(self, j):
        if self.jobs is None:
            return self.raw[j] / 255
        return self.jobs[j // _CHUNK].result()[j % _CHUNK]


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


class BatchedBaseEnv(_RLlibBaseEnv):
    def __init__(self, kind, num_envs, num_agents, contract=None, seed0=73907, convolutional=True, batch_done_resets=True,
                 **engine_kwargs):
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
        ids, self._pending = self._pending, None
        eng = self.engine
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

    def send_actions_array(self, actions, active=None):
        """the same tick from a dense [E, n] array (uint8 action ids / float32 accelerations): no per-env containers"""
        if self.kind == "selfdrive":
            self._acted = (1 - self.engine.download("done_agents")) if active is None else np.asarray(active, np.uint8)
            self.engine.step(actions, self._acted)
        else:
            self.engine.step(actions)
        self.engine.check_faults()
        self._episode_over = set()
        self._pending = list(range(self.num_envs))

    def poll_tensors(self):
        """zero-copy torch views of the engine's output buffers (observations stay in HBM) — what a GPU-resident sampler
        reads instead of poll(); valid until the next send_actions / try_reset"""
        self._pending = None
        return self.engine.torch_tensors()

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

    def stop(self):
        self.engine.close()
