"""BatchedBaseEnv — RLlib `BaseEnv`-shaped vector hook over one engine handle.

RLlib samples through `BaseEnv.poll() / send_actions() / try_reset()`; a `MultiAgentEnv` is normally wrapped one
Python object per sub-env (`to_base_env`).  This class serves the same protocol for E sub-envs from ONE engine
handle: `send_actions` is one kernel launch for all E envs, `poll` hands back the per-env / per-agent dictionaries
RLlib expects, `try_reset(env_id)` is a masked `ce_reset`.  Observations, rewards, dones and infos have exactly the
shapes of the single-env adapters (`environments/*.py`), including the contract wrapper's extra observation entry.

When `ray` is importable the class derives from `ray.rllib.env.BaseEnv`, otherwise it is duck-typed; nothing else
in it depends on RLlib.  Each sub-env keeps a private RNG stream seeded `seed0 + env_index_base + i` (the batched
API's convention), not the process-global generator.
"""
import numpy as np

from . import _lib
from .engine import BatchedEnv

try:  # pragma: no cover - RLlib is absent in the build image
    from ray.rllib.env import BaseEnv as _RLlibBaseEnv
except Exception:
    _RLlibBaseEnv = object

_GRID = ("cleanup", "harvest")


class BatchedBaseEnv(_RLlibBaseEnv):
    def __init__(self, kind, num_envs, num_agents, contract=None, seed0=73907, convolutional=True, **engine_kwargs):
        if kind == "selfdrive":
            raise NotImplementedError("selfdrive steps subsets of agents; use the per-env adapter for it")
        self.kind, self.num_envs, self.num_agents = kind, int(num_envs), int(num_agents)
        self.contract = contract
        self.convolutional = convolutional
        engine_kwargs.setdefault("auto_reset", False)  # RLlib resets through try_reset
        self.engine = BatchedEnv(kind, num_envs, num_agents, contract=contract, **engine_kwargs)
        self._keys = ["a%d" % i for i in range(self.num_agents)]
        self.engine.seed(seed0=seed0)
        self.engine.reset()
        self._fresh = set(range(self.num_envs))  # envs whose next poll returns a reset observation
        self._pending = None

    # ---- observation / info builders (same containers as the single-env adapters) ----
    def _obs_all(self):
        eng, n = self.engine, self.num_agents
        theta = eng.download("theta") if self.contract else None
        if self.kind in _GRID:
            img = eng.download("obs") / 255  # uint8/255 -> float64 (cleanup_new.py:258 / harvest_new.py:229)
            out = []
            for e in range(self.num_envs):
                d = {}
                for i, k in enumerate(self._keys):
                    o = {"image": img[e, i]}
                    if self.contract:
                        o["contract"] = np.array([theta[e], 0.0])
                    d[k] = o
                out.append(d)
            return out
        f = eng.download("features").astype(np.float64)
        if self.contract:
            tail = np.stack([theta, np.zeros_like(theta)], axis=1)
            return [{k: np.concatenate((f[e, i], tail[e])) for i, k in enumerate(self._keys)} for e in range(self.num_envs)]
        return [{k: f[e, i] for i, k in enumerate(self._keys)} for e in range(self.num_envs)]

    def _infos_all(self):
        eng = self.engine
        info = eng.download("info")
        feats = eng.download("features").astype(np.float64)
        theta = eng.download("theta") if self.contract else None
        second = {"cleanup": "cleaned_squares", "harvest": "eaten_close_apples", "harvest_features": "eaten_close_apples",
                  "cleanup_features": "cleaned_squares"}[self.kind]
        out = []
        for e in range(self.num_envs):
            d = {}
            for i, k in enumerate(self._keys):
                ent = {second: int(info[e, i, 1])}
                if self.kind != "cleanup_features":
                    ent["eaten_apples"] = int(info[e, i, 0])
                    ent["feature_obs"] = feats[e, i]
                if self.contract:
                    ent["contract_param"] = np.array([theta[e]])
                d[k] = ent
            out.append(d)
        return out

    # ---- BaseEnv protocol ----
    def poll(self):
        """-> (obs, rewards, dones, infos, off_policy_actions), each {env_id: {agent_id: value}}"""
        obs_all = self._obs_all()
        obs, rew, dones, infos = {}, {}, {}, {}
        if self._pending is None:  # nothing stepped yet: reset observations only
            for e in sorted(self._fresh):
                obs[e] = obs_all[e]
                rew[e] = {k: 0.0 for k in self._keys}
                dones[e] = {"__all__": False}
                infos[e] = {k: {} for k in self._keys}
            self._fresh.clear()
            return obs, rew, dones, infos, {}
        stepped = self._pending
        self._pending = None
        eng = self.engine
        use_float = bool(self.contract) or bool(self.engine.cfg.flags & _lib.FLAG_INEQUITY)
        r = eng.download("reward") if use_float else eng.download("base_reward")
        done = eng.download("done")
        infos_all = self._infos_all()
        for e in stepped:
            obs[e] = obs_all[e]
            rew[e] = {k: (float(r[e, i]) if use_float else int(r[e, i])) for i, k in enumerate(self._keys)}
            d = bool(done[e])
            dones[e] = {"__all__": d, "a0": d, "a1": d}
            infos[e] = infos_all[e]
        return obs, rew, dones, infos, {}

    def send_actions(self, action_dict):
        """{env_id: {agent_id: action}} for every env (one kernel launch steps them all)"""
        if set(action_dict.keys()) != set(range(self.num_envs)):
            raise KeyError("send_actions needs actions for all %d sub-envs in one call" % self.num_envs)
        a = np.zeros((self.num_envs, self.num_agents), np.uint8)
        for e, acts in action_dict.items():
            for i, k in enumerate(self._keys):
                a[e, i] = int(acts[k])
        self.engine.step(a)
        self.engine.check_faults()
        self._pending = sorted(action_dict.keys())

    def try_reset(self, env_id=None):
        ids = range(self.num_envs) if env_id is None else [env_id]
        mask = np.zeros((self.num_envs,), np.uint8)
        mask[list(ids)] = 1
        self.engine.reset(mask=mask)
        obs_all = self._obs_all()
        return {e: obs_all[e] for e in ids}

    def get_sub_environments(self, as_dict=False):
        return {} if as_dict else []

    def stop(self):
        self.engine.close()
