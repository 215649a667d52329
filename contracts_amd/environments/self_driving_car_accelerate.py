"""SelfAcceleratingCarEnv — drop-in for environments/self_driving_car_accelerate.py:18, stepped by the
HIP engine (float64 state, one lane per env).  Same constructor kwargs (:19), spaces (:40-47), reset /
step dictionaries (:49-79,151-250).  The action dict may hold a subset of the agents (RLlib stops
sending actions for agents whose done flag is set); obs/rewards/infos are returned for those keys."""
import numpy as np

from .. import spaces
from ..engine import BatchedEnv
from .metrics import episode_metrics
from .map_env import _Base, pull_global_rng, push_global_rng, restore_pending_state
from .vector_hook import VectorHookMixin

ACCEL_LOW_THRESH, ACCEL_HIGH_THRESH = -0.1, 0.1


class SelfAcceleratingCarEnv(VectorHookMixin, _Base):
    def __init__(self, low_bound=-10.0, high_bound=10.0, start_vel=0.2, start_vel_ambulance=0.8, num_agents=2,
                 collision_on=False, rng="global", device=0, **kwargs):
        self.num_agents = num_agents
        self.low_bound, self.high_bound = low_bound, high_bound
        self.start_vel, self.start_vel_ambulance = start_vel, start_vel_ambulance
        self.collision_on = collision_on
        self.metrics = {"transfers": 0}
        self._keys = ["a%d" % i for i in range(num_agents)]
        self._rng_mode, self._device = rng, device
        self._engine = None
        self._contract = (None, None, None, 0.0)
        self.observation_space = spaces.Box(low=low_bound - 20, high=high_bound + 20,
                                            shape=(2 * (num_agents + 1) + 3,), dtype=np.float32)
        self.action_space = spaces.Box(low=ACCEL_LOW_THRESH, high=ACCEL_HIGH_THRESH, shape=(1,), dtype=np.float32)
        self._ensure_engine().construct()  # __init__ uses no RNG

    def _ensure_engine(self):
        if self._engine is None:
            self._engine = BatchedEnv("selfdrive", 1, self.num_agents, collision_on=self.collision_on,
                                      low_bound=self.low_bound, high_bound=self.high_bound, start_vel=self.start_vel,
                                      start_vel_ambulance=self.start_vel_ambulance, device=self._device)
            c, lo, hi, null_prob = self._contract
            if c is not None:
                self._engine.set_contract(c, lo, hi, null_prob)
            if getattr(self, "_external_theta", False):  # a negotiate / combined / solver stage owns theta (set before pickling)
                self._engine.set_flags(external_theta=True)
            restore_pending_state(self, self._engine)
        return self._engine

    def __getstate__(self):
        d = dict(self.__dict__)
        eng = d.pop("_engine", None)
        if eng is not None:
            d["_pending_state"] = {f: eng.download(f, raw=True) for f in ("sd_state", "rng", "theta", "f64_metrics", "int_metrics",
                                                                          "done", "done_agents", "error_flags")}
            from .._lib import CE_ABI_VERSION
            d["_pending_abi"] = CE_ABI_VERSION
        d["_engine"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    _RESULT_FIELDS = ("rng", "error_flags", "obs_f64", "base_reward", "reward", "done", "done_agents", "info", "sd_info", "f64_metrics", "theta")

    def _call(self, fn, *args):
        """one engine call with the process-global generators mirrored around it; the whole result of the call is then
        fetched in ONE copy (BatchedEnv.prefetch) and the download() calls that follow are served from it"""
        eng = self._ensure_engine()
        if self._rng_mode == "global":
            st = push_global_rng(eng, python_random=True)
            fn(*args)
            eng.prefetch(self._RESULT_FIELDS)
            pull_global_rng(eng, st, python_random=True)
        else:
            fn(*args)
            eng.prefetch(self._RESULT_FIELDS)

    def seed(self, seed=None):
        self._vector_seed0 = seed
        if self._rng_mode == "global":
            import random
            np.random.seed(seed)
            random.seed(seed)
        else:
            self._ensure_engine().seed(np.array([0 if seed is None else seed], np.uint64), replay_constructor=False)

    @property
    def agent_positions(self):
        s = self._engine.download("sd_state")[0]
        return {k: float(s[i]) for i, k in enumerate(self._keys)}

    @property
    def agent_vels(self):
        s, n = self._engine.download("sd_state")[0], self.num_agents
        return {k: float(s[n + i]) for i, k in enumerate(self._keys)}

    @property
    def crossed_agents(self):
        s, n = self._engine.download("sd_state")[0], self.num_agents
        return ["a%d" % int(x) for x in s[4 * n + 2:4 * n + 2 + int(s[4 * n + 1])]]

    def _base_obs(self, keys):
        L = 2 * self.num_agents + 5
        ob = self._engine.download("obs_f64")[0]
        return {k: ob[int(k[1:]), :L].copy() for k in keys}

    def reset(self):
        self._call(self._ensure_engine().reset)
        self.metrics = {"transfers": 0}
        return self._base_obs(self._keys)

    def _step_engine(self, acts):
        n = self.num_agents
        a = np.zeros((1, n), np.float32)
        active = np.zeros((1, n), np.uint8)
        for k, v in acts.items():
            i = int(k[1:])
            a[0, i] = np.float32(np.asarray(v).reshape(-1)[0])
            active[0, i] = 1
        self._call(self._ensure_engine().step, a, active)
        f = int(self._engine.download("error_flags")[0])
        if f & 4:  # the reference raises AttributeError here (collision_check_all is undefined, :160)
            raise AttributeError("'SelfAcceleratingCarEnv' object has no attribute 'collision_check_all'")

    def _dones(self):
        da = self._engine.download("done_agents")[0]
        d = {k: bool(da[i]) for i, k in enumerate(self._keys)}
        d["__all__"] = bool(self._engine.download("done")[0])
        return d

    def _infos(self, keys):
        """the reference's infos dict (:183-189, update_infos :127-149): ambulance stats and is_crashed ride on the first
        key of the action dict, every other key carries zeros there"""
        info = self._engine.download("info")[0]
        rank, dtf = (float(x) for x in self._engine.download("sd_info")[0])
        out = {}
        for k in keys:
            first = k == keys[0]
            out[k] = {"just_passed": bool(info[int(k[1:]), 0]), "is_crashed": int(info[int(k[1:]), 1]) if first else 0,
                      "ambulance_rank": (int(rank) if first else 0.0), "ambulance_dist_to_front": (dtf if first else 0.0)}
        return out

    def step(self, acts):
        keys = list(acts.keys())
        self._step_engine(acts)
        base = self._engine.download("base_reward")[0]  # -1 / -100 (ambulance) / -10000 (crash), before any contract
        r = {k: float(base[int(k[1:])]) for k in keys}
        if int(self._engine.download("info")[0][int(keys[0][1:]), 1]):  # crash: every car is penalised, acting or not (:211)
            r = {k: -10000.0 for k in self._keys}
        self.metrics = episode_metrics("selfdrive", self.num_agents, None, self._engine.download("f64_metrics")[0], False)
        return self._base_obs(keys), r, self._dones(), self._infos(keys)

    def render(self, mode="rgb"):
        return True

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
