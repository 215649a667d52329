"""HarvestFeatures / CleanupFeatures — drop-ins for environments/harvest_features.py:60-336 and
environments/cleanup_features.py:48-309 (the `harvest` / `cleanup` envs of the configs; BASELINE config 0), stepped
by the HIP engine.  Same constructor kwargs, spaces, `reset()` / `step()` dictionaries keyed 'a0'..'a{n-1}',
`metrics`, `compute_equality` / `compute_sustainability`.

The reference draws the spawn shuffle and the respawn doubles from the process-global `random` and the
orientations from the process-global `np.random`; with rng="global" (default) both generator states are handed to
the engine before every call and installed again afterwards, so seeded scripts reproduce.

`HarvestFeatures(image_obs=True)` returns crops of the reference's incrementally painted colour map
(harvest_features.py:124-125,259-264): a history-dependent rendering (a cell two agents shared turns black when the
first one leaves), kept by the adapter on the host from the positions and apple lists the engine returns each step —
it is a view for the demo `__main__`s, not part of the stepped state.  CleanupFeatures stores the flag and never reads
it (cleanup_features.py:55), as here."""
import numpy as np

from .. import spaces
from .._lib import static_map
from ..engine import BatchedEnv
from .map_env import _Base, pull_global_rng, push_global_rng, restore_pending_state
from .vector_hook import VectorHookMixin
from .metrics import episode_metrics

HARVEST_SHAPE, CLEANUP_SHAPE = (16, 38), (25, 18)
N_APPLE = {"harvest_features": 155, "cleanup_features": 103}
POTENTIAL_WASTE_AREA = 119


class _FeatureEnv(VectorHookMixin, _Base):
    KIND = None
    N_ACTIONS = None

    def __init__(self, num_agents=2, horizon=1000, image_obs=False, rng="global", device=0, **kwargs):
        self.num_agents = num_agents
        self.horizon = horizon
        self.image_obs = image_obs
        self.timesteps = 0
        self._keys = ["a%d" % i for i in range(num_agents)]
        self._rng_mode, self._device = rng, device
        self._engine = None
        self._contract = (None, None, None, 0.0)
        self.metrics = {}
        self._build_spaces()
        self._call(self._ensure_engine().construct)  # __init__ shuffles the spawn points and draws orientations
        self._after_layout()

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self):
        if self._engine is None:
            self._engine = BatchedEnv(self.KIND, 1, self.num_agents, horizon=self.horizon, device=self._device)
            c, lo, hi, null_prob = self._contract
            if c is not None:
                self._engine.set_contract(c, lo, hi, null_prob)
            if getattr(self, "_external_theta", False):
                self._engine.set_flags(external_theta=True)
            restore_pending_state(self, self._engine)
        return self._engine

    _paints = False  # HarvestFeatures(image_obs=True) only

    _STATE_FIELDS = ("grid", "agents", "rng", "timestep", "theta", "int_metrics", "f64_metrics", "final_int_metrics",
                     "final_f64_metrics")

    def __getstate__(self):
        d = dict(self.__dict__)
        eng = d.pop("_engine", None)
        if eng is not None:
            d["_pending_state"] = {f: eng.download(f, raw=True) for f in self._STATE_FIELDS}
            from .._lib import CE_ABI_VERSION
            d["_pending_abi"] = CE_ABI_VERSION
        d["_engine"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    _RESULT_FIELDS = ("rng", "error_flags", "done", "base_reward", "reward", "info", "features", "int_metrics", "f64_metrics", "final_int_metrics", "final_f64_metrics", "theta")

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

    # ------------------------------------------------------------------ reference-visible state
    @property
    def agent_pos(self):
        a = self._engine.download("agents")[0]
        return {k: [int(a[i, 0]), int(a[i, 1])] for i, k in enumerate(self._keys)}

    @property
    def agent_orientation(self):
        a = self._engine.download("agents")[0]
        return {k: int(a[i, 2]) for i, k in enumerate(self._keys)}

    def _feature_obs(self, keys):
        f = self._engine.download("features")[0].astype(np.float64)
        return {k: f[int(k[1:])].copy() for k in keys}

    def _refresh_metrics(self, final):
        eng, n = self._engine, self.num_agents
        mi = eng.download("final_int_metrics" if final else "int_metrics")[0]
        mf = eng.download("final_f64_metrics" if final else "f64_metrics")[0]
        self.metrics = episode_metrics(self.KIND, n, mi, mf, final, contract=self._contract[0] is not None)
        # total_reward_dict holds per-step lists in the reference; the engine keeps the two sums the metrics need
        self._sum_r = [int(mi[4 + 2 * n + i]) for i in range(n)]

    def _after_layout(self):
        """constructor / reset hook: the agents have just been placed"""

    def _after_step(self, actions):
        pass

    def _observations(self, keys):
        return self._feature_obs(keys)

    def reset(self):
        self._call(self._ensure_engine().reset)
        self.timesteps = 0
        self._refresh_metrics(False)
        self._after_layout()
        return self._observations(self._keys)

    def step(self, acts):
        keys = list(acts.keys())
        if keys != self._keys:
            raise KeyError("the engine steps all %d agents; got actions for %s" % (self.num_agents, keys))
        a = np.array([[int(acts[k]) for k in self._keys]], np.int64)
        if a.min() < 0 or a.max() > self.N_ACTIONS:
            raise IndexError("action out of range for %s" % type(self).__name__)
        self._call(self._ensure_engine().step, a.astype(np.uint8))
        eng = self._engine
        self.timesteps += 1
        done = bool(eng.download("done")[0])
        base = eng.download("base_reward")[0]
        info = eng.download("info")[0]
        feats = self._feature_obs(keys)
        rewards = {k: float(base[i]) for i, k in enumerate(self._keys)}
        infos = self._infos(info, feats)
        dones = {"__all__": done, "a0": done, "a1": done}
        self._refresh_metrics(done)
        self._after_step(a[0])
        return (feats if not self._paints else self._observations(keys)), rewards, dones, infos

    def render(self):
        pass

    # the reference computes these from total_reward_dict lists; kept for interface parity on dicts of lists
    def compute_equality(self, reward_dict):
        eq, total_sum = 0, 0
        reward_dict = {k: sum(v) for k, v in reward_dict.items()}
        n = len(reward_dict.keys())
        for i in reward_dict.keys():
            for j in reward_dict.keys():
                eq += abs(reward_dict[i] - reward_dict[j])
            total_sum += reward_dict[i]
        if total_sum == 0:
            total_sum = 0.001
        return 1 - eq / (2 * n * total_sum)

    def compute_sustainability(self, reward_dict):
        avg_times = []
        for k in reward_dict.keys():
            t_sum = sum(t * i for t, i in enumerate(reward_dict[k]))
            avg_times.append(t_sum / max(sum(reward_dict[k]), 1))
        return np.mean(avg_times)

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


HARVEST_VIEW_SIZE = 7
_WALL_RGB, _APPLE_RGB, _PLAYER_RGB = (180, 180, 180), (0, 255, 0), (159, 67, 255)  # DEFAULT_COLOURS b"@", b"A", b"P" (harvest_features.py:39-57)


class HarvestFeatures(_FeatureEnv):
    KIND = "harvest_features"
    N_ACTIONS = 7  # Discrete(7); the code path also accepts 7 (a fire action without effect)

    def _build_spaces(self):
        H, W = HARVEST_SHAPE
        na = N_APPLE[self.KIND]
        self._paints = bool(self.image_obs)
        if self._paints:
            self.map = static_map(self.KIND)
            self._apple_cells = np.array([(r, c) for r in range(H) for c in range(W) if self.map[r][c] == "A"])
            self._wall = np.array([[ch == "@" for ch in row] for row in self.map])
            self.observation_space = spaces.Box(low=0, high=255, shape=(2 * HARVEST_VIEW_SIZE + 1, 2 * HARVEST_VIEW_SIZE + 1, 3),
                                                dtype=np.uint8)
            self.action_space = spaces.Discrete(7)
            self.continuous_action_space = spaces.Box(low=-10.0, high=10.0, shape=(7,))
            return
        self.observation_space = spaces.Box(low=np.array([0.0] * (10 + 2 * self.num_agents)),
                                            high=np.array([H, W, 4, H, W, 4, H, W, na + 1, na + 1] + [1] * (2 * self.num_agents)))
        self.action_space = spaces.Discrete(7)
        self.continuous_action_space = spaces.Box(low=-10.0, high=10.0, shape=(7,))

    def _infos(self, info, obs):
        return {k: {"eaten_apples": int(info[i, 0]), "eaten_close_apples": int(info[i, 1]), "feature_obs": obs[k]}
                for i, k in enumerate(self._keys)}

    # ---- image_obs=True: the painted colour map (single_update_map, harvest_features.py:124-125) ----
    def _state_now(self):
        pos = self._engine.download("agents")[0][:, :2].astype(np.int64)
        present = self._engine.feature_state()[0][0][:len(self._apple_cells)] != 0xFFFF
        return pos, present

    def _paint(self, cells, rgb):
        self.world_map_color[cells[:, 0] + HARVEST_VIEW_SIZE, cells[:, 1] + HARVEST_VIEW_SIZE] = rgb

    def _after_layout(self):
        """__init__ / reset: a fresh map; initialize_arrays paints the walls and EVERY apple point (:99-113),
        initialize_players the agents (:115-122)"""
        if not self._paints:
            return
        H, W = HARVEST_SHAPE
        self.world_map_color = np.zeros((H + 2 * HARVEST_VIEW_SIZE, W + 2 * HARVEST_VIEW_SIZE, 3), np.uint8)
        self._paint(np.argwhere(self._wall), _WALL_RGB)
        self._paint(self._apple_cells, _APPLE_RGB)
        self._pos, self._present = self._state_now()
        self._paint(self._pos, _PLAYER_RGB)

    def _after_step(self, actions):
        """a move that went through blackens the cell left and paints the cell entered, agent by agent in key order
        (:194-195) — eating paints nothing more (:206); then the apples spawned this step turn green (:151)"""
        if not self._paints:
            return
        pos, present = self._state_now()
        for i in range(self.num_agents):
            if actions[i] < 4 and (pos[i] != self._pos[i]).any():
                self._paint(self._pos[i:i + 1], (0, 0, 0))
                self._paint(pos[i:i + 1], _PLAYER_RGB)
        self._paint(self._apple_cells[present & ~self._present], _APPLE_RGB)
        self._pos, self._present = pos, present

    def _observations(self, keys):
        if not self._paints:
            return self._feature_obs(keys)
        w = 2 * HARVEST_VIEW_SIZE + 1
        return {k: self.world_map_color[self._pos[int(k[1:])][0]:self._pos[int(k[1:])][0] + w,
                                        self._pos[int(k[1:])][1]:self._pos[int(k[1:])][1] + w].copy() for k in keys}


class CleanupFeatures(_FeatureEnv):
    KIND = "cleanup_features"
    N_ACTIONS = 8  # Discrete(8); the code path also accepts 8 (a beam without effect)

    def _build_spaces(self):
        H, W = CLEANUP_SHAPE
        self.potential_waste_area = POTENTIAL_WASTE_AREA
        self.observation_space = spaces.Box(low=np.array([0.0] * (12 + self.num_agents)),
                                            high=np.array([H, W, 4, H, W, 4, H, W, H, W, N_APPLE[self.KIND] + 1,
                                                           self.potential_waste_area + 1] + [np.inf] * self.num_agents))
        self.action_space = spaces.Discrete(8)
        self.continuous_action_space = spaces.Box(low=-10.0, high=10.0, shape=(8,))

    def _infos(self, info, obs):
        return {k: {"cleaned_squares": int(info[i, 1])} for i, k in enumerate(self._keys)}
