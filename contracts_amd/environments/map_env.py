"""GridEnvAdapter — the RLlib `MultiAgentEnv`-shaped face of one HIP-stepped grid env.

Mirrors the interface of the reference's `MapEnv` subclasses (environments/map_env.py:60-342,
cleanup_new.py:59-267, harvest_new.py:48-239): constructor kwargs, `reset()`/`step()` dictionaries
keyed 'a0'..'a{n-1}', spaces, `metrics`, `seed`, `compute_equality/compute_sustainability`.  All
stepping happens in the engine (one wavefront per env); this class only converts between the engine's
arrays and the reference's Python containers.

RNG semantics.  The reference draws from the PROCESS-GLOBAL `np.random`.  With `rng="global"` (default)
the adapter uploads the current global MT19937 state before every engine call and installs the advanced
state afterwards, so `np.random.seed(s); env = CleanupEnv(...); env.reset(); env.step(...)` consumes the
very same stream, interleaved correctly with any other user of `np.random` in the process.
`rng="private"` keeps a per-env stream inside the engine (what the batched API uses).
"""
import os

import numpy as np

from .. import spaces
from .._lib import static_map
from ..engine import BatchedEnv
from .metrics import episode_metrics

try:  # the real base class when RLlib is installed, so RLlib's isinstance checks pass
    from ray.rllib.env import MultiAgentEnv as _Base  # pragma: no cover
except Exception:
    _Base = object

from .vector_hook import VectorHookMixin  # noqa: E402  (RLlib's to_base_env entry point)

VIEW = 7
# colour LUT of the reference (map_env.py:24-42, cleanup_new.py:42-47), indexed by engine cell code,
# then agents '1'..'9'
CELL_RGB = np.array([[0, 0, 0], [180, 180, 180], [0, 255, 0], [99, 156, 194], [113, 75, 24], [113, 75, 24]], np.uint8)
AGENT_RGB = np.array([[0, 0, 255], [2, 81, 154], [204, 0, 204], [216, 30, 54], [254, 151, 0], [100, 255, 255],
                      [99, 99, 255], [250, 204, 255], [238, 223, 16]], np.uint8)
BEAM_RGB = {1: (255, 255, 0), 2: (100, 255, 255)}  # CE_BEAM_FIRE b"F" (map_env.py:30), CE_BEAM_CLEAN b"C" (cleanup_new.py:43)
CELL_CHARS = np.array([b" ", b"@", b"A", b"H", b"R", b"S"], dtype="S1")
ORIENT_NAMES = ["UP", "RIGHT", "DOWN", "LEFT"]


def push_global_rng(engine, python_random=False):
    st = np.random.get_state(legacy=True)
    words = np.zeros((1, engine.b.rng_words), np.uint32)
    words[0, :624] = st[1]
    words[0, 624] = st[2]
    if python_random:
        import random
        ps = random.getstate()
        words[0, 628:628 + 624] = np.array(ps[1][:624], np.uint32)
        words[0, 628 + 624] = ps[1][624]
    engine.upload("rng", words)
    return st


def pull_global_rng(engine, st, python_random=False):
    words = engine.download("rng")[0]
    np.random.set_state((st[0], words[:624].copy(), int(words[624]), st[3], st[4]))
    if python_random:
        import random
        ps = random.getstate()
        random.setstate((ps[0], tuple(int(x) for x in words[628:628 + 624]) + (int(words[628 + 624]),), ps[2]))


def restore_pending_state(env, engine):
    """upload the env state a pickled adapter carried over (`__getstate__`), after checking that it was written by the
    same engine ABI: field layouts (metric rows, grid pitch, ...) follow the ABI version, and an older blob would
    otherwise fail with an opaque size error inside ce_upload"""
    from .._lib import CE_ABI_VERSION
    pending = getattr(env, "_pending_state", None)
    if not pending:
        return
    abi = getattr(env, "_pending_abi", None)
    if abi != CE_ABI_VERSION:
        raise ValueError("pickled %s carries env state of engine ABI v%s, this build is ABI v%d: re-create the env instead "
                         "of unpickling it" % (type(env).__name__, abi, CE_ABI_VERSION))
    for field, arr in pending.items():
        engine.upload(field, arr)
    env._pending_state = None


def host_np_draw(env, fn):
    """fn(rs) with `rs` the numpy stream the reference would draw from right now: the process-global np.random when the
    adapter mirrors it (rng="global": it is in step with the engine between calls), else the env's private MT19937
    inside the engine, fetched, advanced on the host and written back"""
    if env._rng_mode == "global":
        return fn(np.random)
    eng = env._ensure_engine()
    words = np.array(eng.download("rng"), np.uint32)
    rs = np.random.RandomState()
    rs.set_state(("MT19937", words[0, :624].copy(), int(words[0, 624])))
    out = fn(rs)
    st = rs.get_state()
    words[0, :624], words[0, 624] = st[1], st[2]
    eng.upload("rng", words)
    return out


class AgentView:
    """read-only stand-in for the reference's agent objects (Agent.py:25-160): where an agent is and which way it faces,
    from the engine's state; `env.agents['a3'].pos`, `.orientation`, `.get_char_id()` keep working"""

    def __init__(self, agent_id, row, col, orientation, view_len):
        self.agent_id = agent_id
        self.pos = np.array([row, col])
        self.list_pos = [row, col]
        self.int_orientation = int(orientation)
        self.orientation = ORIENT_NAMES[self.int_orientation]
        self.row_size = self.col_size = view_len

    def get_pos(self):
        return self.pos

    def get_orientation(self):
        return self.orientation

    def get_char_id(self):
        return bytes(str(int(self.agent_id[-1]) + 1), encoding="ascii")  # Agent.py:81-82

    def translate_pos_to_egocentric_coord(self, pos):
        return [self.row_size, self.col_size] + (np.asarray(pos) - self.pos)


class GridEnvAdapter(VectorHookMixin, _Base):
    KIND = None          # "cleanup" | "harvest"
    GRID_SHAPE = None    # (H, W)
    N_ACTIONS = None     # (disable_firing=True, False)

    def __init__(self, ascii_map=None, num_agents=1, disable_firing=True, image_obs=True, return_agent_actions=False,
                 use_collective_reward=False, inequity_averse_reward=False, alpha=0.0, beta=0.0, horizon=1000,
                 one_hot_id=False, rng="global", device=0, vector_rng=None, **kwargs):
        # the reference's constructors pass their module's map explicitly (cleanup_new.py:62, harvest_new.py:51); any other
        # layout goes to the engine as ce_config.ascii_map (walled in, within the shipped layout's frame and cell counts:
        # ce_create names the rule a layout breaks)
        rows = None if ascii_map is None else [r.decode("ascii") if isinstance(r, bytes) else str(r) for r in ascii_map]
        self._map_rows = None if rows is None or rows == static_map(self.KIND) else rows
        if self._map_rows is not None:
            rows = self._map_rows
            if not rows or any(len(r) != len(rows[0]) for r in rows):
                raise ValueError("ascii_map must be a non-empty list of equally long strings")
            self.GRID_SHAPE = (len(rows), len(rows[0]))
            self.N_APPLE_CELLS = sum(r.count("B" if self.KIND == "cleanup" else "A") for r in rows)
            if self.KIND == "cleanup":
                self.POTENTIAL_WASTE_AREA = sum(r.count("H") + r.count("R") for r in rows)
        if inequity_averse_reward:
            assert num_agents > 1, "Cannot use inequity aversion with only one agent!"  # map_env.py:294
        self.num_agents = num_agents
        self.disable_firing = disable_firing
        self.image_obs = image_obs
        self.return_agent_actions = return_agent_actions  # no effect on the returned obs (cleanup_new.py:258)
        self.use_collective_reward = use_collective_reward
        self.inequity_averse_reward = inequity_averse_reward
        self.alpha, self.beta = alpha, beta
        self.horizon = horizon
        self.one_hot_id = one_hot_id
        self.view_len = self.map_padding = VIEW
        self._rng_mode = rng
        self._device = device
        # stream of the batched hook this env's to_base_env() builds (vector_hook.py): "mt19937" = the reference's, "counter" =
        # the engine's Philox stream (faster, no np.random.seed trace); the single env itself always runs the reference's
        self.vector_rng = vector_rng or os.environ.get("CONTRACTS_AMD_VECTOR_RNG", "mt19937")
        if self.vector_rng not in ("mt19937", "counter"):
            raise ValueError("vector_rng must be 'mt19937' or 'counter', got %r" % (self.vector_rng,))
        self._engine = None
        self._contract = (None, None, None, 0.0)
        self._keys = ["a%d" % i for i in range(num_agents)]
        self.metrics = {}
        self._build_spaces()
        self._ensure_engine()
        # the reference constructor consumes RNG (MapEnv.__init__ -> setup_agents, map_env.py:131)
        self._call(self._engine.construct)

    # ------------------------------------------------------------------ engine plumbing
    def _ensure_engine(self):
        if self._engine is None:
            self._engine = BatchedEnv(
                self.KIND, 1, self.num_agents, horizon=self.horizon, firing=not self.disable_firing,
                collective=self.use_collective_reward, inequity=self.inequity_averse_reward, alpha=self.alpha,
                beta=self.beta, device=self._device, beam_trace=True,  # render() overlays the step's beams
                ascii_map=self._map_rows)
            c, lo, hi, null_prob = self._contract
            if c is not None:
                self._engine.set_contract(c, lo, hi, null_prob)
            if getattr(self, "_external_theta", False):
                self._engine.set_flags(external_theta=True)
            restore_pending_state(self, self._engine)
        return self._engine

    def _call(self, fn, *args):
        eng = self._ensure_engine()
        if self._rng_mode == "global":
            st = push_global_rng(eng)
            fn(*args)
            pull_global_rng(eng, st)
        else:
            fn(*args)

    _STATE_FIELDS = ("grid", "agents", "spawn_perm", "waste_perm", "rng", "timestep", "theta", "int_metrics",
                     "f64_metrics", "final_int_metrics", "final_f64_metrics", "beam_map")

    def __getstate__(self):
        # the instance is shipped inside RLlib's env_config (ray_config_utils.py:198-202): drop the device
        # handle, keep the env state, re-create the engine lazily on the other side
        d = dict(self.__dict__)
        eng = d.pop("_engine", None)
        if eng is not None:
            fields = [f for f in self._STATE_FIELDS if not (f == "waste_perm" and self.KIND != "cleanup")]
            d["_pending_state"] = {f: eng.download(f, raw=True) for f in fields}
            from .._lib import CE_ABI_VERSION
            d["_pending_abi"] = CE_ABI_VERSION
        d["_engine"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ------------------------------------------------------------------ spaces
    def _build_spaces(self):
        n, (H, W) = self.num_agents, self.GRID_SHAPE
        na = self.N_ACTIONS[0] if self.disable_firing else self.N_ACTIONS[1]
        self.action_space = spaces.Discrete(na)
        self.continuous_action_space = spaces.Box(low=-10.0, high=10.0, shape=(na,))
        self.global_observation_space = spaces.Dict({"image": spaces.Box(low=0, high=1, shape=(H, W, 3), dtype=np.uint8)})
        self.concatenated_observation_space = spaces.Dict(
            {"image": spaces.Box(low=0, high=1, shape=(15, 15, 3 * n), dtype=np.uint8)})
        self.global_action_space = spaces.MultiDiscrete([na] * n)
        if not self.image_obs:
            self.observation_space = self._feature_space()
        else:
            img = spaces.Box(low=0, high=1, shape=(2 * VIEW + 1, 2 * VIEW + 1, 3), dtype=np.uint8)
            if not self.one_hot_id:
                self.observation_space = spaces.Dict({"image": img})
            else:
                self.observation_space = spaces.Dict({"image": img, "features": spaces.Box(low=0, high=1, shape=(n,))})

    # ------------------------------------------------------------------ reference API
    def seed(self, seed=None):
        """MapEnv.seed (map_env.py:344-345) == np.random.seed(seed)"""
        self._vector_seed0 = seed  # to_base_env(num_envs > 1): replica i is seeded seed + i
        if self._rng_mode == "global":
            np.random.seed(seed)
        else:
            self._ensure_engine().seed(np.array([0 if seed is None else seed], np.uint64), replay_constructor=False)

    def one_hot(self, key):
        v = np.zeros(self.num_agents)
        v[int(key[1:])] = 1
        return v

    def _obs_dict(self, feats=None):
        eng = self._engine
        if not self.image_obs:
            f = eng.download("features")[0].astype(np.float64) if feats is None else feats
            return {k: f[i] for i, k in enumerate(self._keys)}
        img = eng.download("obs")[0]
        out = {}
        for i, k in enumerate(self._keys):
            o = {"image": img[i] / 255}  # uint8/255 -> float64, as cleanup_new.py:258 / harvest_new.py:229
            if self.one_hot_id:
                o["features"] = self.one_hot(k)
            out[k] = o
        return out

    def reset(self):
        self._call(self._ensure_engine().reset)
        self._engine.check_faults()
        self._agents_painted = False
        self._refresh_metrics(final=False)
        return self._obs_dict()

    _STEP_FIELDS = ("rng", "error_flags", "base_reward", "reward", "done", "info", "features", "int_metrics", "f64_metrics",
                    "final_int_metrics", "final_f64_metrics")

    def step(self, acts):
        """one engine step and ONE fetch of its whole result (ce_download_many): 11 per-field copies made a step() call
        0.44 ms, two thirds of it in the copies"""
        a = np.zeros((1, self.num_agents), np.uint8)
        for i, k in enumerate(self._keys):
            v = int(acts[k])
            if not 0 <= v < self.N_ACTIONS[1]:
                raise KeyError(v)  # Agent.action_map raises KeyError on an unknown id (Agent.py:174-176,213-215)
            a[0, i] = v
        eng = self._ensure_engine()
        glob = self._rng_mode == "global"
        st = push_global_rng(eng) if glob else None
        eng.step(a)
        out = eng.download_many(self._STEP_FIELDS + (("obs",) if self.image_obs else ()))
        if glob:  # install the advanced stream as the process-global generator (pull_global_rng)
            words = out["rng"][0]
            np.random.set_state((st[0], words[:624].copy(), int(words[624]), st[3], st[4]))
        if out["error_flags"][0]:
            eng.check_faults()
        self._agents_painted = True
        base, rew_f = out["base_reward"][0], out["reward"][0]
        float_rewards = self.inequity_averse_reward
        r = {k: (float(rew_f[i]) if float_rewards else int(base[i])) for i, k in enumerate(self._keys)}
        done = bool(out["done"][0])
        d = {"__all__": done, "a0": done, "a1": done}  # cleanup_new.py:242 / harvest_new.py:214 (sic)
        info, feats = out["info"][0], out["features"][0].astype(np.float64)
        infos = {}
        for i, k in enumerate(self._keys):
            infos[k] = self._info_entry(int(info[i, 0]), int(info[i, 1]))
            infos[k]["feature_obs"] = feats[i]
        self._set_metrics(out["final_int_metrics" if done else "int_metrics"][0],
                          out["final_f64_metrics" if done else "f64_metrics"][0], done)
        if not self.image_obs:
            return {k: feats[i] for i, k in enumerate(self._keys)}, r, d, infos
        img = eng._dense_obs(out["obs"])[0]
        obs = {}
        for i, k in enumerate(self._keys):
            o = {"image": img[i] / 255}  # uint8/255 -> float64, as cleanup_new.py:258 / harvest_new.py:229
            if self.one_hot_id:
                o["features"] = self.one_hot(k)
            obs[k] = o
        return obs, r, d, infos

    # ------------------------------------------------------------------ state views (host copies)
    @property
    def agent_pos(self):
        return [[int(a[0]), int(a[1])] for a in self._engine.download("agents")[0]]

    @property
    def timesteps(self):
        return int(self._engine.download("timestep")[0])

    @property
    def world_map(self):
        return CELL_CHARS[self._engine.download("grid")[0]]

    # ---- inspection helpers of MapEnv / CleanupEnv / HarvestEnv, computed from the engine's state on demand ----
    @property
    def agents(self):
        a = self._engine.download("agents")[0]
        return {k: AgentView(k, int(a[i, 0]), int(a[i, 1]), int(a[i, 2]), self.view_len) for i, k in enumerate(self._keys)}

    def get_map_with_agents(self):
        """map_env.py:353-374: the character map with the agents' ids ('1'..'9', later agents on top) and, over them,
        the beams of the last step"""
        grid = self.world_map.copy()
        for i, a in enumerate(self._engine.download("agents")[0]):
            grid[a[0], a[1]] = bytes(str(i + 1), encoding="ascii")
        beam = self._engine.download("beam_map")[0]
        grid[beam == 1] = b"F"
        grid[beam == 2] = b"C"
        return grid

    def color_view(self, agent):
        """map_env.py:397-411: the agent's rotated 15 x 15 crop — the engine's observation of the last step / reset"""
        key = agent.agent_id if hasattr(agent, "agent_id") else agent
        return self._engine.download("obs")[0][int(key[1:])]

    def test_if_in_bounds(self, pos):
        H, W = self.GRID_SHAPE
        return 0 <= pos[0] < H and 0 <= pos[1] < W

    def find_visible_agents(self, agent_id):
        """map_env.py:880-913: which other agents (sorted by id) stand inside this agent's 15 x 15 window"""
        a = self._engine.download("agents")[0][:, :2].astype(int)
        me = a[int(agent_id[1:])]
        others = [a[int(k[1:])] for k in sorted(self._keys) if k != agent_id]
        return np.array([1 if (abs(o[0] - me[0]) <= self.view_len and abs(o[1] - me[1]) <= self.view_len) else 0 for o in others],
                        dtype=np.uint8)

    def _cells(self, char):
        return [[int(r), int(c)] for r, c in np.argwhere(self.world_map == char)]

    @property
    def current_apple_points(self):
        return self._cells(b"A")  # row-major scan of the map as it is now (cleanup_new.py:378-385, harvest_new.py)

    def compute_current_apples(self):
        """the reference refreshes its `current_apple_points` attribute here; the property above is always current"""

    @property
    def _static_rows(self):
        return getattr(self, "_map_rows", None) or static_map(self.KIND)

    @property
    def apple_points(self):
        ch = "B" if self.KIND == "cleanup" else "A"
        return [[r, c] for r, row in enumerate(self._static_rows) for c, x in enumerate(row) if x == ch]

    @property
    def spawn_points(self):
        """the spawn list in its current (persistently shuffled) order, map_env.py:821"""
        pts = [[r, c] for r, row in enumerate(self._static_rows) for c, x in enumerate(row) if x == "P"]
        if self.KIND == "cleanup":
            pts = pts + pts  # the constructor collects the 'P' cells a second time (cleanup_new.py:114-115): 20 entries
        return [pts[i] for i in self._engine.download("spawn_perm")[0][:len(pts)]]

    def full_map_to_colors(self):
        """map_env.py:354-392: the map, the agents in agent order, then the beams the last step fired (`beam_pos`,
        kept by the engine under CE_FLAG_BEAM_TRACE) — what run_render.py:41 collects frame by frame"""
        grid = self._engine.download("grid")[0]
        rgb = CELL_RGB[grid].astype(int)
        for i, a in enumerate(self._engine.download("agents")[0]):
            rgb[a[0], a[1]] = AGENT_RGB[i]
        beam = self._engine.download("beam_map")[0]
        for code, colour in BEAM_RGB.items():
            rgb[beam == code] = colour
        return rgb

    def render(self, filename=None, mode="human"):
        """map_env.py:460-475: 'human' draws (or saves) with matplotlib and returns None, any other mode returns the array"""
        rgb_arr = self.full_map_to_colors()
        if mode == "human":
            import matplotlib.pyplot as plt
            plt.cla()
            plt.imshow(rgb_arr, interpolation="nearest")
            if filename is None:
                plt.show(block=False)
            else:
                plt.savefig(filename)
            return None
        return rgb_arr

    def global_view(self):
        """map_env.py:394-395: the colour map without its padding.  reset() leaves the agents unpainted, a step paints
        them (later agent wins), exactly as the egocentric crops see it."""
        grid = self._engine.download("grid")[0]
        rgb = CELL_RGB[grid]
        if getattr(self, "_agents_painted", False):
            rgb = rgb.copy()
            for i, a in enumerate(self._engine.download("agents")[0]):
                rgb[a[0], a[1]] = AGENT_RGB[i]
        return rgb

    def get_global_obs(self):
        return {"image": self.global_view() / 255}

    # ------------------------------------------------------------------ metrics
    def compute_equality(self, reward_dict):
        """cleanup_new.py:422-434 (host helper kept for API parity; the env's own metrics come from the engine)"""
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
            t_sum = 0
            for t, i in enumerate(reward_dict[k]):
                t_sum += t * i
            denom = max(sum(reward_dict[k]), 1)
            avg_times.append(t_sum / denom)
        return np.mean(avg_times)

    def _refresh_metrics(self, final):
        eng = self._engine
        self._set_metrics(eng.download("final_int_metrics" if final else "int_metrics")[0],
                          eng.download("final_f64_metrics" if final else "f64_metrics")[0], final)

    def _set_metrics(self, mi, mf, final):
        self.metrics = episode_metrics(self.KIND, self.num_agents, mi, mf, final, contract=self._contract[0] is not None,
                                       inequity=self.inequity_averse_reward)
