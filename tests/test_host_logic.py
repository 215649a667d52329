"""CPU: host-side mirror of the reference interface — registry tags, contract specs, spaces, sharding."""
import numpy as np
import pytest


def test_registry_tags():
    from contracts_amd.utils import env_creator_functions as ecf
    assert ecf.get_base_env_tag({"environment": "cleanup_new"}) == "CleanupNew"
    assert ecf.get_base_env_tag({"environment": "harvest_new"}) == "HarvestNew"
    assert ecf.get_base_env_tag({"environment": "selfdrive"}) == "SelfDrive"
    with pytest.raises(AssertionError):
        ecf.get_base_env_tag({"environment": "nope"})
    with pytest.raises(ValueError):
        ecf.env_creator("Nope", {})
    # every tag the reference registers (utils/env_creator_functions.py:12-46) resolves to a HIP-backed class
    assert set(ecf._ACCELERATED) == {"SelfDrive", "Harvest", "HarvestNew", "Cleanup", "CleanupNew", "ContractWrapperSubgame",
                                     "ContractWrapperNegotiate", "ContractWrapperCombined", "JointEnv", "NegotiationSolver"}


def test_contract_specs():
    from contracts_amd.contract import contract_list as cl
    c = cl.CleanupContract(4)
    # gym Box semantics: float32 bounds, read back as float64 by the wrapper (two_stage_train.py:39-40)
    assert c.contract_space.high.dtype == np.float32
    assert float(c.contract_space.high[0]) == float(np.float32(0.2)) != 0.2
    assert cl.HarvestFeaturemodLocalContract(8).contract_space.high[0] == 10.0
    assert cl.SelfdriveContractDistprop(4).contract_space.high[0] == 100.0
    assert c.engine_contract == "cleanup" and c.num_agents == 4 and c.default_contract[0] == 0.0


def _flatten_transfers(tr, slots):
    val, tup, share, present = np.zeros(slots), np.zeros(slots, np.uint8), np.zeros((slots, slots)), np.zeros(slots, np.uint8)
    for k, v in tr.items():
        i = int(k[1:])
        present[i] = 1
        if type(v) is tuple:
            tup[i], val[i] = 1, v[0]
            for r, p in v[1].items():
                share[i, int(r[1:])] = p
        else:
            val[i] = v
    return val, tup, share, present


def test_contract_compute_transfer_matches_reference_vectors():
    """the public `compute_transfer` of the three contract classes (contract_list.py:22-27,45-54,69-102) against vectors
    recorded from the reference (tests/golden/make_contract_golden.py): same keys, same numbers, same (value, shares)
    tuples.  (The wrapper never calls it — the transfer is the step kernel's epilogue; this is the method a user calls
    on a contract object directly.)"""
    import os
    from contracts_amd.contract import contract_list as cl
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "contract_transfers.npz"))
    for name, cls in (("cleanup", cl.CleanupContract), ("harvest", cl.HarvestFeaturemodLocalContract)):
        for row in g[name]:
            n, theta = int(row[0]), row[1]
            cleaned, f8, close, val, present = row[2:10], row[10:18], row[18:26], row[26:34], row[34:42]
            keys = ["a%d" % i for i in range(n)]
            infos = {k: {"cleaned_squares": int(cleaned[i]), "eaten_close_apples": int(close[i]),
                         "feature_obs": np.r_[np.zeros(8), f8[i], np.zeros(3)]} for i, k in enumerate(keys)}
            tr = cls(n).compute_transfer({}, {k: 0 for k in keys}, {}, {k: np.array([theta]) for k in keys}, infos)
            v, t, s, pr = _flatten_transfers(tr, 8)
            assert np.array_equal(v, val) and np.array_equal(pr, present) and not t.any()
    for row in g["selfdrive"]:
        n, theta, passed = int(row[0]), row[1], bool(row[2])
        acting, obs_row = row[3:11][:n].astype(bool), row[11:31][:2 * n + 5]
        val, tup, share, present = row[31:43], row[43:55], row[55:199].reshape(12, 12), row[199:211]
        keys = ["a%d" % i for i in range(n)]
        acts = {k: 0 for i, k in enumerate(keys) if acting[i]}
        obs = {k: obs_row.copy() for k in acts}
        obs["a0"] = obs_row.copy()
        infos = {k: {"just_passed": passed if k == "a0" else False} for k in keys}
        tr = cl.SelfdriveContractDistprop(n).compute_transfer(obs, acts, {}, {k: np.array([theta]) for k in keys}, infos)
        v, t, s, pr = _flatten_transfers(tr, 12)
        assert np.array_equal(v, val) and np.array_equal(t, tup) and np.array_equal(s, share) and np.array_equal(pr, present)


def test_sharding():
    from contracts_amd.parallel import env_shard, split_envs
    assert env_shard(3, 8, 16384) == (3 * 16384, 16384)
    with pytest.raises(ValueError):
        env_shard(8, 8, 4)
    parts = split_envs(131072, 8)
    assert parts[0] == (0, 16384) and parts[-1] == (7 * 16384, 16384)
    parts = split_envs(10, 4)
    assert [c for _, c in parts] == [3, 3, 2, 2] and [b for b, _ in parts] == [0, 3, 6, 8]


def test_spaces_standins():
    from contracts_amd import spaces
    b = spaces.Box(low=0, high=0.2, shape=(1,))
    assert b.low.dtype == np.float32 and b.shape == (1,)
    d = spaces.Dict({"image": spaces.Box(0, 1, (15, 15, 3), np.uint8)})
    assert "image" in d.keys() and d["image"].shape == (15, 15, 3)
    assert spaces.Discrete(8).n == 8


def test_config_expansion_matches_runner_semantics():
    """the reference's experiment_configs format (runner.py:89-135): [global, sweep, job...]"""
    from contracts_amd.utils.config import expand_config_list
    cfg = [
        {"num_workers": 8, "horizon": 1000, "env_args": {"image_obs": True}, "solver": True},
        {"num_agents": [2, 4, 8], "lr": [1, 2]},
        {"environment": "cleanup_new", "contract": "CleanupContract"},
        {"environment": "harvest_new", "contract": "HarvestFeaturemodLocalContract", "horizon": 500},
    ]
    jobs = expand_config_list(cfg, seeds=2)
    assert len(jobs) == 3 * 2 * 2 * 2
    # sweep-major, then job order, then seeds; job-level values win over globals
    assert [j["num_agents"] for j in jobs[:8]] == [2] * 8 and jobs[0]["lr"] == 1 and jobs[4]["lr"] == 2
    assert jobs[0]["environment"] == "cleanup_new" and jobs[2]["environment"] == "harvest_new"
    assert jobs[2]["horizon"] == 500 and jobs[0]["horizon"] == 1000 and jobs[0]["num_workers"] == 8
    assert jobs[0]["seed"] == 73907 and jobs[1]["seed"] == 2 * 73907
    assert jobs[0]["experiment_name"] == "cleanup_new-2agents"
    assert cfg[2].get("num_workers") is None  # the input list is not mutated


def test_space_sampling_leaves_the_global_stream_alone():
    """gym 0.21 spaces draw from a private np_random; the stand-ins must too — the adapters hand the process-global
    np.random state to the engine, so a .sample() on it would shift the env's own trajectory"""
    from contracts_amd import spaces
    if spaces.Box.__module__.startswith("gym"):
        pytest.skip("real gym present")
    np.random.seed(5)
    before = np.random.get_state()[1].copy(), np.random.get_state()[2]
    b, d = spaces.Box(0.0, 0.2, shape=(1,)), spaces.Discrete(8)
    b.seed(3)
    d.seed(3)
    x = [b.sample() for _ in range(4)] + [d.sample() for _ in range(4)]
    after = np.random.get_state()
    assert np.array_equal(before[0], after[1]) and before[1] == after[2]
    b.seed(3)
    assert np.array_equal(b.sample(), x[0]) and b.sample().dtype == np.float32


def test_synth_action_mirror_matches_the_library():
    """contracts_amd.synth (numpy) == ce_synth_action_host / ce_synth_hash_host (the device generator's host twin)"""
    from contracts_amd import _lib, synth
    L = _lib.load()
    a = synth.synth_actions_u8(73908, 16380, 6, 8, 997, 5, 8)
    for t in range(5):
        for e in range(6):
            for g in range(8):
                assert a[t, e, g] == L.ce_synth_action_host(73908, 16380 + e, 997 + t, g, 8)
    f = synth.synth_actions_f32(9, 0, 3, 4, 0, 2)
    for t in range(2):
        for e in range(3):
            for g in range(4):
                bits = L.ce_synth_hash_host(9, e, t, g) >> 40
                want = np.float32(np.float32(bits) * np.float32(1.0 / 16777216.0)) * np.float32(0.2) - np.float32(0.1)
                assert f[t, e, g] == want and f.dtype == np.float32


def test_video_export_nearest_upscale(tmp_path):
    """make_video_from_rgb_imgs (env_utils.py:28-58): frames are upscaled with the nearest-neighbour rule and written
    (GIF through PIL here: cv2 is absent); decoding the file gives the upscaled frames back"""
    from contracts_amd.environments.env_utils import _resize_nearest, make_video_from_rgb_imgs
    rs = np.random.RandomState(0)
    pal = np.array([[0, 0, 0], [180, 180, 180], [0, 255, 0], [99, 156, 194], [113, 75, 24], [0, 0, 255]], np.uint8)
    frames = [pal[rs.randint(len(pal), size=(25, 18))] for _ in range(4)]
    up = _resize_nearest(frames[0], (18 * 20, 25 * 20))
    assert up.shape == (500, 360, 3) and np.array_equal(up[::20, ::20], frames[0]) and np.array_equal(up[19::20, 19::20], frames[0])
    path = make_video_from_rgb_imgs(frames, str(tmp_path / "vid"), video_name="t", resize=(360, 500))
    if path.endswith(".gif"):
        from PIL import Image
        im = Image.open(path)
        assert im.n_frames == 4 and im.size == (360, 500)
        for i in range(4):
            im.seek(i)
            assert np.array_equal(np.asarray(im.convert("RGB"))[::20, ::20], frames[i])


def test_static_map_matches_the_fixture_tables():
    """ce_static_map (no GPU needed): the layout the engine builds its tables from, against the apple / spawn / waste
    point lists the reference's own envs reported when the fixtures were made"""
    import golden_check as gc
    from contracts_amd import _lib
    for kind, fixture in (("harvest_features", "feat_harvest_n2"), ("cleanup_features", "feat_cleanup_n2")):
        rows = _lib.static_map(kind)
        g = gc.load(fixture)
        apple = "A" if kind.startswith("harvest") else "B"
        cells = lambda chars: [[r, c] for r, row in enumerate(rows) for c, ch in enumerate(row) if ch in chars]
        assert cells(apple) == g["apple_points"].tolist() and cells("P") == g["spawn_points"].tolist()
        if kind.startswith("cleanup"):
            assert cells("HR") == g["waste_points"].tolist()
    assert _lib.static_map("harvest") == _lib.static_map("harvest_features")
    import pytest
    with pytest.raises(_lib.EngineError):
        _lib.static_map("selfdrive")


def test_episode_metrics_mapping_and_callback_layouts():
    """host logic without a GPU: the engine's metric rows under the reference's keys, and MetricsCallback over the
    layouts it has to serve (the reference's one-env BaseEnv, a wrapper around a base env, the batched hook's views)"""
    from contracts_amd.environments.metrics import episode_metrics
    from contracts_amd.utils.logger_utils import MetricsCallback, episode_env_metrics
    n = 3
    mi = np.arange(40)
    mf = np.arange(40) / 4.0
    m = episode_metrics("cleanup", n, mi, mf, final=True, contract=True)
    assert m["total_apples_eaten"] == 0 and m["raw_env_rewards"] == 1 and m["dirt_cleaned"] == 2 and m["transfers"] == 0.0
    assert [m["a%d-waste_cleaned" % i] for i in range(n)] == [4, 5, 6]
    assert (m["equality"], m["sustainability"], m["transfer_equality"], m["transfer_sustainability"]) == (0.25, 0.5, 0.75, 1.0)
    m = episode_metrics("harvest", n, mi, mf, final=False, inequity=True)
    assert "equality" not in m and m["transfers"] == 0 and m["raw_env_rewards"] == mf[5 + 2 * n]
    assert m["a2-close_apples_consumed"] == mi[4 + n + 2] and m["low_density_apples_eaten"] == 3
    assert set(episode_metrics("cleanup_features", 2, mi, mf, False)) == {"dirt_cleaned", "raw_env_rewards", "transfers"}
    assert episode_metrics("selfdrive", 4, None, mf, False) == {"transfers": 0.0}

    class Env:
        def __init__(self, metrics, base=None):
            self.metrics = metrics
            if base is not None:
                self.base_env = base

    class OneEnvBaseEnv:  # RLlib's wrapper around a single MultiAgentEnv, as the reference's callback expects it
        def __init__(self, env):
            self._unwrapped_env = env

    class Vector:
        def __init__(self, envs):
            self._envs = envs

        def get_sub_environments(self):
            return self._envs

    assert episode_env_metrics(OneEnvBaseEnv(Env({"x": 1}))) == {"x": 1}
    wrapped = Env({"accepted": 1}, base=Env({"transfers": 2.0, "accepted": 0}))
    assert episode_env_metrics(OneEnvBaseEnv(wrapped)) == {"transfers": 2.0, "accepted": 1}  # the wrapper's keys win
    vec = Vector([Env({"k": 0}), Env({"k": 1}), Env({"k": 2})])
    assert episode_env_metrics(vec, env_index=2) == {"k": 2}

    class Episode:
        custom_metrics = None

    ep, cb = Episode(), MetricsCallback()
    cb.on_episode_start(base_env=vec, episode=ep)
    assert ep.custom_metrics == {}
    cb.on_episode_step(base_env=vec, episode=ep)
    cb.on_episode_end(base_env=vec, episode=ep, env_index=1)
    assert ep.custom_metrics == {"k": 1}


def test_agent_view_mirrors_the_reference_agent_fields():
    from contracts_amd.environments.map_env import AgentView
    a = AgentView("a7", 3, 11, 2, 7)
    assert a.pos.tolist() == [3, 11] and a.list_pos == [3, 11] and a.orientation == "DOWN" and a.int_orientation == 2
    assert a.get_char_id() == b"8" and a.get_pos() is a.pos and a.get_orientation() == "DOWN"
    assert a.row_size == a.col_size == 7
    assert list(a.translate_pos_to_egocentric_coord([4, 9])) == [8, 5]


def test_to_base_env_maps_the_env_configuration(monkeypatch):
    """host logic of the RLlib hook, no GPU: `to_base_env(num_envs=E)` builds ONE BatchedBaseEnv whose
    engine configuration is the env's own — kind, agents, horizon, firing, reward flags, contract id and float32 bounds,
    null_prob, seed — and falls back to object-per-sub-env semantics where the batched hook does not apply.  The engine
    is replaced by a recorder (the product path itself has no CPU engine: tests/test_cabi.py::test_no_gpu_fails_loudly)."""
    import contracts_amd.engine as engine_mod
    import contracts_amd.vector_env as vector_env
    from contracts_amd import _lib

    made = []

    class RecordingEngine:
        def __init__(self, kind, num_envs, num_agents, **kw):
            self.kind, self.E, self.n, self.kw = kind, num_envs, num_agents, kw
            self.cfg = engine_mod.make_config(kind, num_envs, num_agents, **kw)
            self.firing = bool(kw.get("firing", False))
            self.num_actions = engine_mod.NUM_ACTIONS.get((kind, self.firing))
            self.seeded = None
            made.append(self)

        def seed(self, seeds=None, seed0=0, **k):
            self.seeded = seed0

        def construct(self, *a, **k):
            pass

        def reset(self, *a, **k):
            pass

        def set_contract(self, *a):
            self.contract_set = a

        def set_flags(self, **k):
            pass

        def upload(self, *a, **k):
            pass

        def download(self, field, *a, **k):
            return np.zeros((1, 628 * 2), np.uint32) if field == "rng" else np.zeros((1, 8), np.float64)

        def prefetch(self, *a, **k):
            pass

        def close(self):
            pass

    monkeypatch.setattr(vector_env, "BatchedEnv", RecordingEngine)
    import contracts_amd.environments.map_env as map_env
    import contracts_amd.environments.feature_envs as feature_envs
    import contracts_amd.environments.self_driving_car_accelerate as sdc
    for mod in (map_env, feature_envs, sdc):
        monkeypatch.setattr(mod, "BatchedEnv", RecordingEngine)
    from contracts_amd.contract.contract_list import CleanupContract, SelfdriveContractDistprop
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    from contracts_amd.environments.vector_hook import SubEnvBaseEnv

    base = CleanupEnv(num_agents=4, horizon=77, disable_firing=False, inequity_averse_reward=True, alpha=5.0, beta=0.05,
                      rng="private")
    env = SeparateContractSubgameStage(base, CleanupContract(4), 4, True, null_prob=0.25)
    base.seed(9001)
    made.clear()
    venv = env.to_base_env(make_env=None, num_envs=64, remote_envs=False)
    assert isinstance(venv, vector_env.BatchedBaseEnv) and len(made) == 1
    eng = made[0]
    assert (eng.kind, eng.E, eng.n, eng.seeded) == ("cleanup", 64, 4, 9001)
    c = eng.cfg
    assert c.horizon == 77 and c.contract == _lib.CONTRACT["cleanup"] and c.null_prob == 0.25
    assert c.contract_low == 0.0 and c.contract_high == float(np.float32(0.2))  # the Box's float32 bound
    assert c.flags & _lib.FLAG_FIRING and c.flags & _lib.FLAG_INEQUITY and not c.flags & _lib.FLAG_AUTO_RESET
    assert (c.alpha, c.beta) == (5.0, 0.05) and venv.contract == "cleanup" and venv.convolutional is True

    plain = HarvestEnv(num_agents=3, horizon=50, rng="private")
    made.clear()
    v2 = plain.to_base_env(num_envs=8)
    assert isinstance(v2, vector_env.BatchedBaseEnv) and made[0].kind == "harvest" and made[0].cfg.contract == 0
    assert not made[0].cfg.flags & _lib.FLAG_FIRING and made[0].cfg.horizon == 50
    assert not made[0].cfg.flags & _lib.FLAG_RNG_COUNTER  # the reference's stream unless the env asks otherwise

    # opt-in to the engine's counter stream for the batched hook: per env (vector_rng=) or per process (environment variable)
    made.clear()
    HarvestEnv(num_agents=3, horizon=50, rng="private", vector_rng="counter").to_base_env(num_envs=8)
    assert made[-1].E == 8 and made[-1].cfg.flags & _lib.FLAG_RNG_COUNTER
    monkeypatch.setenv("CONTRACTS_AMD_VECTOR_RNG", "counter")
    made.clear()
    CleanupEnv(num_agents=4, rng="private").to_base_env(num_envs=8)
    assert made[-1].E == 8 and made[-1].cfg.flags & _lib.FLAG_RNG_COUNTER
    monkeypatch.delenv("CONTRACTS_AMD_VECTOR_RNG")
    with pytest.raises(ValueError):
        HarvestEnv(num_agents=3, rng="private", vector_rng="xorshift")

    car = SeparateContractSubgameStage(sdc.SelfAcceleratingCarEnv(num_agents=4, collision_on=True, rng="private"),
                                       SelfdriveContractDistprop(4), 4, False)
    made.clear()
    v3 = car.to_base_env(num_envs=16)
    assert made[0].kind == "selfdrive" and made[0].cfg.flags & _lib.FLAG_COLLISION and made[0].cfg.contract_high == 100.0
    assert v3.convolutional is False

    # fallbacks: one sub-env, remote sub-envs, feature-vector grid envs -> the adapters themselves, one object per sub-env
    assert isinstance(plain.to_base_env(num_envs=1), SubEnvBaseEnv)
    built = []

    def make_env(i):
        built.append(i)
        return HarvestEnv(num_agents=3, horizon=50, rng="private")

    remote = plain.to_base_env(make_env=make_env, num_envs=3, remote_envs=True)
    assert isinstance(remote, SubEnvBaseEnv) and remote.num_envs == 3 and built == [1, 2]
    feat = CleanupEnv(num_agents=2, image_obs=False, rng="private")
    with pytest.raises(ValueError):
        feat.to_base_env(num_envs=2)  # needs make_env to build the other sub-env

    # a layout of one's own travels to the batched hook as the engine's ascii_map (the shipped one stays None)
    small = ["@@@@@@@", "@HB  P@", "@RB  P@", "@@@@@@@"]
    custom = CleanupEnv(ascii_map=small, num_agents=2, rng="private")
    assert custom.GRID_SHAPE == (4, 7) and custom.POTENTIAL_WASTE_AREA == 2 and custom.N_APPLE_CELLS == 2
    assert custom.global_observation_space["image"].shape == (4, 7, 3)
    made.clear()
    custom.to_base_env(num_envs=8)
    assert made[-1].kw["ascii_map"] == small and made[-1].cfg.map_rows == 4 and made[-1].cfg.map_cols == 7
    made.clear()
    CleanupEnv(num_agents=2, rng="private").to_base_env(num_envs=8)
    assert made[-1].kw["ascii_map"] is None and not made[-1].cfg.ascii_map

    # an unseeded env: replica 0's seed is what the global np.random would draw next, but the global stream does not move
    np.random.seed(4242)
    before = np.random.get_state()[1].copy(), np.random.get_state()[2]
    fresh = HarvestEnv(num_agents=3, horizon=50, rng="private")
    made.clear()
    fresh.to_base_env(num_envs=4)
    after = np.random.get_state()
    assert np.array_equal(after[1], before[0]) and after[2] == before[1]
    twin = np.random.RandomState(4242)
    first = int(twin.randint(0, 2 ** 31 - 1))
    assert made[-1].seeded == first
    # a second unseeded hook of the same process (train + eval vector envs, two env kinds) gets a DIFFERENT seed — the private
    # generator moves on — while the global stream still has not moved; re-seeding the script reproduces the sequence
    made.clear()
    HarvestEnv(num_agents=3, horizon=50, rng="private").to_base_env(num_envs=4)
    assert made[-1].seeded == int(twin.randint(0, 2 ** 31 - 1)) != first
    assert np.array_equal(np.random.get_state()[1], before[0]) and np.random.get_state()[2] == before[1]
    np.random.seed(4243)  # the global state changed: the private generator restarts from it
    made.clear()
    HarvestEnv(num_agents=3, horizon=50, rng="private").to_base_env(num_envs=4)
    assert made[-1].seeded == int(np.random.RandomState(4243).randint(0, 2 ** 31 - 1))

    # recycle_dicts: "auto" recycles the dictionary trees where RLlib copies every observation at once (Dict spaces: the grid
    # kinds); the Box-space feature kinds keep the recycled machinery but hand out fresh observation rows / info dictionaries
    # every tick ("fresh_obs"); the env attribute / the environment variable force a mode
    from contracts_amd.environments.feature_envs import HarvestFeatures
    assert plain.to_base_env(num_envs=4).recycle_dicts is True
    hf = HarvestFeatures(num_agents=2, horizon=50, rng="private")
    assert hf.to_base_env(num_envs=4).recycle_dicts == "fresh_obs"
    hf.vector_recycle_dicts = False
    assert hf.to_base_env(num_envs=4).recycle_dicts is False
    hf.vector_recycle_dicts = True
    assert hf.to_base_env(num_envs=4).recycle_dicts is True
    hf.vector_recycle_dicts = None
    monkeypatch.setenv("CONTRACTS_AMD_VECTOR_RECYCLE", "checked")
    assert hf.to_base_env(num_envs=4).recycle_dicts == "checked" and plain.to_base_env(num_envs=4).recycle_dicts == "checked"
    monkeypatch.setenv("CONTRACTS_AMD_VECTOR_RECYCLE", "off")
    assert plain.to_base_env(num_envs=4).recycle_dicts is False
    monkeypatch.setenv("CONTRACTS_AMD_VECTOR_RECYCLE", "sometimes")
    with pytest.raises(ValueError, match="CONTRACTS_AMD_VECTOR_RECYCLE"):
        plain.to_base_env(num_envs=4)
    monkeypatch.delenv("CONTRACTS_AMD_VECTOR_RECYCLE")
    assert car.to_base_env(num_envs=4).recycle_dicts is False  # selfdrive: key sets change with the acting cars, always rebuilt
    with pytest.raises(ValueError):
        vector_env.BatchedBaseEnv("cleanup", 4, 2, recycle_dicts="maybe")
    # np.random.seed's range: seed0 + index must stay within 32 bits on the reference's stream; the counter stream takes 64
    big = HarvestEnv(num_agents=3, horizon=50, rng="private")
    big.seed(0xffffffff - 2)
    with pytest.raises(ValueError, match="2\\*\\*32"):
        big.to_base_env(num_envs=8)
    bigc = HarvestEnv(num_agents=3, horizon=50, rng="private", vector_rng="counter")
    bigc.seed(0xffffffff - 2)
    made.clear()
    bigc.to_base_env(num_envs=8)
    assert made[-1].seeded == 0xffffffff - 2

    # a subclass of a shipped contract that overrides compute_transfer is no longer what the fused epilogue computes: the
    # wrapper falls back to the reference's host protocol (compute_transfer called every step) instead of ignoring the override
    class DoubleCleanup(CleanupContract):
        def compute_transfer(self, obs, acts, rews, params, infos=None):
            return {k: 2 * v for k, v in super().compute_transfer(obs, acts, rews, params, infos).items()}

    class RenamedCleanup(CleanupContract):  # no override: still the epilogue
        pass

    assert CleanupContract(4).fused_epilogue() == "cleanup" and RenamedCleanup(4).fused_epilogue() == "cleanup"
    assert DoubleCleanup(4).fused_epilogue() is None
    b2 = CleanupEnv(num_agents=4, rng="private")
    assert SeparateContractSubgameStage(b2, DoubleCleanup(4), 4, True)._host_contract is True
    assert SeparateContractSubgameStage(CleanupEnv(num_agents=4, rng="private"), RenamedCleanup(4), 4, True)._host_contract is False
    assert isinstance(SeparateContractSubgameStage(b2, DoubleCleanup(4), 4, True).to_base_env(make_env=lambda i: b2, num_envs=2),
                      SubEnvBaseEnv)


def test_checked_recycling_wrappers_refuse_stale_reads():
    """recycle_dicts="checked" (vector_env.py): arrays / dictionaries stamped with their generation's epoch read normally while
    the generation stands and raise StaleDictError once it has moved on, through every ordinary way of reading them"""
    from contracts_amd.vector_env import StaleDictError, _EpochArray, _EpochDict, _stamp

    class Gen:
        epoch = 3

    g = Gen()
    buf = np.arange(12.0).reshape(3, 4)
    row = _stamp(buf[1], g)
    d = _EpochDict(g, (("image", row), ("n", 5)))
    assert isinstance(row, _EpochArray) and row.base is not None and np.shares_memory(row, buf)
    assert row[2] == 6.0 and float((row + 1).sum()) == 26.0 and type(row + 1) is np.ndarray
    assert np.array_equal(np.concatenate((row, row)), np.r_[buf[1], buf[1]]) and row.copy().tolist() == [4.0, 5.0, 6.0, 7.0]
    assert d["n"] == 5 and list(d) == ["image", "n"] and d.get("image") is row and "n" in d and len(d) == 2
    import copy
    import pickle
    dc, pk = copy.deepcopy(d), pickle.loads(pickle.dumps(d))  # copies / pickles are plain values, not handles on the generation
    assert type(dc) is dict and type(dc["image"]) is np.ndarray and dc["image"].tolist() == [4.0, 5.0, 6.0, 7.0] and dc["n"] == 5
    assert type(pk) is dict and np.array_equal(pk["image"], buf[1]) and not np.shares_memory(dc["image"], buf)
    sub = row[1:]  # plain from here on: a value already read is the consumer's
    assert type(sub) is np.ndarray
    g.epoch += 1  # the generation's buffers are about to be rewritten
    buf.fill(np.nan)
    for read in (lambda: row[0], lambda: row + 1, lambda: np.concatenate((row, row)), lambda: row.copy(), lambda: row.tolist(),
                 lambda: list(row), lambda: repr(row), lambda: np.add(1.0, row), lambda: row.astype(np.float32), lambda: d["n"],
                 lambda: d.get("n"), lambda: list(d.items()), lambda: list(d), lambda: "n" in d, lambda: row.sum(),
                 lambda: np.copy(row), lambda: np.mean(row), lambda: copy.deepcopy(d), lambda: copy.deepcopy(row), lambda: pickle.dumps(d)):
        with pytest.raises(StaleDictError):
            read()
    assert np.isnan(np.asarray(row)).all()  # the one path no hook sees reads the poison, not plausible data
    assert dc["image"].tolist() == [4.0, 5.0, 6.0, 7.0]  # what was copied in time is the consumer's
    fresh = _stamp(buf[0], g)  # the generation's next hand-out is valid again
    buf[0] = 1.0
    assert fresh.sum() == 4.0


def test_profile_provenance_rules(monkeypatch):
    """contracts_amd.build.provenance(): a measurement may be attributed to a commit only when the in-tree library is a build of
    the sources that lie here and those sources were a committed state when it was built (tools/collect_profiles.sh refuses to
    write a profile set otherwise; bench.py quotes git_head / kernels_sha16 beside `roofline.traffic`)"""
    from contracts_amd import build as b
    rec = {"git_head": "a" * 40, "git_dirty_sources": False, "kernels_sha16": "k" * 16, "lib_sha16": "l" * 16, "built_at": "now"}
    monkeypatch.setattr(b, "last_build", lambda: dict(rec))
    monkeypatch.setattr(b, "needs_build", lambda: False)
    prov, why = b.provenance()
    assert why is None and prov["git_head"] == "a" * 40 and prov["kernels_sha16"] == "k" * 16
    monkeypatch.setattr(b, "needs_build", lambda: True)  # sources edited since the build
    assert b.provenance()[0] is None and "not a build of the sources" in b.provenance()[1]
    monkeypatch.setattr(b, "needs_build", lambda: False)
    monkeypatch.setattr(b, "last_build", lambda: dict(rec, git_dirty_sources=True))  # built from uncommitted kernel sources
    assert b.provenance()[0] is None and "differ from HEAD" in b.provenance()[1]
    monkeypatch.setattr(b, "last_build", lambda: dict(rec, git_head=None))
    assert b.provenance()[0] is None and "no git HEAD" in b.provenance()[1]
    # the hash names the kernel sources and the flags, nothing else
    fp = b.fingerprint()
    assert b.kernels_sha16(fp) == b.kernels_sha16(dict(fp)) and len(b.kernels_sha16(fp)) == 16
    fp2 = {"files": dict(fp["files"], **{"ce_api.hip": "0" * 16}), "flags": fp["flags"]}
    assert b.kernels_sha16(fp2) != b.kernels_sha16(fp)


def test_bench_compact_line_stays_parseable_and_small():
    """the driver parses bench.py's LAST stdout line and lost round 5's (25-29 KB -> `parsed: null`).  The
    stdout line is now bench.compact(): built here from the committed full records of earlier rounds, it must stay under 4 KB,
    be one standalone JSON object and carry the contract's keys + roofline + cpu_baseline + parity_in_run + summary."""
    import glob
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    srcs = sorted(glob.glob(os.path.join(root, "profiles", "r0[456]_bench_driver.json")))
    assert srcs
    for src in srcs:
        d = json.load(open(src))
        d["summary"] = bench.summary(d)
        line = bench.compact(d, "gpurun_out/bench_full.json")
        assert "\n" not in line and len(line) < 4096 and len(line) <= bench.COMPACT_MAX_BYTES + 256, (src, len(line))
        c = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
            assert k in c, (src, k)
        assert c["value"] == d["value"] and c["ms_per_step"] == d["ms_per_step"]  # the pair the driver cross-checks: unrounded
        assert c["config"]["workload"] == d["config"]["workload"] and "model" not in c["config"]
        r = c["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        assert abs(r["frac"] - d["roofline"]["frac"]) < 1e-4 and r["traffic"] == d["roofline"]["traffic"]
        cb = c["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
        assert set(c["summary"]) >= {"C4", "parity_all_ok"}
    # no parity leg ran -> parity_all_ok is None, not a vacuous True
    d = json.load(open(srcs[-1]))
    d.pop("parity_in_run", None)
    for row in d.get("configs", []):
        row.pop("parity_in_run", None)
    (d.get("counter_rng") or {}).pop("parity_in_run", None)
    sm = bench.summary(d)
    assert sm["parity_all_ok"] is None and sm["parity_legs_run"] == 0
    # a summary that would push the line over the cap is cut down to the headline row, never the record itself
    d["summary"] = dict(bench.summary(d), **{"pad%d" % i: "x" * 64 for i in range(64)})
    assert len(bench.compact(d, None)) < bench.COMPACT_MAX_BYTES


def test_gc_pause_restores_the_collectors_state():
    """vector_env._gc_paused (held while a tick's dictionaries are built): the collector is off inside, and afterwards exactly
    what it was before — also when the caller runs with it disabled, when pauses nest, and when the body raises"""
    import gc
    from contracts_amd.vector_env import _gc_paused
    was = gc.isenabled()
    try:
        gc.enable()
        with _gc_paused():
            assert not gc.isenabled()
            with _gc_paused():
                assert not gc.isenabled()
            assert not gc.isenabled()
        assert gc.isenabled()
        with pytest.raises(KeyError):
            with _gc_paused():
                raise KeyError("x")
        assert gc.isenabled()
        gc.disable()
        with _gc_paused():
            assert not gc.isenabled()
        assert not gc.isenabled()  # a caller that runs with the collector off keeps it off
    finally:
        (gc.enable if was else gc.disable)()
