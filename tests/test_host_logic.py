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
    with pytest.raises(NotImplementedError):
        c.compute_transfer({}, {}, {}, {}, {})


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
