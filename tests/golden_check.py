"""Replays a golden fixture (tests/golden/*.npz, produced by the reference itself) through an
implementation exposing the Oracle/BatchedEnv protocol (seed/reset/step + state arrays) and
asserts step-by-step identity.  Shared by the CPU-oracle tests and the GPU parity tests."""
import glob
import hashlib
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def grid_kwargs(g):
    kind = str(g["kind"])
    contract = None
    if int(g["contract"]):
        contract = "cleanup" if kind == "cleanup" else "harvest_local"
    kw = dict(contract=contract, horizon=int(g["horizon"]), firing=bool(int(g["firing"])))
    if "collective" in g and int(g["collective"]):
        kw["collective"] = True
    if "inequity" in g and int(g["inequity"]):
        kw.update(inequity=True, alpha=float(g["alpha"]), beta=float(g["beta"]))
    if "ascii_map" in g:  # m_* fixtures: the reference built on a custom layout
        kw["ascii_map"] = [str(r) for r in g["ascii_map"]]
    return kind, int(g["n"]), kw


METRIC_INT = {"total_apples_eaten": 0, "raw_env_rewards": 1, "dirt_cleaned": 2, "low_density_apples_eaten": 3}
METRIC_F64 = {"transfers": 0, "equality": 1, "sustainability": 2, "transfer_equality": 3, "transfer_sustainability": 4}


def metrics_dict(kind, n, mi, mf, final, float_base=False):
    """reference-style env.metrics dict from the engine's metric arrays (float_base: inequity aversion — the env's rewards
    are floats and the reference sums those: CE_MF_RAW_ENV_REWARDS_F)"""
    out = {"total_apples_eaten": mi[0], "raw_env_rewards": mf[5 + 2 * n] if float_base else mi[1], "transfers": mf[0]}
    if kind == "cleanup":
        out["dirt_cleaned"] = mi[2]
        for i in range(n):
            out["a%d-waste_cleaned" % i] = mi[4 + i]
    else:
        out["low_density_apples_eaten"] = mi[3]
        for i in range(n):
            out["a%d-apples_consumed" % i] = mi[4 + i]
            out["a%d-close_apples_consumed" % i] = mi[4 + n + i]
    if final:
        out["equality"], out["sustainability"] = mf[1], mf[2]
        out["transfer_equality"], out["transfer_sustainability"] = mf[3], mf[4]
    return out


def replay_grid(g, impl, env=0, check_obs=True, sync=lambda: None, get=None):
    """impl: object with seed/reset/step and array attributes; `get(name)` returns a host
    array for a field (defaults to attribute access, i.e. oracle host views)."""
    if get is None:
        def get(name):
            return getattr(impl, name)
    kind, n = str(g["kind"]), int(g["n"])
    seed = int(g["seed"])
    E = impl.E
    static_spawn = g["static_spawn"]
    image_obs = bool(int(g["image_obs"])) if "image_obs" in g else True
    float_base = "inequity" in g and bool(int(g["inequity"]))
    check_obs = check_obs and image_obs

    def spawn_cells():
        sp = get("spawn_perm")[env][: len(g["ctor_spawn_perm"])]  # (the buffer is 20 entries wide; a custom layout uses fewer)
        return [int(p) % len(static_spawn) if kind == "cleanup" else int(p) for p in sp]

    counter = "rng_mode" in g and str(g["rng_mode"]) == "counter"  # p_* fixtures: the reference run over the counter stream

    def rng_record():
        """what the fixture recorded after every operation: MT19937 position + key fingerprint, or — counter mode — the
        (key0, key1, generation) row"""
        rng = get("rng")[env]
        if counter:
            return tuple(int(x) for x in rng[:3])
        return (int(rng[624]), int(hashlib.sha256(rng[:624].tobytes()).hexdigest()[:8], 16))

    seeds = np.full((E,), seed, np.uint64)
    impl.seed(seeds)
    sync()
    if counter:
        assert rng_record() == tuple(int(x) for x in g["ctor_mt"]), "generator state after the constructor"
    assert np.array_equal(get("agents")[env][:, :3], g["ctor_agents"]), "constructor agents"
    assert spawn_cells() == list(g["ctor_spawn_perm"]), "constructor spawn list"
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    n_obs = len(g["obs"])
    for ep in range(len(g["ep_start"])):
        impl.reset()
        sync()
        assert np.array_equal(get("grid")[env], g["reset_grid"][ep]), "reset grid ep%d" % ep
        assert np.array_equal(get("agents")[env][:, :3], g["reset_agents"][ep]), "reset agents ep%d" % ep
        if check_obs:
            assert np.array_equal(get("obs")[env], g["reset_obs"][ep]), "reset obs ep%d" % ep
        if not image_obs:  # feature-vector mode: the reset observation is the feature vector
            assert np.array_equal(get("features")[env].astype(np.float64), g["reset_features"][ep]), "reset features"
        assert rng_record() == tuple(int(x) for x in g["reset_mt"][ep]), "generator state after reset ep%d" % ep
        assert get("theta")[env] == g["theta"][ep], "theta ep%d" % ep
        for t in range(ep_start[ep], ep_start[ep + 1]):
            acts = np.broadcast_to(g["actions"][t], (E, n))
            impl.step(acts)
            sync()
            tag = "step %d (ep %d)" % (t, ep)
            assert np.array_equal(get("agents")[env][:, :3], g["agents"][t]), "agents " + tag
            assert np.array_equal(get("grid")[env], g["grid"][t]), "grid " + tag
            if not float_base:
                assert np.array_equal(get("base_reward")[env], g["base_rew"][t]), "base reward " + tag
            np.testing.assert_allclose(get("reward")[env], g["rew"][t], rtol=0, atol=1e-9, err_msg="reward " + tag)
            info = get("info")[env]
            assert np.array_equal(info[:, 0], g["eaten"][t]), "eaten_apples " + tag
            second = g["cleaned"][t] if kind == "cleanup" else g["eaten_close"][t]
            assert np.array_equal(info[:, 1], second), "info[1] " + tag
            assert np.array_equal(get("features")[env].astype(np.float64), g["feature_obs"][t]), "feature_obs " + tag
            assert int(get("done")[env]) == int(g["done"][t]), "done " + tag
            if check_obs:
                ob = get("obs")[env]
                if t < n_obs:
                    assert np.array_equal(ob, g["obs"][t]), "obs " + tag
                sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(ob).tobytes()).digest(), np.uint8)
                assert np.array_equal(sha, g["obs_sha"][t]), "obs sha " + tag
            assert rng_record() == tuple(int(x) for x in g["mt"][t]), "generator state " + tag
            if kind == "cleanup":
                nw = g["waste_perm"].shape[1]  # (119 for the shipped layout; the buffer is that wide for every layout)
                assert np.array_equal(get("waste_perm")[env][:nw], g["waste_perm"][t]), "waste perm " + tag
        # metrics at the end of the recorded episode
        keys = str(g["metrics_keys_ep%d" % ep]).split(",")
        vals = g["metrics_vals_ep%d" % ep]
        final = "equality" in keys
        mi = get("final_int_metrics" if final else "int_metrics")[env]
        mf = get("final_f64_metrics" if final else "f64_metrics")[env]
        md = metrics_dict(kind, n, mi, mf, final, float_base)
        for k, v in zip(keys, vals):
            if k.startswith("transfer") and int(g["contract"]) == 0:
                continue
            np.testing.assert_allclose(md[k], v, rtol=1e-12, atol=1e-9, err_msg="metric %s ep%d" % (k, ep))
        assert spawn_cells() == list(g["spawn_perm"][ep]), "spawn list after ep%d" % ep


def replay_selfdrive(g, impl, env=0, sync=lambda: None, get=None, atol=1e-9):
    if get is None:
        def get(name):
            return getattr(impl, name)
    n = int(g["n"])
    E = impl.E
    impl.seed(np.full((E,), int(g["seed"]), np.uint64))
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    for ep in range(len(g["ep_start"])):
        impl.reset()
        sync()
        np.testing.assert_allclose(get("obs_f64")[env], g["reset_obs"][ep], rtol=0, atol=atol, err_msg="reset obs")
        np.testing.assert_allclose(get("theta")[env], g["theta"][ep], rtol=0, atol=atol)
        for t in range(ep_start[ep], ep_start[ep + 1]):
            impl.step(np.broadcast_to(g["actions"][t], (E, n)), np.broadcast_to(g["active"][t], (E, n)))
            sync()
            tag = "step %d (ep %d)" % (t, ep)
            act = g["active"][t].astype(bool)
            np.testing.assert_allclose(get("obs_f64")[env][act], g["obs"][t][act], rtol=0, atol=atol, err_msg="obs " + tag)
            np.testing.assert_allclose(get("reward")[env][act], g["rew"][t][act], rtol=0, atol=1e-6, err_msg="rew " + tag)
            st = get("sd_state")[env]
            np.testing.assert_allclose(st[:n], g["pos"][t], rtol=0, atol=atol, err_msg="pos " + tag)
            np.testing.assert_allclose(st[n:2 * n], g["vel"][t], rtol=0, atol=atol, err_msg="vel " + tag)
            assert np.array_equal(get("done_agents")[env], g["done"][t][:n]), "done " + tag
            assert int(get("done")[env]) == int(g["done"][t][n]), "done_all " + tag
            assert np.array_equal(get("info")[env][:, 0][act], g["just_passed"][t][act]), "just_passed " + tag
            np.testing.assert_allclose(get("f64_metrics")[env][0], g["transfers_metric"][t], rtol=0, atol=1e-6,
                                       err_msg="transfers metric " + tag)
            nc = int(st[4 * n + 1])
            crossed = [int(x) for x in st[4 * n + 2:4 * n + 2 + nc]]
            assert crossed == [int(x) for x in g["crossed"][t] if x >= 0], "crossed " + tag
            if "dist_to_front" in g:  # update_infos' bookkeeping and the infos of the first acting key
                np.testing.assert_allclose(st[2 * n:3 * n], g["dist_to_front"][t], rtol=0, atol=atol, err_msg="dist_to_front " + tag)
                si = get("sd_info")[env]
                assert si[0] == g["amb_rank"][t], "ambulance_rank %s: %s != %s" % (tag, si[0], g["amb_rank"][t])
                np.testing.assert_allclose(si[1], g["amb_dtf"][t], rtol=0, atol=atol, err_msg="ambulance_dist_to_front " + tag)
                first = int(np.nonzero(g["active"][t])[0][0])
                assert int(get("info")[env][first, 1]) == int(g["is_crashed"][t]), "is_crashed " + tag


def feat_kwargs(g):
    kind = str(g["kind"])
    contract = None
    if int(g["contract"]):
        contract = "harvest_local" if kind == "harvest_features" else "cleanup"
    return kind, int(g["n"]), dict(contract=contract, horizon=int(g["horizon"]))


def _order_from_stamps(stamps, width):
    """cell indices sorted by list-order stamp (0xffff = absent), padded with -1 like the fixture"""
    present = np.nonzero(stamps != 0xffff)[0]
    order = present[np.argsort(stamps[present], kind="stable")]
    out = np.full((width,), -1, np.int16)
    out[: len(order)] = order
    return out


def replay_feat(g, impl, env=0, sync=lambda: None, get=None):
    """HarvestFeatures / CleanupFeatures fixture (tests/golden/feat_*.npz) through an Oracle/BatchedEnv-like impl"""
    if get is None:
        def get(name):
            return getattr(impl, name)
    kind, n = str(g["kind"]), int(g["n"])
    E = impl.E
    NA, NW = len(g["apple_points"]), len(g["waste_points"])

    def fp(block):
        rng = get("rng")[env]
        w = rng[628 * block: 628 * block + 625]
        return (int(w[624]), int(hashlib.sha256(w[:624].tobytes()).hexdigest()[:8], 16))

    def check_lists(tag, apple_ref, waste_ref):
        assert np.array_equal(_order_from_stamps(get("apple_stamp")[env][:NA], NA), apple_ref), "apple list " + tag
        if NW:
            assert np.array_equal(_order_from_stamps(get("waste_stamp")[env][:NW], NW), waste_ref), "waste list " + tag

    impl.seed(np.full((E,), int(g["seed"]), np.uint64))
    sync()
    assert np.array_equal(get("agents")[env][:, :3], g["ctor_agents"]), "constructor agents"
    assert fp(0) == tuple(int(x) for x in g["ctor_mt_np"]) and fp(1) == tuple(int(x) for x in g["ctor_mt_py"]), "constructor RNG"
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    for ep in range(len(g["ep_start"])):
        impl.reset()
        sync()
        tag = "reset ep%d" % ep
        assert np.array_equal(get("agents")[env][:, :3], g["reset_agents"][ep]), "agents " + tag
        assert np.array_equal(get("features")[env].astype(np.float64), g["reset_obs"][ep]), "obs " + tag
        check_lists(tag, g["reset_apple_order"][ep], g["reset_waste_order"][ep] if NW else None)
        assert fp(0) == tuple(int(x) for x in g["reset_mt_np"][ep]), "numpy stream " + tag
        assert fp(1) == tuple(int(x) for x in g["reset_mt_py"][ep]), "python stream " + tag
        assert get("theta")[env] == g["theta"][ep], "theta " + tag
        for t in range(ep_start[ep], ep_start[ep + 1]):
            impl.step(np.broadcast_to(g["actions"][t], (E, n)))
            sync()
            tag = "step %d (ep %d)" % (t, ep)
            assert np.array_equal(get("agents")[env][:, :3], g["agents"][t]), "agents " + tag
            assert np.array_equal(get("base_reward")[env], g["base_rew"][t]), "base reward " + tag
            np.testing.assert_allclose(get("reward")[env], g["rew"][t], rtol=0, atol=1e-9, err_msg="reward " + tag)
            info = get("info")[env]
            assert np.array_equal(info[:, 0], g["info0"][t]) and np.array_equal(info[:, 1], g["info1"][t]), "infos " + tag
            assert np.array_equal(get("features")[env].astype(np.float64), g["feature_obs"][t]), "feature obs " + tag
            assert int(get("done")[env]) == int(g["done"][t]), "done " + tag
            check_lists(tag, g["apple_order"][t], g["waste_order"][t] if NW else None)
            assert fp(1) == tuple(int(x) for x in g["mt_py"][t]), "python stream " + tag
            assert fp(0) == tuple(int(x) for x in g["mt_np"][t]), "numpy stream " + tag
        keys = str(g["metrics_keys_ep%d" % ep]).split(",")
        vals = g["metrics_vals_ep%d" % ep]
        final = "equality" in keys
        mi = get("final_int_metrics" if final else "int_metrics")[env]
        mf = get("final_f64_metrics" if final else "f64_metrics")[env]
        md = {"total_apples_eaten": mi[0], "raw_env_rewards": mi[1], "dirt_cleaned": mi[2], "low_density_apples_eaten": mi[3],
              "transfers": mf[0], "equality": mf[1], "sustainability": mf[2], "transfer_equality": mf[3],
              "transfer_sustainability": mf[4]}
        for k, v in zip(keys, vals):
            if k.startswith("transfer") and int(g["contract"]) == 0:
                continue
            np.testing.assert_allclose(md[k], v, rtol=1e-12, atol=1e-9, err_msg="metric %s ep%d" % (k, ep))


# colours of the reference's render (map_env.py:24-42, cleanup_new.py:42-47): cell codes, agents '1'..'9', beams F / C
RENDER_CELL_RGB = np.array([[0, 0, 0], [180, 180, 180], [0, 255, 0], [99, 156, 194], [113, 75, 24], [113, 75, 24]], np.uint8)
RENDER_AGENT_RGB = np.array([[0, 0, 255], [2, 81, 154], [204, 0, 204], [216, 30, 54], [254, 151, 0], [100, 255, 255],
                             [99, 99, 255], [250, 204, 255], [238, 223, 16]], np.uint8)
RENDER_BEAM_RGB = {1: (255, 255, 0), 2: (100, 255, 255)}


def compose_render(grid, agents, beam):
    """full_map_to_colors (map_env.py:354-392): map, agents in agent order, beams on top"""
    rgb = RENDER_CELL_RGB[grid].copy()
    for i, a in enumerate(agents):
        rgb[a[0], a[1]] = RENDER_AGENT_RGB[i]
    for code, colour in RENDER_BEAM_RGB.items():
        rgb[beam == code] = colour
    return rgb


def replay_render(g, impl, env=0, sync=lambda: None, get=None):
    """render_* fixtures: firing enabled, beam trace on; every step's beam cells and composed colour map"""
    if get is None:
        def get(name):
            return getattr(impl, name)
    n, E = int(g["n"]), impl.E
    impl.seed(np.full((E,), int(g["seed"]), np.uint64))
    impl.reset()
    sync()
    assert not get("beam_map")[env].any(), "beams after reset"
    assert np.array_equal(compose_render(get("grid")[env], get("agents")[env], get("beam_map")[env]), g["reset_rgb"])
    for t in range(len(g["actions"])):
        impl.step(np.broadcast_to(g["actions"][t], (E, n)))
        sync()
        beam = get("beam_map")[env]
        assert np.array_equal(beam, g["beam"][t]), "beam cells step %d" % t
        img = compose_render(get("grid")[env], get("agents")[env], beam)
        if t < len(g["rgb"]):
            assert np.array_equal(img, g["rgb"][t]), "render step %d" % t
        sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), np.uint8)
        assert np.array_equal(sha, g["rgb_sha"][t]), "render sha step %d" % t
        rng = get("rng")[env]
        fp = int(hashlib.sha256(rng[:624].tobytes()).hexdigest()[:8], 16)
        assert (int(rng[624]), fp) == tuple(int(x) for x in g["mt"][t]), "MT state step %d" % t
        assert int(get("done")[env]) == int(g["done"][t])
        if int(g["done"][t]):
            impl.reset()
            sync()
            assert not get("beam_map")[env].any(), "beams after reset"
