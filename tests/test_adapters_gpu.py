"""GPU: the drop-in adapters, driven exactly like the reference's own classes
(`np.random.seed(s); env = CleanupEnv(num_agents=n); wrapper = SeparateContractSubgameStage(...);
reset(); step({agent: action})`), against the traces the reference produced (tests/golden)."""
import hashlib
import os
import pickle
import random

import numpy as np
import pytest

import golden_check as gc

pytestmark = pytest.mark.gpu


def _mt_fp():
    st = np.random.get_state()
    return (int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16))


@pytest.mark.parametrize("name,steps", [("g1_cleanup_n4", 1150), ("g3c_cleanup_n4_cleaner", 250), ("g2_harvest_n8", 300),
                                        ("g4_cleanup_n8_fire", 120), ("g6_harvest_n1_nocontract", 100),
                                        ("g9b_cleanup_n3_inequity_done", 105),
                                        ("g9b_harvest_n4_inequity_contract_done", 110)])
def test_grid_adapter_trace(name, steps):
    _grid_adapter_trace(name, steps, host_contract=False)


@pytest.mark.parametrize("name,steps", [("m1_cleanup_small_n3", 260), ("m2_cleanup_mid_n4_fire", 240), ("m3_harvest_small_n4", 200)])
def test_grid_adapter_trace_custom_layout(name, steps):
    """CleanupEnv(ascii_map=...) / HarvestEnv(ascii_map=...) on hand-made layouts, process-global np.random and all, against
    the reference's own trace on the same layout"""
    _grid_adapter_trace(name, steps, host_contract=False)


class _PayPerCleanedSquare:
    """what a user of the reference writes: a Contract subclass with its own compute_transfer (scalar transfers, split
    evenly by the wrapper).  No `engine_contract`: the wrapper must call it on the host, with the reference's arguments."""

    @staticmethod
    def make(kind, n):
        from contracts_amd.contract.contract import Contract
        from contracts_amd.spaces import Box

        class PayPerCleanedSquare(Contract):
            def __init__(self, num_agents):
                super().__init__(Box(shape=(1,), low=0, high=0.2), np.array([0.0]), num_agents)

            def compute_transfer(self, obs, acts, rews, params, infos=None):
                assert set(obs) == set(acts) == set(rews)
                return {k: -params[k][0] * infos[k]["cleaned_squares"] for k in acts}

        class ChargeSparseHarvest(Contract):
            def __init__(self, num_agents):
                super().__init__(Box(shape=(1,), low=0, high=10.0), np.array([0.0]), num_agents)

            def compute_transfer(self, obs, acts, rews, params, infos=None):
                return {k: params[k][0] if infos[k]["feature_obs"][8] < 4 and infos[k]["eaten_close_apples"] > 0 else 0
                        for k in acts}

        return PayPerCleanedSquare(n) if kind == "cleanup" else ChargeSparseHarvest(n)


@pytest.mark.parametrize("name,steps", [("g1_cleanup_n4", 1150), ("g2_harvest_n8", 300),
                                        ("g9b_harvest_n4_inequity_contract_done", 110)])
def test_user_defined_contract_on_the_host_protocol(name, steps):
    """the same reference traces, but the contract is a user's class the engine knows nothing about: the wrapper steps
    the base env on the GPU, calls compute_transfer and redistributes as two_stage_train.py:62-121 does — rewards,
    theta draws (global np.random order), metrics incl. transfers / transfer_equality / transfer_sustainability"""
    _grid_adapter_trace(name, steps, host_contract=True)


def _grid_adapter_trace(name, steps, host_contract):
    from contracts_amd.contract import contract_list as cl
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    g = gc.load(name)
    kind, n, seed = str(g["kind"]), int(g["n"]), int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    cls = CleanupEnv if kind == "cleanup" else HarvestEnv
    extra = {}
    if "inequity" in g and int(g["inequity"]):  # float env rewards: raw_env_rewards / equality / sustainability come from float accumulators
        extra = dict(inequity_averse_reward=True, alpha=float(g["alpha"]), beta=float(g["beta"]))
    if int(g["horizon"]) != 1000:
        extra["horizon"] = int(g["horizon"])
    if "ascii_map" in g:  # m_* fixtures: the reference on a hand-made layout (the constructors' first argument)
        extra["ascii_map"] = [str(r) for r in g["ascii_map"]]
    env = cls(num_agents=n, disable_firing=not bool(int(g["firing"])), **extra)
    assert _mt_fp() == tuple(int(x) for x in g["ctor_mt"])  # the constructor consumed the global stream
    contract = bool(int(g["contract"]))
    if contract:
        con = cl.CleanupContract(n) if kind == "cleanup" else cl.HarvestFeaturemodLocalContract(n)
        if host_contract:
            con = _PayPerCleanedSquare.make(kind, n)
            assert con.engine_contract is None
        top = SeparateContractSubgameStage(env, con, n, True)
        assert top._host_contract == host_contract
    else:
        top = env
    keys = ["a%d" % i for i in range(n)]
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    t = 0
    for ep in range(len(g["ep_start"])):
        o = top.reset()
        assert _mt_fp() == tuple(int(x) for x in g["reset_mt"][ep])
        for i, k in enumerate(keys):
            assert o[k]["image"].dtype == np.float64
            assert np.array_equal(o[k]["image"], g["reset_obs"][ep][i] / 255)
            if contract:
                assert np.array_equal(o[k]["contract"], np.array([g["theta"][ep], 0.0]))
        for t in range(ep_start[ep], min(ep_start[ep + 1], steps)):
            acts = {k: int(g["actions"][t][i]) for i, k in enumerate(keys)}
            o, r, d, info = top.step(acts)
            assert set(d.keys()) == {"__all__", "a0", "a1"} and d["__all__"] == bool(g["done"][t])
            for i, k in enumerate(keys):
                if t < len(g["obs"]):
                    assert np.array_equal(o[k]["image"], g["obs"][t][i] / 255), (t, k)
                assert abs(float(r[k]) - g["rew"][t][i]) < 1e-9, (t, k)
                assert info[k]["eaten_apples"] == g["eaten"][t][i]
                second = "cleaned_squares" if kind == "cleanup" else "eaten_close_apples"
                assert info[k][second] == (g["cleaned"] if kind == "cleanup" else g["eaten_close"])[t][i]
                assert np.array_equal(info[k]["feature_obs"], g["feature_obs"][t][i])
                if contract:
                    assert info[k]["contract_param"][0] == g["theta"][ep]
            assert _mt_fp() == tuple(int(x) for x in g["mt"][t]), "global np.random diverged at step %d" % t
        if steps < ep_start[ep + 1]:
            break  # truncated replay: later episodes of the fixture depend on the full first one
        if True:
            mk = str(g["metrics_keys_ep%d" % ep]).split(",")
            assert set(mk) == set(env.metrics.keys()), (set(mk) ^ set(env.metrics.keys()))
            for k, v in zip(mk, g["metrics_vals_ep%d" % ep]):
                assert abs(float(env.metrics[k]) - v) < 1e-9 * max(1.0, abs(v)), k
    env.close()


@pytest.mark.parametrize("name", ["g5_selfdrive_n4", "g5_selfdrive_n4_collision", "g5_selfdrive_n3_collision"])
def test_selfdrive_adapter_trace(name):
    from contracts_amd.contract import contract_list as cl
    from contracts_amd.environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    g = gc.load(name)
    n, seed = int(g["n"]), int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    env = SelfAcceleratingCarEnv(num_agents=n, collision_on=bool(int(g["collision_on"])))
    top = SeparateContractSubgameStage(env, cl.SelfdriveContractDistprop(n), n, False)
    keys = ["a%d" % i for i in range(n)]
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    for ep in range(len(g["ep_start"])):
        o = top.reset()
        for i, k in enumerate(keys):
            np.testing.assert_allclose(o[k], g["reset_obs"][ep][i], rtol=0, atol=1e-9)
        for t in range(ep_start[ep], ep_start[ep + 1]):
            act = g["active"][t].astype(bool)
            acts = {k: np.array([float(g["actions"][t][i])]) for i, k in enumerate(keys) if act[i]}
            o, r, d, info = top.step(acts)
            assert list(o.keys()) == [k for i, k in enumerate(keys) if act[i]]
            for i, k in enumerate(keys):
                if act[i]:
                    np.testing.assert_allclose(o[k], g["obs"][t][i], rtol=0, atol=1e-9)
                    assert abs(r[k] - g["rew"][t][i]) < 1e-6
                    assert info[k]["just_passed"] == bool(g["just_passed"][t][i])
                assert d[k] == bool(g["done"][t][i])
            # the infos the reference attaches to the first acting key (every other key: zeros)
            first = next(iter(acts))
            assert info[first]["ambulance_rank"] == g["amb_rank"][t] and info[first]["is_crashed"] == g["is_crashed"][t]
            assert abs(info[first]["ambulance_dist_to_front"] - g["amb_dtf"][t]) < 1e-9
            assert all(info[k]["ambulance_rank"] == 0.0 and info[k]["is_crashed"] == 0 for k in acts if k != first)
            assert d["__all__"] == bool(g["done"][t][n])
            assert abs(env.metrics["transfers"] - g["transfers_metric"][t]) < 1e-6
        with pytest.raises(AttributeError):  # the reference raises when stepped after __all__ (…accelerate.py:160)
            top.step({"a0": np.array([0.0])})
    env.close()


def test_adapter_pickle_roundtrip_and_options():
    from contracts_amd.environments.cleanup_new import CleanupEnv
    np.random.seed(5)
    env = CleanupEnv(num_agents=3, one_hot_id=True)
    o = env.reset()
    assert set(o["a1"].keys()) == {"image", "features"} and np.array_equal(o["a1"]["features"], [0, 1, 0])
    rs = np.random.RandomState(0)
    for _ in range(10):
        env.step({k: int(rs.randint(8)) for k in ("a0", "a1", "a2")})
    st = np.random.get_state()
    clone = pickle.loads(pickle.dumps(env))  # shipped inside env_config (ray_config_utils.py:198-202)
    acts = [{k: int(rs.randint(8)) for k in ("a0", "a1", "a2")} for _ in range(15)]
    outs = []
    for e in (env, clone):
        np.random.set_state(st)
        outs.append([e.step(a) for a in acts])
    for (o1, r1, d1, i1), (o2, r2, d2, i2) in zip(*outs):
        assert r1 == r2 and d1 == d2
        for k in o1:
            assert np.array_equal(o1[k]["image"], o2[k]["image"])
            assert np.array_equal(i1[k]["feature_obs"], i2[k]["feature_obs"])
    with pytest.raises(KeyError):
        env.step({"a0": 11, "a1": 0, "a2": 0})
    # feature-vector mode (image_obs=False): obs is the 12+n feature vector, also at reset
    np.random.seed(6)
    fenv = CleanupEnv(num_agents=2, image_obs=False)
    fo = fenv.reset()
    assert fo["a0"].shape == (14,) and fo["a0"][11] == 56.0 and fenv.observation_space.shape == (14,)
    fo2, _, _, info = fenv.step({"a0": 4, "a1": 4})
    assert np.array_equal(fo2["a1"], info["a1"]["feature_obs"])
    env.close()
    clone.close()
    fenv.close()


def test_reference_config_file_drives_the_engine():
    """a job list in the reference's experiment_configs format builds HIP-backed envs (config plumbing)"""
    from contracts_amd.utils.config import build_env, expand_config_list
    cfg = [{"num_workers": 1, "horizon": 1000, "env_args": {"image_obs": True}, "solver": True},
           {"num_agents": [2]},
           {"environment": "cleanup_new", "contract": "CleanupContract"},
           {"environment": "harvest_new", "contract": "HarvestFeaturemodLocalContract", "separate": True},
           {"environment": "selfdrive", "contract": "SelfdriveContractDistprop", "env_args": {}}]
    jobs = expand_config_list(cfg)
    assert len(jobs) == 3
    # the other first-stage wirings of ray_config_utils.py:156-208 and the second stage of `solver: true` (test.json)
    import sys
    sys.path.insert(0, gc.GOLDEN_DIR)
    from ref_harness import StubPPOTrainer
    from contracts_amd.environments import two_stage_train as tst
    from contracts_amd.utils.config import build_second_stage
    top, base, contract = build_env(dict(jobs[0], joint=True, env_args={"image_obs": True, "concatenated_obs": True}))
    assert isinstance(top, tst.JointEnv) and top.reset()["a0"]["image"].shape == (15, 15, 6)
    base.close()
    top, base, contract = build_env(dict(jobs[0], combined=True))
    assert isinstance(top, tst.SeparateContractCombinedStage) and set(top.reset()) == {"a0", "a1"}
    base.close()
    top, base, contract = build_env(jobs[0])
    solver = build_second_stage(dict(jobs[0], solver_samples=5), base, contract, ["ckpt"], {"n_act": 8, "seed": 3}, StubPPOTrainer)
    assert isinstance(solver, tst.NegotiationSolver) and solver.num_samples == 50 and solver.decision_rule == "majority"
    o = solver.reset()
    assert 0.0 <= float(solver.contract_param[0]) <= float(contract.contract_space.high[0]) and o["a1"]["contract"][0] == solver.contract_param[0]
    neg = build_second_stage(dict(jobs[0], solver=False), base, contract, ["ckpt"], {"n_act": 8, "seed": 3}, StubPPOTrainer)
    assert isinstance(neg, tst.SeparateContractNegotiateStage) and neg.reset()["a0"]["contract"][-1] == 2
    base.close()
    for job in jobs:
        np.random.seed(job["seed"])
        random.seed(job["seed"])
        top, base, contract = build_env(job)
        o = top.reset()
        assert set(o.keys()) == {"a0", "a1"}
        if job["environment"] == "selfdrive":
            o, r, d, _ = top.step({k: np.array([0.05]) for k in o})
            assert o["a0"].shape == (2 * 2 + 5 + 2,) and r["a0"] == -100.0
        else:
            o, r, d, info = top.step({k: 4 for k in o})
            assert o["a0"]["image"].shape == (15, 15, 3)
            assert ("contract" in o["a0"]) == (not job.get("separate"))
        base.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["joint_cleanup_n3_global", "joint_cleanup_n3_concat", "joint_harvest_n2_global"])
def test_joint_env_trace(name):
    """JointEnv (centralised agent over the pixel envs; global-map and concatenated-view observations) against
    fixtures produced by the reference's own JointEnv (tests/golden/make_joint_golden.py)"""
    import hashlib
    import random
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.two_stage_train import JointEnv
    g = np.load("%s/%s.npz" % (gc.GOLDEN_DIR, name))
    kind, n, seed, mode = str(g["kind"]), int(g["n"]), int(g["seed"]), str(g["mode"])
    np.random.seed(seed)
    random.seed(seed)
    base = (CleanupEnv if kind == "cleanup" else HarvestEnv)(num_agents=n)
    env = JointEnv(base, num_agents=n, global_obs=mode == "global", concatenated_obs=mode == "concat")
    assert tuple(env.observation_space["image"].shape) == tuple(g["obs_space_shape"])
    assert list(env.action_space.nvec) == list(g["act_nvec"])
    o = env.reset()
    assert np.array_equal(o["a0"]["image"], g["reset_obs"] / 255)
    for t in range(len(g["actions"])):
        o, r, d, info = env.step({"a0": g["actions"][t]})
        img = o["a0"]["image"]
        u = np.rint(img * 255).astype(np.uint8)
        assert np.array_equal(u / 255, img)
        if t < len(g["obs"]):
            assert np.array_equal(u, g["obs"][t]), t
        assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(u).tobytes()).digest(), np.uint8), g["obs_sha"][t]), t
        assert float(r["a0"]) == g["rew"][t] and d["__all__"] == bool(g["done"][t]) and d["a0"] == d["__all__"]
        keys = sorted(info["a0"].keys())
        assert ",".join(keys) == str(g["info_keys"][t])
        vals = np.concatenate([np.atleast_1d(np.asarray(info["a0"][k], np.float64)).ravel() for k in keys])
        assert np.array_equal(vals, g["info_vals"][t]), t
        st = np.random.get_state()
        assert (int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16)) == tuple(int(x) for x in g["mt"][t])
    base.close()


def _fps():
    st, ps = np.random.get_state(), random.getstate()[1]
    return [int(st[2]), int(hashlib.sha256(st[1].tobytes()).hexdigest()[:8], 16),
            int(ps[624]), int(hashlib.sha256(np.array(ps[:624], np.uint32).tobytes()).hexdigest()[:8], 16)]


def _check_stage_obs(o, keys, sha_ref, contract_ref, tag):
    img = np.stack([np.rint(np.asarray(o[k]["image"]) * 255).astype(np.uint8) for k in keys])
    assert np.array_equal(np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8), sha_ref), "image " + tag
    assert np.array_equal(np.stack([np.asarray(o[k]["contract"], np.float64) for k in keys]), contract_ref), "contract obs " + tag


@pytest.mark.gpu
def test_combined_stage_trace():
    """SeparateContractCombinedStage (propose / accept / play in one action space) against the reference's own trace"""
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.two_stage_train import SeparateContractCombinedStage
    g = np.load("%s/stage_combined_cleanup_n4.npz" % gc.GOLDEN_DIR)
    n, seed, horizon = int(g["n"]), int(g["seed"]), int(g["horizon"])
    np.random.seed(seed)
    random.seed(seed)
    base = CleanupEnv(num_agents=n, horizon=horizon)
    env = SeparateContractCombinedStage(base, contract_list.CleanupContract(n), n, True)
    assert np.array_equal(env.action_space.low, g["action_low"]) and np.array_equal(env.action_space.high, g["action_high"])
    keys = ["a%d" % i for i in range(n)]
    o = env.reset()
    _check_stage_obs(o, keys, g["reset_sha"], g["reset_contract"], "reset")
    for t in range(len(g["actions"])):
        o, r, d, info = env.step({k: g["actions"][t][i] for i, k in enumerate(keys)})
        _check_stage_obs(o, keys, g["obs_sha"][t], g["contract_obs"][t], "step %d" % t)
        np.testing.assert_allclose([float(r[k]) for k in keys], g["rew"][t], rtol=0, atol=1e-9)
        assert d["__all__"] == bool(g["done"][t]) and _fps() == list(g["fp"][t]), t
        if d["__all__"]:
            env.reset()
    base.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stage_negotiate_cleanup_n4", "stage_negotiate_cleanup_n2"])
def test_negotiate_stage_trace(name):
    """SeparateContractNegotiateStage: the subgame is rolled to termination inside one outer step with frozen policies
    (the harness's deterministic StubPPOTrainer on both sides)"""
    import sys
    sys.path.insert(0, gc.GOLDEN_DIR)
    from ref_harness import StubPPOTrainer
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.two_stage_train import SeparateContractNegotiateStage
    g = np.load("%s/%s.npz" % (gc.GOLDEN_DIR, name))
    n, seed, horizon = int(g["n"]), int(g["seed"]), int(g["horizon"])
    np.random.seed(seed)
    random.seed(seed)
    base = CleanupEnv(num_agents=n, horizon=horizon)
    env = SeparateContractNegotiateStage(base, contract_list.CleanupContract(n), n, horizon, {"n_act": 8, "seed": seed + 7},
                                         "stub-env", "stub-path", True, False, trainer_factory=StubPPOTrainer)
    assert np.array_equal(env.action_space.low, g["action_low"]) and np.array_equal(env.action_space.high, g["action_high"])
    keys = ["a%d" % i for i in range(n)]
    t = 0
    for ep in range(len(g["accepted"])):
        env.reset()
        for stage in range(2):
            o, r, d, info = env.step({k: g["actions"][t][i] for i, k in enumerate(keys)})
            _check_stage_obs(o, keys, g["obs_sha"][t], g["contract_obs"][t], "ep %d stage %d" % (ep, stage))
            np.testing.assert_allclose([float(r[k]) for k in keys], g["rew"][t], rtol=0, atol=1e-9)
            assert d["__all__"] == bool(g["done"][t]) and _fps() == list(g["fp"][t]), (ep, stage)
            t += 1
        assert int(env.metrics["accepted"]) == int(g["accepted"][ep])
        assert np.array_equal(np.asarray(env.metrics["contract"], np.float64), g["proposed"][ep])
        assert len(env.frozen_trainer.calls) == int(g["trainer_calls"][ep])
    base.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stage_solver_majority_cleanup_n4", "stage_solver_max_cleanup_n3"])
def test_negotiation_solver_trace(name):
    """NegotiationSolver: the contract is searched at reset over the frozen value heads (the harness's stub on both
    sides), then the episode runs under it — chosen parameter, observations, transferred rewards and both RNG streams"""
    import sys
    sys.path.insert(0, gc.GOLDEN_DIR)
    from ref_harness import StubPPOTrainer
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.two_stage_train import NegotiationSolver
    g = np.load("%s/%s.npz" % (gc.GOLDEN_DIR, name))
    n, seed, horizon, steps = int(g["n"]), int(g["seed"]), int(g["horizon"]), int(g["steps"])
    np.random.seed(seed)
    random.seed(seed)
    base = CleanupEnv(num_agents=n, horizon=horizon)
    env = NegotiationSolver(base, contract_list.CleanupContract(n), n, horizon, {"n_act": 8, "seed": seed + 7}, "stub-env",
                            "stub-path", True, False, contract_samples=int(g["samples"]), decision_rule=str(g["rule"]),
                            trainer_factory=StubPPOTrainer)
    env.contract_param_space.seed(seed + 11)  # as the fixture's generator did (a gym space samples from its own np_random)
    keys = ["a%d" % i for i in range(n)]
    t = 0
    for ep in range(len(g["chosen"])):
        o = env.reset()
        assert np.array_equal(np.asarray(env.contract_param, np.float64), g["chosen"][ep]), ep
        _check_stage_obs(o, keys, g["reset_sha"][ep], g["reset_contract"][ep], "ep %d reset" % ep)
        for s in range(steps):
            o, r, d, info = env.step({k: int(g["actions"][t][i]) for i, k in enumerate(keys)})
            _check_stage_obs(o, keys, g["obs_sha"][t], g["contract_obs"][t], "ep %d step %d" % (ep, s))
            np.testing.assert_allclose([float(r[k]) for k in keys], g["rew"][t], rtol=0, atol=1e-6)
            assert d["__all__"] == bool(g["done"][t]) and _fps() == list(g["fp"][t]), (ep, s)
            t += 1
    base.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["render_cleanup_n5_firing", "render_harvest_n6_firing"])
def test_render_trace(name):
    """what run_render.py:41 collects: env.full_map_to_colors() after every step — map, agents and the step's beams"""
    import hashlib
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    g = np.load("%s/%s.npz" % (gc.GOLDEN_DIR, name))
    n, seed = int(g["n"]), int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    env = (CleanupEnv if str(g["kind"]) == "cleanup" else HarvestEnv)(num_agents=n, disable_firing=False, horizon=int(g["horizon"]))
    env.reset()
    first = env.full_map_to_colors()
    assert str(first.dtype) == str(g["rgb_dtype"]) and np.array_equal(first, g["reset_rgb"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(env.render(mode="rgb_array").astype(np.uint8).tobytes()).digest(), np.uint8),
                          g["render_rgb_sha"])
    for t in range(len(g["actions"])):
        _, _, d, _ = env.step({"a%d" % i: int(g["actions"][t][i]) for i in range(n)})
        img = env.full_map_to_colors().astype(np.uint8)
        if t < len(g["rgb"]):
            assert np.array_equal(img, g["rgb"][t]), t
        assert np.array_equal(np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8), g["rgb_sha"][t]), t
        assert d["__all__"] == bool(g["done"][t])
        if d["__all__"]:
            env.reset()
    env.close()


@pytest.mark.gpu
def test_batched_contract_evaluation_matches_sequential_episodes():
    """run_solver's loop (one NegotiationSolver-style episode per sampled contract, run_solver.py:18-73) against the
    batched sweep: K contracts as K replicas of one handle — identical episode rewards, double for double"""
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.two_stage_train import SeparateContractEnv
    from contracts_amd.run_solver import evaluate_contracts
    n, horizon, K = 4, 60, 6
    thetas = np.array([0.0, 0.03, 0.08, 0.12, 0.17, float(np.float32(0.2))])
    seeds = np.arange(K, dtype=np.uint64) * 7907 + 99

    def act(k, t, a):
        return int((k * 131 + t * 31 + a * 7 + (t * t) % 5) % 8)

    class FixedContract(SeparateContractEnv):  # the episode part of NegotiationSolver with a given parameter
        def __init__(self, base, contract, n, theta):
            super().__init__(base, contract, n, True)
            self._external_theta(True)
            self._theta_value = np.array([theta])

        def reset(self):
            obs = self.base_env.reset()
            self._set_theta(self._theta_value)
            self.params = {"a%d" % i: self._theta_value for i in range(self.num_agents)}
            return self._with_contract(obs, list(self.params))

    want = []
    for k in range(K):
        np.random.seed(int(seeds[k]))
        random.seed(int(seeds[k]))
        base = CleanupEnv(num_agents=n, horizon=horizon)
        env = FixedContract(base, contract_list.CleanupContract(n), n, thetas[k])
        obs = env.reset()
        ep, d, t = 0, {"__all__": False}, 0
        while not d["__all__"] and t < horizon:
            obs, r, d, info = env.step({"a%d" % a: act(k, t, a) for a in range(n)})
            for key in obs:
                ep += r[key]
            t += 1
        want.append(ep)
        base.close()
    got = evaluate_contracts("cleanup", n, "CleanupContract", thetas, seeds,
                             lambda obs, th, t: np.array([[act(k, t, a) for a in range(n)] for k in range(K)]), horizon=horizon)
    assert got["steps"] == horizon
    assert np.array_equal(got["ep_rewards"], np.array(want, np.float64)), (got["ep_rewards"], want)
    assert got["mean reward"] == float(np.mean(want)) and got["std contract"] == float(np.std(thetas))
    ar = got["agent_rewards"]  # transfers are zero-sum over the agents: the parameter shows per agent, not in the total
    assert np.array_equal(ar[0], np.rint(ar[0])) and np.abs(ar[1:] - np.rint(ar[1:])).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("rng", ["global", "private"])
def test_user_contract_with_proportional_recipients(rng):
    """the (value, {recipient: proportion}) form of a transfer (two_stage_train.py:73-81), a null_prob, and both RNG
    modes of the adapter: checked against a twin env under a contract that transfers nothing (same theta draws, so the
    same stream) and the redistribution formula written out with numpy"""
    from contracts_amd.contract.contract import Contract
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    from contracts_amd.spaces import Box
    n, horizon = 3, 40
    shares = np.array([[0.0, 0.75, 0.25], [0.5, 0.0, 0.5], [1.0, 0.0, 0.0]])

    class ApplesTax(Contract):
        def __init__(self, num_agents, rate):
            super().__init__(Box(shape=(1,), low=0.5, high=2.0), np.array([0.0]), num_agents)
            self.rate = rate

        def compute_transfer(self, obs, acts, rews, params, infos=None):
            return {k: (self.rate * params[k][0] * infos[k]["eaten_apples"],
                        {"a%d" % j: shares[int(k[1:]), j] for j in range(n) if shares[int(k[1:]), j] > 0}) for k in acts}

    streams = {}

    def on(which, fn, *args):  # two envs on the one process-global stream: each call runs on its env's own copy
        if rng == "global":
            np.random.set_state(streams[which])
        out = fn(*args)
        if rng == "global":
            streams[which] = np.random.get_state()
        return out

    tops = {}
    for which, rate in (("twin", 0.0), ("taxed", 1.0)):
        np.random.seed(77)
        streams[which] = np.random.get_state()
        env = on(which, lambda: HarvestEnv(num_agents=n, horizon=horizon, rng=rng))
        if rng == "private":
            env.seed(1234)
        tops[which] = SeparateContractSubgameStage(env, ApplesTax(n, rate), n, True, null_prob=0.5)
    twin, taxed = tops["twin"], tops["taxed"]
    keys = ["a%d" % i for i in range(n)]
    rs = np.random.RandomState(5)
    thetas = []
    for ep in range(6):
        bo, o = on("twin", twin.reset), on("taxed", taxed.reset)
        theta = taxed.params["a0"]
        thetas.append(float(theta[0]))
        assert np.array_equal(theta, twin.params["a0"]) and 0.5 <= theta[0] <= 2.0
        assert np.array_equal(o["a1"]["contract"], [theta[0], 0.0]) and np.array_equal(o["a2"]["image"], bo["a2"]["image"])
        total = 0.0
        for t in range(horizon):
            acts = {k: int(rs.randint(8)) for k in keys}
            bo, br, bd, bi = on("twin", twin.step, acts)
            o, r, d, info = on("taxed", taxed.step, acts)
            eaten = np.array([bi[k]["eaten_apples"] for k in keys], np.float64)
            amount = theta[0] * eaten
            want = np.array([br[k] for k in keys], np.float64) - amount + amount @ shares
            assert np.allclose([r[k] for k in keys], want, rtol=0, atol=1e-12), (ep, t)
            assert all(np.array_equal(o[k]["image"], bo[k]["image"]) for k in keys) and d == bd
            assert info["a1"]["contract_param"] is taxed.params["a1"]
            total += amount.sum()
            assert abs(taxed.base_env.metrics["transfers"] - total) < 1e-9 and twin.base_env.metrics["transfers"] == 0
        assert d["__all__"] and total > 0
        m, mt = taxed.base_env.metrics, twin.base_env.metrics
        assert m["equality"] == mt["equality"] and mt["transfer_equality"] == mt["equality"]
        assert mt["transfer_sustainability"] == mt["sustainability"] and m["transfer_equality"] != m["equality"]
    assert any(th == 0.5 for th in thetas) and any(th > 0.5 for th in thetas)  # null_prob = 0.5: both branches drawn
    if rng == "global":
        assert np.array_equal(streams["twin"][1], streams["taxed"][1])
    twin.base_env.close()
    taxed.base_env.close()


@pytest.mark.gpu
def test_batched_selfdrive_contract_evaluation_matches_sequential_episodes():
    """the selfdrive leg of the batched sweep: cars that are done stop acting, replicas finish at different steps"""
    from contracts_amd.contract import contract_list
    from contracts_amd.environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
    from contracts_amd.environments.two_stage_train import SeparateContractEnv
    from contracts_amd.run_solver import evaluate_contracts
    n, K, horizon = 4, 7, 400
    thetas = np.array([0.0, 5.0, 12.5, 33.0, 50.0, 77.7, 100.0])
    seeds = np.arange(K, dtype=np.uint64) * 104729 + 5

    def act(k, t, a):
        return np.float32(((k * 37 + t * 11 + a * 5) % 21 - 8) / 100.0)

    class FixedContract(SeparateContractEnv):
        def __init__(self, base, contract, n, theta):
            super().__init__(base, contract, n, False)
            self._external_theta(True)
            self._theta_value = np.array([theta])

        def reset(self):
            obs = self.base_env.reset()
            self._set_theta(self._theta_value)
            self.params = {"a%d" % i: self._theta_value for i in range(self.num_agents)}
            return self._with_contract(obs, list(self.params))

    want, want_steps = [], []
    for k in range(K):
        np.random.seed(int(seeds[k]))
        random.seed(int(seeds[k]))
        base = SelfAcceleratingCarEnv(num_agents=n)
        env = FixedContract(base, contract_list.SelfdriveContractDistprop(n), n, thetas[k])
        obs = env.reset()
        active = ["a%d" % a for a in range(n)]
        ep, d, t = 0, {"__all__": False}, 0
        while not d["__all__"] and t < horizon:
            obs, r, d, info = env.step({key: np.array([act(k, t, int(key[1:]))]) for key in active})
            for key in d:
                if d[key] and key in active:
                    active.remove(key)
            for key in obs:
                ep += r[key]
            t += 1
        want.append(ep)
        want_steps.append(t)
        base.close()
    assert len(set(want_steps)) > 1  # the replicas really finish at different steps
    got = evaluate_contracts("selfdrive", n, "SelfdriveContractDistprop", thetas, seeds,
                             lambda obs, th, t: np.array([[act(k, t, a) for a in range(n)] for k in range(K)], np.float32),
                             horizon=horizon)
    assert got["steps"] == max(want_steps)
    np.testing.assert_allclose(got["ep_rewards"], np.array(want, np.float64), rtol=0, atol=1e-9)


def test_harvest_features_image_obs_trace():
    """HarvestFeatures(image_obs=True) against the reference's trace: crops of the incrementally painted colour map
    (incl. the steps where a cell two agents shared has turned black under the one that stayed), the whole map's hash,
    the feature rows in the infos; plus a pickle round trip in mid-episode"""
    from contracts_amd.environments.feature_envs import CleanupFeatures, HarvestFeatures
    g = gc.load("featimg_harvest_n6")
    n, seed = int(g["n"]), int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    env = HarvestFeatures(num_agents=n, horizon=int(g["horizon"]), image_obs=True)
    assert env.observation_space.shape == tuple(g["obs_space_shape"]) and env.observation_space.dtype == np.uint8
    assert np.array_equal(env.world_map_color, g["ctor_world"])
    keys = ["a%d" % i for i in range(n)]
    ep_start = list(g["ep_start"]) + [len(g["actions"])]
    quirk = 0
    for ep in range(len(g["ep_start"])):
        o = env.reset()
        assert np.array_equal(np.stack([o[k] for k in keys]), g["reset_obs"][ep])
        for t in range(ep_start[ep], ep_start[ep + 1]):
            if t == 100:
                env = pickle.loads(pickle.dumps(env))
            o, r, d, info = env.step({k: int(g["actions"][t][i]) for i, k in enumerate(keys)})
            got = np.stack([o[k] for k in keys])
            assert got.dtype == np.uint8 and np.array_equal(got, g["obs"][t]), t
            assert int(hashlib.sha256(env.world_map_color.tobytes()).hexdigest()[:8], 16) == int(g["world"][t]), t
            assert np.array_equal(np.stack([info[k]["feature_obs"] for k in keys]), g["feature_obs"][t])
            assert [float(r[k]) for k in keys] == list(g["rew"][t]) and d["__all__"] == bool(g["done"][t])
            quirk += int((got[:, 7, 7] != np.array([159, 67, 255])).any())
    assert quirk > 0
    env.close()
    c = CleanupFeatures(num_agents=2, image_obs=True)  # the reference stores the flag and never reads it
    assert c.image_obs and c.reset()["a0"].shape == (14,)
    c.close()


def test_shipped_ascii_map_and_layouts_within_the_caps_are_accepted():
    from contracts_amd.environments import cleanup_new, harvest_new
    assert len(cleanup_new.CLEANUP_MAP) == 25 and len(harvest_new.HARVEST_MAP[0]) == 38
    first = []
    for kw in (dict(ascii_map=cleanup_new.CLEANUP_MAP), {}):  # the former is what the reference's default argument does
        np.random.seed(3)
        env = cleanup_new.CleanupEnv(num_agents=2, **kw)
        first.append(env.reset())
        env.close()
    assert all(np.array_equal(first[0][k]["image"], first[1][k]["image"]) for k in ("a0", "a1"))
    # a layout of one's own (here: the shipped one minus an apple cell) goes to the engine as ce_config.ascii_map ...
    other = list(harvest_new.HARVEST_MAP)
    other[3] = other[3].replace("A", " ", 1)
    env = harvest_new.HarvestEnv(ascii_map=other, num_agents=2)
    env.reset()
    assert len(env.apple_points) == 154 and env.world_map.shape == (16, 38) and env.N_APPLE_CELLS == 154
    assert sorted(map(tuple, env.current_apple_points)) == sorted(map(tuple, env.apple_points))  # harvest starts with every apple
    env.close()
    small = ["@@@@@@@", "@HB  P@", "@RB  P@", "@@@@@@@"]
    env = cleanup_new.CleanupEnv(ascii_map=small, num_agents=2)
    o = env.reset()
    assert env.world_map.shape == (4, 7) and env.potential_waste_area == 2 and o["a0"]["image"].shape == (15, 15, 3)
    assert env.full_map_to_colors().shape == (4, 7, 3) and env.global_observation_space["image"].shape == (4, 7, 3)
    assert sorted(map(tuple, env.waste_points)) == [(1, 1), (2, 1)] and sorted(map(tuple, env.spawn_points)) == [(1, 5), (1, 5), (2, 5), (2, 5)]
    assert sorted(map(tuple, env.apple_points)) == [(1, 2), (2, 2)] and env.get_map_with_agents().shape == (4, 7)
    env.step({"a0": 7, "a1": 4})
    env.close()
    # ... and one that breaks a rule of the header is refused by ce_create, which names the rule
    from contracts_amd._lib import EngineError
    with pytest.raises(EngineError, match="walled in"):
        cleanup_new.CleanupEnv(ascii_map=["@@@@@@@", "@HB  P ", "@RB  P@", "@@@@@@@"], num_agents=1)
    with pytest.raises(EngineError, match="more agents than spawn points"):
        cleanup_new.CleanupEnv(ascii_map=small, num_agents=3)


@pytest.mark.parametrize("name", ["inspect_cleanup_n4", "inspect_harvest_n5"])
def test_inspection_helpers_trace(name):
    """the read-only helpers users of the reference poke at between steps — get_map_with_agents, env.agents[...],
    find_visible_agents, current apple / waste lists, compute_permitted_area, the shuffled spawn / waste lists,
    color_view — against the reference's values after reset and after every step"""
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    g = gc.load(name)
    kind, n, seed = str(g["kind"]), int(g["n"]), int(g["seed"])
    np.random.seed(seed)
    random.seed(seed)
    env = (CleanupEnv if kind == "cleanup" else HarvestEnv)(num_agents=n, disable_firing=False, horizon=int(g["horizon"]))
    keys = ["a%d" % i for i in range(n)]
    assert env.apple_points == g["apple_points"].tolist()
    env.reset()
    orient = {"UP": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3}

    def unpad(rows):
        return [list(map(int, r)) for r in rows if r[0] >= 0]

    for t in range(len(g["actions"])):
        if t > 0:
            _, _, d, _ = env.step({k: int(g["actions"][t][i]) for i, k in enumerate(keys)})
            assert d["__all__"] == bool(g["done"][t])
        tag = "t=%d" % t
        assert np.array_equal(env.get_map_with_agents().view(np.uint8).reshape(env.GRID_SHAPE), g["map"][t]), tag
        agents = env.agents
        assert [[int(agents[k].pos[0]), int(agents[k].pos[1]), orient[agents[k].orientation]] for k in keys] == g["poses"][t].tolist()
        assert [agents[k].get_char_id()[0] for k in keys] == g["char_ids"][t].tolist()
        assert agents["a1"].get_orientation() == agents["a1"].orientation and agents["a0"].row_size == 7
        assert np.array_equal(np.stack([env.find_visible_agents(k) for k in keys]), g["visible"][t]), tag
        env.compute_current_apples()
        assert env.current_apple_points == unpad(g["apples"][t]), tag
        assert env.spawn_points == g["spawn_points"][t].tolist(), tag
        if kind == "cleanup":
            env.compute_current_wastes()
            assert env.current_waste_points == unpad(g["wastes"][t]), tag
            assert env.compute_permitted_area() == int(g["permitted"][t])
            assert env.waste_points == g["waste_points"][t].tolist(), tag
            env.compute_probabilities()
            assert (env.current_waste_spawn_prob == 0) == (1 - int(g["permitted"][t]) / 119 >= 0.4)
        else:
            loc = [int(agents["a0"].pos[0]), int(agents["a0"].pos[1])]
            want = sum(1 for p in env.current_apple_points if (p[0] - loc[0]) ** 2 + (p[1] - loc[1]) ** 2 <= 5)
            assert env.count_apples_in_radius(5, loc) == want
        assert np.array_equal(np.stack([env.color_view(agents[k]) for k in keys]), g["views"][t]), tag
        assert env.test_if_in_bounds([0, 0]) and not env.test_if_in_bounds([env.GRID_SHAPE[0], 0])
        if t > 0 and g["done"][t]:
            env.reset()
    env.close()


def test_run_rendering_writes_episode_videos(tmp_path):
    """run_render.py's loop over the drop-in env: frames from device state, one file per rendered episode"""
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.run_render import run_rendering
    np.random.seed(3)
    env = CleanupEnv(num_agents=3, horizon=12, disable_firing=False)
    rs = np.random.RandomState(1)
    paths = run_rendering(env, lambda ob: int(rs.randint(9)), str(tmp_path / "renders"), num_renders=2)
    assert len(paths) == 2 and all(os.path.getsize(p) > 1000 for p in paths)
    if paths[0].endswith(".gif"):
        from PIL import Image
        im = Image.open(paths[0])
        assert im.n_frames == 12 and im.size == (18 * 20, 25 * 20)
    env.close()


def test_versioned_state_rejects_other_layouts(tmp_path):
    """checkpoints and pickled adapters carry the engine ABI version and every parameter that enters a step: a blob from
    another layout / another configuration is refused with a clear message instead of an opaque size error or a silently
    different continuation"""
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    from contracts_amd.environments.cleanup_new import CleanupEnv
    a = BatchedEnv("cleanup", 4, 3, contract="cleanup", horizon=30)
    a.seed(seed0=1)
    a.reset()
    st = a.state_dict()
    assert int(st["_meta"][6]) == _lib.CE_ABI_VERSION and "_meta_f64" in st
    b = BatchedEnv("cleanup", 4, 3, contract="cleanup", horizon=30)
    b.load_state_dict(st)  # same configuration: fine
    old = dict(st)
    old["_meta"] = st["_meta"][:6]  # what round 2 wrote
    del old["_meta_f64"]
    with pytest.raises(ValueError, match="predates the versioned format"):
        b.load_state_dict(old)
    other_abi = dict(st)
    other_abi["_meta"] = st["_meta"].copy()
    other_abi["_meta"][6] += 1
    with pytest.raises(ValueError, match="ABI v"):
        b.load_state_dict(other_abi)
    c = BatchedEnv("cleanup", 4, 3, contract="cleanup", horizon=30, contract_high=0.1)
    with pytest.raises(ValueError, match="contract_high"):
        c.load_state_dict(st)
    d = BatchedEnv("cleanup", 4, 3, contract="cleanup", horizon=30, null_prob=0.25)
    with pytest.raises(ValueError, match="null_prob"):
        d.load_state_dict(st)
    for e in (a, b, c, d):
        e.close()
    # a cached download is read-only: mutating it must raise, not corrupt later downloads
    a = BatchedEnv("cleanup", 2, 2)
    a.seed(seed0=3)
    a.reset()
    a.prefetch(("agents", "timestep"))
    with pytest.raises(ValueError):
        a.download("agents")[0, 0, 0] = 9
    a.close()
    # pickled adapter blobs
    np.random.seed(2)
    env = CleanupEnv(num_agents=2)
    env.reset()
    clone = pickle.loads(pickle.dumps(env))
    assert clone._pending_abi == _lib.CE_ABI_VERSION
    clone._pending_abi = _lib.CE_ABI_VERSION + 1
    with pytest.raises(ValueError, match="re-create the env"):
        clone.step({"a0": 4, "a1": 4})
    env.close()
