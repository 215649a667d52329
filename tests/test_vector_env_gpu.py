"""RLlib BaseEnv-shaped vector hook: poll / send_actions / try_reset over one engine handle,
checked against the CPU oracle stepping the same seeds, including done -> try_reset cycles."""
import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract,horizon,rng", [
    ("cleanup", 4, "cleanup", 25, "mt19937"), ("harvest", 3, None, 30, "mt19937"), ("harvest_features", 2, "harvest_local", 20, "mt19937"),
    ("cleanup", 4, "cleanup", 25, "counter"), ("harvest", 5, "harvest_local", 30, "counter")])  # the engine's own stream: vs its oracle twin
def test_base_env_protocol_matches_oracle(kind, n, contract, horizon, rng):
    from contracts_amd.vector_env import BatchedBaseEnv
    from oracle.pyoracle import Oracle
    E, T = 6, 70
    venv = BatchedBaseEnv(kind, E, n, contract=contract, seed0=500, horizon=horizon, rng=rng)
    orc = Oracle(kind, E, n, contract=contract, horizon=horizon, rng=rng)
    orc.seed(seed0=500)
    orc.reset()
    keys = ["a%d" % i for i in range(n)]
    grid = kind in ("cleanup", "harvest")

    def check_obs(ob, e):
        for i, k in enumerate(keys):
            if grid:
                assert np.array_equal(ob[k]["image"], orc.obs[e][i] / 255)
                if contract:
                    assert np.array_equal(ob[k]["contract"], [orc.theta[e], 0.0])
            else:
                f = orc.features[e][i].astype(np.float64)
                assert np.array_equal(ob[k], np.concatenate((f, [orc.theta[e], 0.0])) if contract else f)

    obs, rew, dones, infos, _ = venv.poll()
    assert sorted(obs.keys()) == list(range(E))
    for e in range(E):
        check_obs(obs[e], e)
    rs = np.random.RandomState(9)
    na = venv.engine.num_actions
    resets = 0
    for t in range(T):
        a = rs.randint(na, size=(E, n))
        venv.send_actions({e: {k: int(a[e, i]) for i, k in enumerate(keys)} for e in range(E)})
        orc.step(a.astype(np.uint8))
        obs, rew, dones, infos, _ = venv.poll()
        for e in range(E):
            check_obs(obs[e], e)
            for i, k in enumerate(keys):
                np.testing.assert_allclose(rew[e][k], orc.reward[e][i] if contract else orc.base_reward[e][i], rtol=0, atol=1e-9)
            assert dones[e]["__all__"] == bool(orc.done[e])
            if kind != "cleanup_features":
                assert infos[e]["a0"]["eaten_apples"] == orc.info[e][0][0]
        done_ids = [e for e in range(E) if dones[e]["__all__"]]
        if done_ids:
            mask = np.zeros((E,), np.uint8)
            mask[done_ids] = 1
            orc.reset(mask)
            for e in done_ids:
                ob = venv.try_reset(e)
                check_obs(ob[e], e)
                resets += 1
    assert resets >= E  # every sub-env went through at least one done -> try_reset cycle
    venv.stop()
    orc.close()


@pytest.mark.gpu
def test_base_env_protocol_selfdrive_matches_oracle():
    """selfdrive through the vector hook: agents that are done stop acting (their keys leave the dicts), infos carry the
    ambulance stats on the first acting key, done -> try_reset cycles"""
    from contracts_amd.vector_env import BatchedBaseEnv
    from oracle.pyoracle import Oracle
    E, n, T = 7, 4, 260
    venv = BatchedBaseEnv("selfdrive", E, n, contract="selfdrive_distprop", seed0=900)
    orc = Oracle("selfdrive", E, n, contract="selfdrive_distprop")
    orc.seed(seed0=900)
    orc.reset()
    keys = ["a%d" % i for i in range(n)]
    obs, rew, dones, infos, _ = venv.poll()
    for e in range(E):
        for i, k in enumerate(keys):
            np.testing.assert_allclose(obs[e][k], orc.obs_f64[e][i], rtol=0, atol=1e-9)
    rs = np.random.RandomState(2)
    agent_done = np.zeros((E, n), bool)
    resets = 0
    for t in range(T):
        a = rs.uniform(-0.15, 0.15, size=(E, n)).astype(np.float32)
        act = (~agent_done).astype(np.uint8)
        venv.send_actions({e: {k: np.array([a[e, i]]) for i, k in enumerate(keys) if act[e, i]} for e in range(E)})
        orc.step(a, act)
        obs, rew, dones, infos, _ = venv.poll()
        for e in range(E):
            acting = [k for i, k in enumerate(keys) if act[e, i]]
            assert list(obs[e].keys()) == acting and list(rew[e].keys()) == acting
            for i, k in enumerate(keys):
                if act[e, i]:
                    np.testing.assert_allclose(obs[e][k], orc.obs_f64[e][i], rtol=0, atol=1e-9)
                    assert abs(rew[e][k] - orc.reward[e][i]) < 1e-9
                    assert infos[e][k]["just_passed"] == bool(orc.info[e][i][0])
                assert dones[e][k] == bool(orc.done_agents[e][i])
            first = acting[0]
            assert infos[e][first]["ambulance_rank"] == orc.sd_info[e][0]
            assert abs(infos[e][first]["ambulance_dist_to_front"] - orc.sd_info[e][1]) < 1e-9
            assert dones[e]["__all__"] == bool(orc.done[e])
            agent_done[e] = orc.done_agents[e].astype(bool)
        done_ids = [e for e in range(E) if dones[e]["__all__"]]
        if done_ids:
            mask = np.zeros((E,), np.uint8)
            mask[done_ids] = 1
            orc.reset(mask)
            for e in done_ids:
                ob = venv.try_reset(e)
                for i, k in enumerate(keys):
                    np.testing.assert_allclose(ob[e][k], orc.obs_f64[e][i], rtol=0, atol=1e-9)
                agent_done[e] = False
                resets += 1
    assert resets >= E
    venv.stop()
    orc.close()


@pytest.mark.gpu
def test_vector_hook_scales_to_the_headline_batch():
    """E = 16 384 sub-envs (cleanup n = 8 + contract) through the dict protocol, every env's observation / reward / done /
    info dictionary in the caller's hands each tick, and a synchronized horizon where all E envs are reset through per-env
    try_reset calls: the recycled dictionaries (the default) sustain >= 1 M env-steps/s on an idle box, and the reset storm
    must cost O(E) (one masked launch + one copy for the whole batch)."""
    import time
    from contracts_amd.vector_env import BatchedBaseEnv
    E, n, horizon = 16384, 8, 6
    venv = BatchedBaseEnv("cleanup", E, n, contract="cleanup", horizon=horizon)
    keys = ["a%d" % i for i in range(n)]
    obs, _, _, _, _ = venv.poll()
    assert len(obs) == E
    rs = np.random.RandomState(0)
    acts = rs.randint(8, size=(horizon, E, n))
    for t in range(horizon):
        if t == 2:  # the first two ticks build the two dictionary generations; the steady state is what is timed
            t0 = time.perf_counter()
        venv.send_actions({e: dict(zip(keys, acts[t, e].tolist())) for e in range(E)})
        obs, rew, dones, infos, _ = venv.poll()
        touched = 0
        for e, ob in obs.items():  # what RLlib's sampler does: walk every env of the tick
            touched += ob["a0"]["image"].shape[0] + len(rew[e]) + len(infos[e]) + int(dones[e]["__all__"])
    dt = time.perf_counter() - t0
    rate = (horizon - 2) * E / dt
    assert all(dones[e]["__all__"] for e in range(E))  # the synchronized horizon
    launches_before = venv.engine.download("timestep").max()
    t1 = time.perf_counter()
    for e in range(E):
        ob = venv.try_reset(e)
        assert ob[e]["a3"]["image"].shape == (15, 15, 3)
    dt_reset = time.perf_counter() - t1
    assert venv.engine.download("timestep").max() == 0 and launches_before == horizon
    print("vector hook: %.0f env-steps/s through the dict protocol; reset storm of %d envs %.3f s" % (rate, E, dt_reset))
    assert rate >= 150000, rate  # a gate against per-entry Python work creeping back in (the action dictionaries of this loop
    # are themselves built in Python, ~15 ms per tick), not a benchmark: bench.py's boundary section reports the rate
    assert dt_reset < 1.0, dt_reset  # O(E): a few hundred lazily built observation dictionaries per 10 ms, not E launches
    # the tensor path: no Python containers at all
    t2 = time.perf_counter()
    for t in range(20):
        venv.send_actions_array(acts[t % horizon].astype(np.uint8))
        tens = venv.poll_tensors()
    venv.engine.synchronize()
    rate_t = 20 * E / (time.perf_counter() - t2)
    assert tens["obs"].shape == (E, n, 15, 15, 3) and rate_t > 10 * rate
    venv.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,contract", [("cleanup", "cleanup"), ("harvest_features", "harvest_local"), ("selfdrive", "selfdrive_distprop")])
def test_episode_metrics_reach_the_callback(kind, contract):
    """MetricsCallback.on_episode_end on the batched hook: the finished episode's metrics of the env RLlib names by
    env_index — equal to what a single-env adapter with the same private seed reports — also after the batched reset
    that the first try_reset of the tick triggers for every done env"""
    from contracts_amd.contract import contract_list as cl
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.feature_envs import HarvestFeatures
    from contracts_amd.environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    from contracts_amd.utils.logger_utils import MetricsCallback
    from contracts_amd.vector_env import BatchedBaseEnv
    E, n, horizon, seed0 = 5, 3, 12, 4242
    kw = {} if kind == "selfdrive" else {"horizon": horizon}
    venv = BatchedBaseEnv(kind, E, n, contract=contract, seed0=seed0, convolutional=kind == "cleanup", **kw)
    keys = ["a%d" % i for i in range(n)]

    def action(e, t, i):
        return np.array([0.05 + 0.01 * ((e + t + i) % 5)]) if kind == "selfdrive" else (e * 7 + t * 3 + i) % 7

    class Episode:
        custom_metrics = None

    # the same episodes through single-env adapters on private streams with the same seeds
    want = []
    for e in range(E):
        if kind == "cleanup":
            base, con = CleanupEnv(num_agents=n, horizon=horizon, rng="private"), cl.CleanupContract(n)
        elif kind == "harvest_features":
            base, con = HarvestFeatures(num_agents=n, horizon=horizon, rng="private"), cl.HarvestFeaturemodLocalContract(n)
        else:
            base, con = SelfAcceleratingCarEnv(num_agents=n, rng="private"), cl.SelfdriveContractDistprop(n)
        base._ensure_engine().seed(np.array([seed0 + e], np.uint64))  # replays the constructor like the batched API
        top = SeparateContractSubgameStage(base, con, n, kind == "cleanup")
        top.reset()
        alive, t, d = list(keys), 0, {"__all__": False}
        while not d["__all__"]:
            _, _, d, _ = top.step({k: action(e, t, int(k[1:])) for k in alive})
            alive = [k for k in alive if not d.get(k, False)] if kind == "selfdrive" else alive
            t += 1
        want.append((t, dict(base.metrics)))
        base.close()

    cb, got, t = MetricsCallback(), {}, 0
    venv.poll()
    alive = {e: list(keys) for e in range(E)}
    finished = set()
    while len(got) < E:
        venv.send_actions({e: {k: action(e, t, int(k[1:])) for k in alive[e]} if e not in finished else {} for e in range(E)}
                          if kind == "selfdrive" else {e: {k: action(e, t, i) for i, k in enumerate(keys)} for e in range(E)})
        obs, rew, dones, infos, _ = venv.poll()
        t += 1
        done_now = [e for e in range(E) if dones[e]["__all__"] and e not in got]
        for j, e in enumerate(done_now):
            assert t == want[e][0]
            if j == 1 and kind != "selfdrive":
                venv.try_reset(done_now[0])  # resets every done env of the tick; the others' final metrics must survive
            ep = Episode()
            cb.on_episode_start(base_env=venv, episode=ep)
            cb.on_episode_end(base_env=venv, episode=ep, env_index=e)
            got[e] = ep.custom_metrics
            finished.add(e)
        if kind == "selfdrive":
            for e in range(E):
                alive[e] = [k for k in alive[e] if not dones[e].get(k, False)]
            if len(got) < E and finished:
                break  # selfdrive replicas end at different ticks and a done env cannot idle in send_actions
    for e, m in got.items():
        assert set(m) == set(want[e][1]), (set(m) ^ set(want[e][1]))
        for k, v in want[e][1].items():
            assert abs(float(m[k]) - float(v)) < 1e-9, (e, k, m[k], v)
    assert got and (kind == "selfdrive" or len(got) == E)
    assert len(venv.get_sub_environments()) == E and venv.get_sub_environments()[-1].env_id == E - 1
    venv.stop()


def _sampler_loop(base_env, policy, ticks):
    """what RLlib's sampler (`_env_runner`) does with a BaseEnv: poll -> per-episode bookkeeping -> policy -> send_actions,
    and on a done episode try_reset + continue from the reset observation.  Returns (episodes, agent_steps, transcript)."""
    episodes, agent_steps, transcript = 0, 0, []
    for _ in range(ticks):
        obs, rew, dones, infos, off = base_env.poll()
        actions = {}
        for env_id in obs:
            o = obs[env_id]
            if dones[env_id]["__all__"]:
                episodes += 1
                o = base_env.try_reset(env_id)[env_id]  # RLlib resets the sub-env and feeds the reset obs to the policy
            actions[env_id] = {k: policy(env_id, k, o[k]) for k in o}
            agent_steps += len(o)
            transcript.append((env_id, {k: float(v) for k, v in rew[env_id].items()}, dones[env_id]["__all__"]))
        base_env.send_actions(actions)
    return episodes, agent_steps, transcript


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cleanup_contract", "harvest_plain", "harvest_features_contract", "selfdrive_contract"])
def test_to_base_env_returns_the_batched_hook(which):
    """`env.to_base_env(num_envs=E)` — the call RLlib's RolloutWorker makes on a MultiAgentEnv with
    num_envs_per_worker = E — returns ONE BatchedBaseEnv configured like the env (kind, agents, horizon, flags, contract
    bounds); a sampler loop over it issues exactly one step launch per tick (ce_timing_* counts launches) and its
    transcript equals the oracle stepping the same seeds.  ray is absent here: the sampler loop above stands in for
    `_env_runner` and the class derives from ray's BaseEnv only when ray is importable."""
    from contracts_amd.contract.contract_list import CleanupContract, HarvestFeaturemodLocalContract, SelfdriveContractDistprop
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.feature_envs import HarvestFeatures
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
    from contracts_amd.environments.two_stage_train import SeparateContractSubgameStage
    from contracts_amd.vector_env import BatchedBaseEnv
    from oracle.pyoracle import Oracle
    E, ticks = 24, 75
    np.random.seed(11)
    if which == "cleanup_contract":
        n, kind, contract, horizon = 4, "cleanup", "cleanup", 30
        env = SeparateContractSubgameStage(CleanupEnv(num_agents=n, horizon=horizon), CleanupContract(n), n, True)
    elif which == "harvest_plain":
        n, kind, contract, horizon = 3, "harvest", None, 20
        env = HarvestEnv(num_agents=n, horizon=horizon, disable_firing=False)
    elif which == "harvest_features_contract":
        n, kind, contract, horizon = 2, "harvest_features", "harvest_local", 25
        env = SeparateContractSubgameStage(HarvestFeatures(num_agents=n, horizon=horizon), HarvestFeaturemodLocalContract(n), n, False)
    else:
        n, kind, contract, horizon = 4, "selfdrive", "selfdrive_distprop", None
        env = SeparateContractSubgameStage(SelfAcceleratingCarEnv(num_agents=n), SelfdriveContractDistprop(n), n, False)
    base = env.base_env if contract else env
    base.seed(4321)  # MapEnv.seed: replica i of the vector env runs the stream seeded 4321 + i
    venv = env.to_base_env(make_env=None, num_envs=E, remote_envs=False, remote_env_batch_wait_ms=0,
                           restart_failed_sub_environments=False)
    assert isinstance(venv, BatchedBaseEnv) and venv.num_envs == E and venv.kind == kind and venv.contract == contract
    cfg = venv.engine.cfg
    if horizon:
        assert cfg.horizon == horizon
    if contract == "cleanup":
        assert cfg.contract_high == float(np.float32(0.2))  # the Box's float32 bound (two_stage_train.py:39-40)
    firing = which == "harvest_plain"
    assert venv.engine.firing == firing
    na = venv.engine.num_actions
    rs = np.random.RandomState(3)

    def policy(env_id, key, ob):
        if kind == "selfdrive":
            return np.array([rs.uniform(-0.1, 0.1)], np.float32)
        return int(rs.randint(na))

    venv.engine.timing_begin()
    episodes, agent_steps, transcript = _sampler_loop(venv, policy, ticks)
    ms, launches = venv.engine.timing_end()
    assert launches == ticks, "one step launch per sampler tick, got %d for %d ticks" % (launches, ticks)
    assert agent_steps >= E * ticks and (episodes >= E or kind == "selfdrive")

    # the same sampler loop over the oracle, behind a minimal BaseEnv face: identical transcript
    kw = dict(contract=contract)
    if horizon:
        kw["horizon"] = horizon
    if firing:
        kw["firing"] = True
    orc = Oracle(kind, E, n, **kw)
    orc.seed(seed0=4321)
    orc.reset()
    keys = ["a%d" % i for i in range(n)]

    class OracleBaseEnv:
        def __init__(self):
            self.stepped = False
            self.acted = np.ones((E, n), np.uint8)

        def _obs(self, e):
            if kind == "selfdrive":
                who = range(n) if not self.stepped else np.nonzero(self.acted[e])[0]
                return {keys[i]: 0 for i in who}
            return {k: 0 for k in keys}

        def poll(self):
            if not self.stepped:
                z = {e: {k: 0.0 for k in keys} for e in range(E)}
                return ({e: self._obs(e) for e in range(E)}, z, {e: {"__all__": False} for e in range(E)}, {}, {})
            r = orc.reward if (contract or kind == "selfdrive") else orc.base_reward
            rew = {e: {keys[i]: float(r[e][i]) for i in range(n) if self.acted[e, i]} for e in range(E)}
            return ({e: self._obs(e) for e in range(E)}, rew, {e: {"__all__": bool(orc.done[e])} for e in range(E)}, {}, {})

        def try_reset(self, e):
            m = np.zeros((E,), np.uint8)
            m[e] = 1
            orc.reset(m)
            self.acted[e] = 1
            return {e: {k: 0 for k in keys}}

        def send_actions(self, acts):
            if kind == "selfdrive":
                a = np.zeros((E, n), np.float32)
                act = np.zeros((E, n), np.uint8)
                for e in range(E):
                    for k, v in acts[e].items():
                        a[e, int(k[1:])] = v[0]
                        act[e, int(k[1:])] = 1
                self.acted = act
                orc.step(a, act)
            else:
                orc.step(np.array([[acts[e][k] for k in keys] for e in range(E)], np.uint8))
            self.stepped = True

    ob = OracleBaseEnv()
    rs = np.random.RandomState(3)
    if kind != "selfdrive":  # (selfdrive's per-agent key sets: test_base_env_protocol_selfdrive_matches_oracle)
        ep2, steps2, transcript2 = _sampler_loop(ob, policy, ticks)
        assert (ep2, steps2) == (episodes, agent_steps)
        assert len(transcript) == len(transcript2)
        for (e1, r1, d1), (e2, r2, d2) in zip(transcript, transcript2):
            assert e1 == e2 and d1 == d2 and r1.keys() == r2.keys()
            for k in r1:
                assert abs(r1[k] - r2[k]) < 1e-9
    venv.stop()
    orc.close()
    env.close() if hasattr(env, "close") else base.close()


@pytest.mark.gpu
def test_to_base_env_fallbacks_keep_object_per_env_semantics():
    """num_envs == 1, feature-vector grid envs and user-defined host contracts get SubEnvBaseEnv: RLlib's
    object-per-sub-env wrapper over the adapters themselves, make_env(i) building the additional ones"""
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.vector_hook import SubEnvBaseEnv
    np.random.seed(5)
    env = CleanupEnv(num_agents=2, horizon=5)
    one = env.to_base_env(num_envs=1)
    assert isinstance(one, SubEnvBaseEnv) and one.get_sub_environments() == [env]
    obs, rew, dones, infos, _ = one.poll()
    assert list(obs) == [0] and obs[0]["a0"]["image"].shape == (15, 15, 3)
    one.send_actions({0: {"a0": 4, "a1": 4}})
    obs, rew, dones, infos, _ = one.poll()
    assert rew[0] == {"a0": 0, "a1": 0} and dones[0]["__all__"] is False
    feat = CleanupEnv(num_agents=2, horizon=5, image_obs=False)
    made = []

    def make_env(i):
        made.append(i)
        return CleanupEnv(num_agents=2, horizon=5, image_obs=False)

    three = feat.to_base_env(make_env=make_env, num_envs=3)
    assert isinstance(three, SubEnvBaseEnv) and three.num_envs == 3 and made == [1, 2]
    obs, *_ = three.poll()
    assert sorted(obs) == [0, 1, 2] and obs[2]["a1"].shape == (14,)
    with pytest.raises(ValueError):
        feat.to_base_env(num_envs=2)
    three.stop()
    one.stop()


@pytest.mark.gpu
def test_try_reset_batching_is_optional_and_metrics_follow_the_tick():
    """(a) with batch_done_resets=False try_reset(e) resets exactly env e — a done env the caller leaves alone
    stays done; with the default the first try_reset after a tick resets every done env in one launch; (b) env_metrics()
    serves the finished episode's rows from the poll that reported the done until the next send_actions, the running
    episode's afterwards"""
    from contracts_amd.vector_env import BatchedBaseEnv
    E, n, H = 4, 2, 6
    keys = ["a0", "a1"]
    for batching in (False, True):
        venv = BatchedBaseEnv("harvest", E, n, seed0=11, horizon=H, batch_done_resets=batching)
        venv.poll()
        for t in range(H):
            venv.send_actions({e: {k: 4 for k in keys} for e in range(E)})
            obs, rew, dones, infos, _ = venv.poll()
        assert all(dones[e]["__all__"] for e in range(E))
        m_done = venv.env_metrics(2)
        assert "equality" in m_done  # the finished episode's rows
        venv.try_reset(1)
        ts = venv.engine.download("timestep")
        assert ts[1] == 0
        assert (ts[[0, 2, 3]] == 0).all() if batching else (ts[[0, 2, 3]] == H).all()
        assert "equality" in venv.env_metrics(2)  # a reset in between does not clear them
        for e in (0, 2, 3):
            venv.try_reset(e)
        venv.send_actions({e: {k: 4 for k in keys} for e in range(E)})
        assert "equality" not in venv.env_metrics(2)  # stepped again: the running episode (before the next poll, too)
        venv.poll()
        assert "equality" not in venv.env_metrics(2)
        venv.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract", [("cleanup", 8, "cleanup"), ("harvest", 3, None), ("cleanup", 2, None),
                                             ("harvest_features", 2, "harvest_local"), ("harvest_features", 4, None),
                                             ("cleanup_features", 2, "cleanup"), ("cleanup_features", 3, None)])
def test_recycled_dict_protocol_equals_the_rebuilt_one(kind, n, contract):
    """recycle_dicts=True (dictionary trees kept over page-locked snapshots, refreshed in place by the C loops) hands out,
    tick for tick, what recycle_dicts=False builds from scratch — same keys in the same order, same value types, equal
    values — and what poll() returned at tick t is still intact at tick t + 1 (the two-generation contract)"""
    from contracts_amd.vector_env import BatchedBaseEnv
    E, T, horizon = 70, 45, 13
    kw = dict(contract=contract, seed0=321, horizon=horizon)
    if kind in ("cleanup", "harvest"):
        kw["firing"] = True
    fast, slow = BatchedBaseEnv(kind, E, n, recycle_dicts=True, **kw), BatchedBaseEnv(kind, E, n, recycle_dicts=False, **kw)
    keys = ["a%d" % i for i in range(n)]
    rs = np.random.RandomState(4)
    na = fast.engine.num_actions

    def same_tree(a, b, path=""):
        assert type(a) is type(b) or (isinstance(a, (float, np.floating)) and isinstance(b, (float, np.floating))), (path, type(a), type(b))
        if isinstance(a, dict):
            assert list(a) == list(b), (path, list(a), list(b))
            for k in a:
                same_tree(a[k], b[k], "%s/%s" % (path, k))
        elif isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), path
        else:
            assert a == b, (path, a, b)

    for v in (fast, slow):
        v.poll()
    held = None
    for t in range(T):
        a = rs.randint(na, size=(E, n))
        ad = {e: {k: int(a[e, i]) for i, k in enumerate(keys)} for e in range(E)}
        fast.send_actions(ad)
        slow.send_actions(ad)
        fo, fr, fd, fi, _ = fast.poll()
        so, sr, sd, si, _ = slow.poll()
        assert type(fo) is dict and sorted(fo) == list(range(E))
        for e in range(E):
            same_tree(fo[e], so[e], "obs/%d" % e)
            same_tree(fr[e], sr[e], "rew/%d" % e)
            same_tree(fd[e], sd[e], "done/%d" % e)
            same_tree(fi[e], si[e], "info/%d" % e)
        if held is not None:  # last tick's dictionaries were not touched by this tick
            (ho, hr, hd, hi), (co, cr, cd, ci) = held
            for e in (0, E // 2, E - 1):
                same_tree(ho[e], co[e], "held obs")
                same_tree(hr[e], cr[e], "held rew")
                same_tree(hi[e], ci[e], "held info")
            assert ho is not fo
        import copy
        held = ((fo, fr, fd, fi), tuple(copy.deepcopy({e: m[e] for e in (0, E // 2, E - 1)}) for m in (fo, fr, fd, fi)))
        done_ids = [e for e in range(E) if fd[e]["__all__"]]
        assert done_ids == [e for e in range(E) if sd[e]["__all__"]]
        for e in done_ids:
            same_tree(fast.try_reset(e)[e], slow.try_reset(e)[e], "reset/%d" % e)
    # a bad action id surfaces at the poll of the recycled path (the step itself is asynchronous)
    from contracts_amd._lib import EngineError
    bad = {e: dict.fromkeys(keys, 0) for e in range(E)}
    bad[3]["a0"] = 200
    fast.send_actions(bad)
    with pytest.raises(EngineError):
        fast.poll()
    fast.stop()
    slow.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,contract", [("harvest_features", 2, "harvest_local"), ("harvest_features", 2, None), ("cleanup", 3, "cleanup"),
                                             ("cleanup_features", 2, None)])
def test_a_sampler_that_holds_observations_for_several_ticks(kind, n, contract):
    """What RLlib's collectors do with a Box observation space (NoPreprocessor / NoFilter hand the env's ndarray through by
    reference, the rollout fragment keeps it): a fake sampler keeps every obs / info object poll() returned for 1, 2 and 3 ticks
    and then reads it.  The DEFAULT hook of the Box-space feature kinds (recycle_dicts="auto" -> rebuilt per tick) and every
    hook with recycle_dicts=False must give it back unchanged; recycle_dicts="checked" gives it back unchanged after one tick
    — the documented contract of the recycled trees — and raises StaleDictError after two or three, with the raw buffers
    poisoned (NaN), instead of handing out a later tick's values; tick for tick its values equal the rebuilt path's."""
    import copy
    from contracts_amd.vector_env import BatchedBaseEnv, StaleDictError
    E, T = 40, 9
    kw = dict(contract=contract, seed0=55, horizon=50)
    keys = ["a%d" % i for i in range(n)]
    box_space = kind in ("harvest_features", "cleanup_features")
    default = BatchedBaseEnv(kind, E, n, **kw)
    assert default.recycle_dicts == ("fresh_obs" if box_space else True)  # Box spaces: recycled machinery, fresh rows / info dicts per tick
    rebuilt = default if box_space else BatchedBaseEnv(kind, E, n, recycle_dicts=False, **kw)
    checked = BatchedBaseEnv(kind, E, n, recycle_dicts="checked", **kw)
    assert checked.recycle_dicts == "checked"
    rs = np.random.RandomState(8)
    na = default.engine.num_actions
    sample = (0, E // 3, E - 1)

    def equal(a, b):
        if isinstance(a, dict):
            return list(a) == list(b) and all(equal(a[k], b[k]) for k in a)
        if isinstance(a, np.ndarray):
            return a.shape == b.shape and np.array_equal(np.asarray(a), np.asarray(b))
        return a == b

    for v in {id(default): default, id(rebuilt): rebuilt, id(checked): checked}.values():
        v.poll()
    held = {"rebuilt": [], "checked": []}
    stale_seen = 0
    for t in range(T):
        a = rs.randint(na, size=(E, n))
        ad = {e: {k: int(a[e, i]) for i, k in enumerate(keys)} for e in range(E)}
        out = {}
        for name, v in (("rebuilt", rebuilt), ("checked", checked)):
            v.send_actions(ad)
            o, r, d, i, _ = v.poll()
            out[name] = (o, i)
            held[name].append(({e: o[e] for e in sample}, {e: i[e] for e in sample},
                               copy.deepcopy({e: {k: (np.array(x) if isinstance(x, np.ndarray) else {kk: np.array(xx) for kk, xx in x.items()})
                                                   for k, x in o[e].items()} for e in sample}),
                               copy.deepcopy({e: {k: {kk: (np.array(xx) if isinstance(xx, np.ndarray) else xx) for kk, xx in inf.items()}
                                                   for k, inf in i[e].items()} for e in sample})))
        for e in sample:  # the stamped hand-out holds the same values as the rebuilt one, this tick
            assert equal(out["checked"][0][e], out["rebuilt"][0][e]) and equal(out["checked"][1][e], out["rebuilt"][1][e]), (t, e)
        for age in (1, 2, 3):
            if t - age < 0:
                continue
            ho, hi, co, ci = held["rebuilt"][t - age]
            for e in sample:  # rebuilt per tick: what the sampler kept is what it was given
                assert equal(ho[e], co[e]) and equal(hi[e], ci[e]), ("rebuilt", t, age, e)
            ho, hi, co, ci = held["checked"][t - age]
            if age == 1:
                for e in sample:
                    assert equal(ho[e], co[e]) and equal(hi[e], ci[e]), ("checked", t, age, e)
            else:
                for e in sample:
                    with pytest.raises(StaleDictError):
                        equal(ho[e], co[e])
                    with pytest.raises(StaleDictError):
                        equal(hi[e], ci[e])
                    stale_seen += 1
                # the raw memory behind an array kept from that tick was poisoned before it was rewritten... and has been
                # rewritten since (age 2: by this very tick); what np.asarray() of the stale wrapper reads is never the kept value's
                # owner any more — the wrapper's hooks are what protects the consumer
    assert stale_seen > 0
    # a sampler that keeps the TOP-LEVEL mappings of tick t and first indexes them at t + 2 must get StaleDictError
    # too — not wrappers stamped with the newer epoch over the newer tick's data
    a = rs.randint(na, size=(E, n))
    ad = {e: {k: int(a[e, i]) for i, k in enumerate(keys)} for e in range(E)}
    checked.send_actions(ad)
    o_kept, _, _, i_kept, _ = checked.poll()
    first = o_kept[sample[0]]  # indexed in its own tick: fine, and stays readable for one more tick
    for _ in range(2):
        checked.send_actions(ad)
        checked.poll()
    with pytest.raises(StaleDictError):
        o_kept[sample[1]]
    with pytest.raises(StaleDictError):
        i_kept[sample[1]]
    with pytest.raises(StaleDictError):
        equal(first, first)
    import pickle
    fresh_o = checked.poll  # (a pickled stamped array comes back with its dtype)
    checked.send_actions(ad)
    o_now, _, _, i_now, _ = checked.poll()
    some = i_now[sample[0]]["a0"].get("feature_obs") if not (kind == "cleanup_features") else None
    if some is not None:
        back = pickle.loads(pickle.dumps(some))
        assert isinstance(back, np.ndarray) and back.dtype == np.float64 and np.array_equal(back, np.asarray(some.copy()))
    for v in {id(default): default, id(rebuilt): rebuilt, id(checked): checked}.values():
        v.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["joint_cleanup_n3_global", "joint_cleanup_n3_concat", "joint_harvest_n2_global"])
def test_joint_baseline_through_the_batched_hook(name):
    """`to_base_env(num_envs=E)` of a JointEnv over a pixel env is ONE handle (BatchedJointBaseEnv), one
    launch per tick.  E = 1 024; sub-env 5 is seeded like the reference-generated joint_* fixture and fed its actions: reset
    observation, every step's observation (global colour map from ce_global_view / stacked views), summed reward, dones and
    summed infos reproduce the reference's JointEnv trace; every OTHER sub-env is held to the oracle (views, global map composed
    from its grid + agents, rewards, infos) on its own random actions, incl. a mid-episode try_reset (agents unpainted)."""
    import hashlib
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.harvest_new import HarvestEnv
    from contracts_amd.environments.map_env import AGENT_RGB, CELL_RGB
    from contracts_amd.environments.two_stage_train import JointEnv
    from contracts_amd.environments.vector_hook import to_base_env
    from contracts_amd.vector_env import BatchedJointBaseEnv
    from oracle.pyoracle import Oracle
    import golden_check as gc
    g = np.load("%s/%s.npz" % (gc.GOLDEN_DIR, name))
    kind, n, seed, mode = str(g["kind"]), int(g["n"]), int(g["seed"]), str(g["mode"])
    E, J = 1024, 5
    base = (CleanupEnv if kind == "cleanup" else HarvestEnv)(num_agents=n)
    env = JointEnv(base, num_agents=n, global_obs=mode == "global", concatenated_obs=mode == "concat")
    venv = to_base_env(env, num_envs=E, seed0=seed - J)
    assert isinstance(venv, BatchedJointBaseEnv) and venv.num_envs == E and venv.mode == ("global" if mode == "global" else "concatenated")
    orc = Oracle(kind, E, n)
    orc.seed(seed0=seed - J)
    orc.reset()

    def want_image(e, painted):
        if mode == "concat":
            return np.concatenate([orc.obs[e][i] for i in range(n)], axis=-1)
        rgb = CELL_RGB[orc.grid[e].reshape(venv.engine.b.grid_h, venv.engine.b.grid_w)].copy()
        if painted:
            for i, a in enumerate(orc.agents[e]):
                rgb[a[0], a[1]] = AGENT_RGB[i]
        return rgb

    def u8(img):
        u = np.rint(img * 255).astype(np.uint8)
        assert np.array_equal(u / 255, img)
        return u

    obs, rew, dones, infos, _ = venv.poll()
    assert sorted(obs.keys()) == list(range(E)) and rew[7] == {"a0": 0.0} and dones[7] == {"__all__": False} and infos[7] == {"a0": {}}
    assert np.array_equal(u8(obs[J]["a0"]["image"]), g["reset_obs"]) and tuple(obs[J]["a0"]["image"].shape) == tuple(g["obs_space_shape"])
    for e in range(0, E, 37):
        assert np.array_equal(u8(obs[e]["a0"]["image"]), want_image(e, False)), e
    rs = np.random.RandomState(4)
    na = venv.engine.num_actions
    second = "cleaned_squares" if kind == "cleanup" else "eaten_close_apples"
    painted = np.zeros(E, bool)
    for t in range(len(g["actions"])):
        a = rs.randint(na, size=(E, n)).astype(np.uint8)
        a[J] = g["actions"][t]
        venv.send_actions({e: {"a0": a[e]} for e in range(E)})
        orc.step(a)
        painted[:] = True
        obs, rew, dones, infos, _ = venv.poll()
        img = u8(obs[J]["a0"]["image"])
        if t < len(g["obs"]):
            assert np.array_equal(img, g["obs"][t]), t
        assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), np.uint8), g["obs_sha"][t]), t
        assert float(rew[J]["a0"]) == g["rew"][t] and dones[J] == {"a0": bool(g["done"][t]), "__all__": bool(g["done"][t])}
        keys = sorted(infos[J]["a0"].keys())
        assert ",".join(keys) == str(g["info_keys"][t])
        vals = np.concatenate([np.atleast_1d(np.asarray(infos[J]["a0"][k], np.float64)).ravel() for k in keys])
        assert np.array_equal(vals, g["info_vals"][t]), t
        for e in range(t % 29, E, 29):  # the rest of the batch against the oracle, a different residue class every tick
            assert np.array_equal(u8(obs[e]["a0"]["image"]), want_image(e, painted[e])), (t, e)
            assert rew[e]["a0"] == sum(int(x) for x in orc.base_reward[e]) and type(rew[e]["a0"]) is int
            inf = infos[e]["a0"]
            assert inf["eaten_apples"] == int(orc.info[e][:, 0].sum()) and inf[second] == int(orc.info[e][:, 1].sum())
            assert np.array_equal(inf["feature_obs"], orc.features[e].astype(np.float64).sum(axis=0))
        if t == 40:  # a mid-episode reset of two sub-envs: their maps show no agents until they step again
            mask = np.zeros(E, np.uint8)
            mask[[3, 900]] = 1
            orc.reset(mask)
            for e in (3, 900):
                ob = venv.try_reset(e)
                painted[e] = False
                assert np.array_equal(u8(ob[e]["a0"]["image"]), want_image(e, False)), e
    # the tensor path: the same tick from a dense plane, the global map stays on the device
    a = rs.randint(na, size=(E, n)).astype(np.uint8)
    venv.send_actions_array(a)
    orc.step(a)
    tt = venv.poll_tensors()
    venv.check_faults()
    if mode == "global":
        gv = tt["global_view"].cpu().numpy()
        for e in range(0, E, 41):
            assert np.array_equal(gv[e], want_image(e, True)), e
    assert np.array_equal(tt["reward"].cpu().numpy().sum(axis=1), orc.base_reward.sum(axis=1))
    venv.stop()
    base.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["global", "concatenated"])
def test_joint_image_blocks_take_turns(mode):
    """BatchedJointBaseEnv converts a stepped tick's float64 images into one of two host blocks used in turn: what poll() handed
    out at tick t is untouched through t + 1 and rewritten at t + 2 (BatchedBaseEnv's recycling contract, for the image arrays
    only; reset observations never take a turn); recycle_images=False hands out new arrays every tick; the hook maps the
    recycling knob onto it ("off" / "checked" -> new arrays)."""
    from contracts_amd.environments.cleanup_new import CleanupEnv
    from contracts_amd.environments.two_stage_train import JointEnv
    from contracts_amd.environments.vector_hook import to_base_env
    from contracts_amd.vector_env import BatchedJointBaseEnv
    E, n = 300, 3
    rs = np.random.RandomState(12)
    for recycle in ("auto", False):
        venv = BatchedJointBaseEnv("cleanup", E, n, mode=mode, seed0=11, recycle_images=recycle)
        first = venv.poll()[0]
        r0 = first[7]["a0"]["image"]
        r0_copy = r0.copy()
        held = []
        for t in range(5):
            venv.send_actions_array(rs.randint(venv.engine.num_actions, size=(E, n)).astype(np.uint8))
            obs, rew, dones, infos, _ = venv.poll()
            assert type(obs) is dict and len(obs) == E and type(infos[E - 1]["a0"]["feature_obs"]) is np.ndarray
            img = obs[7]["a0"]["image"]
            held.append((img, img.copy()))
            if t == 2:  # a reset observation in between does not disturb the turn order
                ob = venv.try_reset(9)
                assert not any(np.shares_memory(ob[9]["a0"]["image"], h[0]) for h in held)
            if t >= 1:
                assert np.array_equal(held[t - 1][0], held[t - 1][1]), t  # tick t - 1 is still what it was
                assert not np.shares_memory(held[t][0], held[t - 1][0])
            if t >= 2:
                assert np.shares_memory(held[t][0], held[t - 2][0]) == (recycle == "auto"), t
                if recycle is False:
                    assert np.array_equal(held[t - 2][0], held[t - 2][1])
        assert np.array_equal(r0, r0_copy)
        venv.stop()
    base = CleanupEnv(num_agents=n)
    for word, want in (("off", False), ("checked", False), ("auto", True), (None, True)):
        env = JointEnv(base, num_agents=n, global_obs=mode == "global", concatenated_obs=mode == "concatenated")
        venv = to_base_env(env, num_envs=8, seed0=3, recycle_dicts=word)
        assert isinstance(venv, BatchedJointBaseEnv) and venv._recycle_images is want
        venv.stop()
    base.close()
