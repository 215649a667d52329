"""`to_base_env()` for the drop-in env classes — the point where RLlib turns a `MultiAgentEnv` into the `BaseEnv` its
sampler polls (ray/rllib/env/multi_agent_env.py: `MultiAgentEnv.to_base_env(make_env, num_envs, remote_envs,
remote_env_batch_wait_ms, restart_failed_sub_environments)`; reference call sites: the env classes derive from
`MultiAgentEnv` — environments/map_env.py:60, two_stage_train.py:17, self_driving_car_accelerate.py:18,
harvest_features.py:60 — and `num_envs_per_worker` reaches RLlib through the trainer config built in
utils/ray_config_utils.py:126-214).

RLlib's own implementation wraps `num_envs` Python env objects and steps them one after the other.  Here, for
`num_envs > 1`, the hook returns ONE `BatchedBaseEnv` over one engine handle with `num_envs` replicas configured like the
env it was called on (kind, agents, horizon, firing / reward flags, contract and its bounds): a sampler tick is one kernel
launch, with `runner.py` / `ray_config_utils.py` unchanged.  The replicas run private RNG streams seeded
`seed0 + index` (`seed0` = the value last passed to `env.seed()`, else the next draw of a COPY of the process-global
`np.random`: seeded scripts stay reproducible and the global stream does not move).

A `JointEnv` over a pixel grid env (`global_obs` or `concatenated_obs`: the reference's joint baseline) gets a
`BatchedJointBaseEnv`: the same single handle, the centralised agent's MultiDiscrete action split into the [E, n] plane, summed
rewards / infos, the global colour map produced on the device (ce_global_view) or the stacked egocentric views.

Configurations the batched hook does not serve — a single sub-env, remote sub-envs, grid envs in feature-vector or
one-hot mode, a user-defined host contract, the negotiate / combined / solver stages, a JointEnv over the float path — get
`SubEnvBaseEnv`, which keeps
RLlib's object-per-sub-env semantics over the adapters themselves (`make_env(i)` builds the additional ones).
"""
import numpy as np

from ..vector_env import BatchedBaseEnv, _RLlibBaseEnv


class SubEnvBaseEnv(_RLlibBaseEnv):
    """`BaseEnv` protocol over a list of per-env adapter objects (what RLlib's MultiAgentEnvWrapper does): every
    sub-env is stepped by its own `step()` call.  The fallback of `to_base_env` and the `num_envs == 1` case."""

    def __init__(self, envs):
        self.envs = list(envs)
        self._fresh = set(range(len(self.envs)))
        self._last = {}

    def poll(self):
        obs, rew, dones, infos = {}, {}, {}, {}
        for i in sorted(self._fresh):
            o = self.envs[i].reset()
            obs[i], rew[i], dones[i] = o, {k: 0.0 for k in o}, {"__all__": False}
            infos[i] = {k: {} for k in o}
        self._fresh.clear()
        for i, (o, r, d, inf) in self._last.items():
            obs[i], rew[i], dones[i], infos[i] = o, r, d, inf
        self._last = {}
        return obs, rew, dones, infos, {}

    def send_actions(self, action_dict):
        for i, acts in action_dict.items():
            self._last[i] = self.envs[i].step(acts)

    def try_reset(self, env_id=None):
        ids = range(len(self.envs)) if env_id is None else [env_id]
        return {i: self.envs[i].reset() for i in ids}

    def get_sub_environments(self, as_dict=False):
        return dict(enumerate(self.envs)) if as_dict else self.envs

    @property
    def num_envs(self):
        return len(self.envs)

    def stop(self):
        for e in self.envs:
            close = getattr(e, "close", None)
            if close:
                close()


def _engine_config(base):
    """(kind, engine kwargs) of a base adapter, or None when the batched hook cannot serve its configuration"""
    kind = getattr(base, "KIND", None)
    if kind in ("cleanup", "harvest"):
        if not base.image_obs or base.one_hot_id:
            return None  # the vector hook hands out image observations only
        return kind, dict(horizon=base.horizon, firing=not base.disable_firing, collective=base.use_collective_reward,
                          inequity=base.inequity_averse_reward, alpha=base.alpha, beta=base.beta,
                          rng=getattr(base, "vector_rng", "mt19937"), ascii_map=getattr(base, "_map_rows", None))
    if kind in ("harvest_features", "cleanup_features"):
        if getattr(base, "image_obs", False) and kind == "harvest_features":
            return None  # the history-dependent painted map is a host-side view of the single-env adapter
        return kind, dict(horizon=base.horizon)
    if type(base).__name__ == "SelfAcceleratingCarEnv":
        return "selfdrive", dict(collision_on=base.collision_on, low_bound=base.low_bound, high_bound=base.high_bound,
                                 start_vel=base.start_vel, start_vel_ambulance=base.start_vel_ambulance)
    return None


_TWIN = {"state": None, "rs": None}  # private generator behind the default seeds (see vector_seed0)


def vector_seed0(env):
    """seed of replica 0 (replica i gets seed0 + i): the value last passed to `env.seed()`, else the next draw of a PRIVATE
    generator started from a copy of the process-global np.random state — seeded scripts stay reproducible, the global stream
    (which the rng="global" adapters mirror for reference parity) does not move, and two unseeded hooks of one process (a
    train and an eval vector env, two env kinds) draw DIFFERENT seeds: the private generator advances from call to call.  It is
    restarted from the global state whenever that has changed since the last call (np.random.seed(), or draws in between)."""
    s = getattr(env, "_vector_seed0", None)
    if s is None:
        st = np.random.get_state()
        key = (st[0], st[1].tobytes(), st[2], st[3], st[4])
        if _TWIN["state"] != key:
            _TWIN["state"] = key
            _TWIN["rs"] = np.random.RandomState()
            _TWIN["rs"].set_state(st)
        s = int(_TWIN["rs"].randint(0, 2 ** 31 - 1))
    return int(s)


def _resolve_recycle(recycle_dicts, *envs):
    """the recycling knob: the argument, else an env's `vector_recycle_dicts`, else CONTRACTS_AMD_VECTOR_RECYCLE, else 'auto'"""
    for env in envs:
        if recycle_dicts is None:
            recycle_dicts = getattr(env, "vector_recycle_dicts", None)
    if recycle_dicts is None:
        import os
        recycle_dicts = os.environ.get("CONTRACTS_AMD_VECTOR_RECYCLE", "auto")
    if isinstance(recycle_dicts, str):
        word = recycle_dicts.strip().lower()
        if word not in _RECYCLE_WORDS:
            raise ValueError("recycle_dicts / CONTRACTS_AMD_VECTOR_RECYCLE: %r is not one of %s" % (recycle_dicts, sorted(_RECYCLE_WORDS)))
        recycle_dicts = _RECYCLE_WORDS[word]
    return recycle_dicts


def to_base_env(env, make_env=None, num_envs=1, remote_envs=False, remote_env_batch_wait_ms=0,
                restart_failed_sub_environments=False, seed0=None, recycle_dicts=None):
    """see the module docstring; `env` is a base adapter or a SeparateContractSubgameStage around one.
    `recycle_dicts` (BatchedBaseEnv's knob, vector_env.py module docstring): None = the env's `vector_recycle_dicts`
    attribute / constructor kwarg, else CONTRACTS_AMD_VECTOR_RECYCLE (auto | on | off | checked), else "auto" — recycled
    dictionary trees only where RLlib copies every observation at once (Dict spaces: the grid kinds); Box-space kinds rebuild."""
    from .two_stage_train import JointEnv, SeparateContractSubgameStage
    num_envs = int(num_envs)
    base, contract_kw, convolutional = env, {}, True
    batched_ok = num_envs > 1 and not remote_envs
    if batched_ok and isinstance(env, JointEnv) and (env.global_obs or env.concatenated_obs) and hasattr(env.base_env, "_ensure_engine"):
        # the joint baseline over a pixel grid env (experiment_configs/cleanup-joint-2agents.json): one handle, one launch per tick
        jcfg = _engine_config(env.base_env)
        if jcfg is not None and jcfg[0] in ("cleanup", "harvest"):
            from ..vector_env import BatchedJointBaseEnv
            kind, kw = jcfg
            if seed0 is None:
                seed0 = vector_seed0(env.base_env)
            if kw.get("rng", "mt19937") != "counter" and int(seed0) + num_envs - 1 > 0xffffffff:
                raise ValueError("to_base_env: seed %d + %d sub-envs runs past 2**32 - 1 (np.random.seed's range)" % (seed0, num_envs))
            # the joint env recycles image BLOCKS only (its dictionaries are new every tick); "checked" has nothing to wrap
            # there and gets fresh arrays, like "off"
            rec = _resolve_recycle(recycle_dicts, env, env.base_env)
            return BatchedJointBaseEnv(kind, num_envs, env.base_env.num_agents, mode="global" if env.global_obs else "concatenated",
                                       seed0=seed0, device=getattr(env.base_env, "_device", 0),
                                       recycle_images=False if rec in (False, "checked") else "auto", **kw)
    if isinstance(env, SeparateContractSubgameStage):
        base = env.base_env
        convolutional = env.convolutional
        if env._host_contract or getattr(base, "_external_theta", False):
            batched_ok = False
        else:
            name, lo, hi, null_prob = env.contract.engine_spec(env.null_prob)
            contract_kw = dict(contract=name, contract_low=lo, contract_high=hi, null_prob=null_prob)
    elif not hasattr(env, "_ensure_engine"):
        batched_ok = False  # a negotiate / combined / solver stage or JointEnv: host-side protocol per env object
    cfg = _engine_config(base) if batched_ok else None
    if cfg is None:
        envs = [env]
        if num_envs > 1:
            if make_env is None:
                raise ValueError("to_base_env(num_envs=%d) of this configuration needs make_env to build the other sub-envs" % num_envs)
            envs += [make_env(i) for i in range(1, num_envs)]
        return SubEnvBaseEnv(envs)
    kind, kw = cfg
    kw.update(contract_kw)
    if seed0 is None:
        seed0 = vector_seed0(base)
    if kw.get("rng", "mt19937") != "counter" and int(seed0) + num_envs - 1 > 0xffffffff:
        # np.random.seed() takes 32 bits: replica i is seeded seed0 + i, and none of them may leave that range
        raise ValueError("to_base_env: seed %d + %d sub-envs runs past 2**32 - 1 (np.random.seed's range); seed the env lower "
                         "or use vector_rng='counter' (64-bit seeds)" % (seed0, num_envs))
    recycle_dicts = _resolve_recycle(recycle_dicts, env, base)
    return BatchedBaseEnv(kind, num_envs, base.num_agents, seed0=seed0, convolutional=convolutional,
                          device=getattr(base, "_device", 0), recycle_dicts=recycle_dicts, **kw)


_RECYCLE_WORDS = {"auto": "auto", "checked": "checked", "on": True, "true": True, "1": True, "off": False, "false": False, "0": False}


class VectorHookMixin:
    """gives an env class RLlib's `to_base_env` entry point (same signature), resolved by `to_base_env` above"""

    vector_recycle_dicts = None  # None = CONTRACTS_AMD_VECTOR_RECYCLE / "auto"; set on the instance (or class) to force a mode

    def to_base_env(self, make_env=None, num_envs=1, remote_envs=False, remote_env_batch_wait_ms=0,
                    restart_failed_sub_environments=False):
        return to_base_env(self, make_env, num_envs, remote_envs, remote_env_batch_wait_ms, restart_failed_sub_environments)
