"""Import harness for the upstream reference (runs ONLY in the build container).

The reference at /root/reference is a Python/RLlib project; `gym`, `ray` and `cv2`
are not installed here.  This module injects minimal stand-in modules (our own
code, modelled on the gym-0.21 / ray-2.2 behaviours the hot path relies on, see
SURVEY.md Appendix A) into `sys.modules`, puts /root/reference on `sys.path` and
hands back the reference classes.  It is used by `make_golden.py` to produce the
committed fixtures and by `tests/test_oracle_vs_reference.py` (skipped when
/root/reference is absent, i.e. on the GPU box).

Nothing from /root/reference is copied: the reference is imported where it lies.
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("CONTRACTS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "environments"))


# --------------------------------------------------------------------------- #
# gym.spaces stand-ins (gym 0.21 semantics: Box default dtype float32, bounds
# are broadcast to `shape` and cast to dtype).
# --------------------------------------------------------------------------- #
class _Space:
    """gym 0.21: sample() draws from the space's own np_random (seedable with .seed()), not from the global stream"""

    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self.seed()
        return self._np_random

    def seed(self, seed=None):
        self._np_random = np.random.RandomState(seed)
        return [seed]


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low), shape).astype(dtype)
        self.high = np.broadcast_to(np.asarray(high), shape).astype(dtype)
        super().__init__(shape, dtype)

    def sample(self):
        return self.np_random.uniform(self.low, self.high).astype(self.dtype)


class Discrete(_Space):
    def __init__(self, n):
        self.n = n
        super().__init__((), np.int64)

    def sample(self):
        return int(self.np_random.randint(self.n))


class MultiDiscrete(_Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, np.int64)


class Dict(_Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)
        super().__init__(None, None)

    def keys(self):
        return self.spaces.keys()

    def __getitem__(self, k):
        return self.spaces[k]


def _install_stubs():
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_contracts_amd_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    spaces = mod("gym.spaces", Box=Box, Discrete=Discrete, MultiDiscrete=MultiDiscrete, Dict=Dict)
    gym = mod("gym", spaces=spaces, Env=object, _contracts_amd_stub=True)
    gym.spaces = spaces

    class MultiAgentEnv:  # ray.rllib.env.MultiAgentEnv: empty base is enough
        pass

    class DefaultCallbacks:
        pass

    class TaskSettableEnv:
        pass

    ray = mod("ray")
    rllib = mod("ray.rllib")
    env = mod("ray.rllib.env", MultiAgentEnv=MultiAgentEnv)
    algos = mod("ray.rllib.algorithms")
    cbs = mod("ray.rllib.algorithms.callbacks", DefaultCallbacks=DefaultCallbacks)
    agents = mod("ray.rllib.agents", ppo=None)
    utils = mod("ray.rllib.utils")
    fw = mod(
        "ray.rllib.utils.framework",
        try_import_tf=lambda: (None, None, None),
        try_import_torch=lambda: (None, None),
    )
    apis = mod("ray.rllib.env.apis")
    tse = mod("ray.rllib.env.apis.task_settable_env", TaskSettableEnv=TaskSettableEnv)
    ray.rllib = rllib
    rllib.env, rllib.algorithms, rllib.agents, rllib.utils = env, algos, agents, utils
    algos.callbacks = cbs
    utils.framework = fw
    env.apis = apis
    apis.task_settable_env = tse
    mod("cv2")


def load_reference():
    """Returns a namespace with the reference classes of the hot path."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    from environments.cleanup_new import CleanupEnv
    from environments.harvest_new import HarvestEnv
    from environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
    from environments.two_stage_train import SeparateContractSubgameStage
    import contract.contract_list as contract_list

    ns.CleanupEnv = CleanupEnv
    ns.HarvestEnv = HarvestEnv
    ns.SelfAcceleratingCarEnv = SelfAcceleratingCarEnv
    ns.SeparateContractSubgameStage = SeparateContractSubgameStage
    ns.contract_list = contract_list
    return ns


class StubPPOTrainer:
    """Deterministic stand-in for `ray.rllib.agents.ppo.PPOTrainer` (frozen subgame policies of the negotiate stage):
    actions come from a counter-seeded RandomState, independent of the observation, so that the reference stage and
    the rebuilt one — both handed this class — ask for the same action sequence."""

    def __init__(self, config=None, env=None):
        self.n_act = int(config["n_act"])
        self.rs = np.random.RandomState(int(config["seed"]))
        self.calls = []

    def load_checkpoint(self, path):
        pass

    restore = load_checkpoint

    def compute_single_action(self, obs, policy_id=None, **kw):
        self.calls.append(policy_id)
        self._last = (obs, policy_id)
        return int(self.rs.randint(self.n_act))

    # value head of the "policy" that just ran (NegotiationSolver reads trainer.get_policy(id).model.value_function()):
    # a smooth function of the contract parameter in the observation, different per policy, plus seeded noise
    def get_policy(self, policy_id):
        return self

    @property
    def model(self):
        return self

    def value_function(self):
        obs, policy_id = self._last
        c = float(obs["contract"][0]) if isinstance(obs, dict) else float(np.asarray(obs)[-2])
        idx = 0 if policy_id == "policy" else int(policy_id[1:])
        return np.float32(3.0 * np.sin(23.0 * c + 1.7 * idx + 4.0) + 0.05 * self.rs.standard_normal())


def install_stub_trainer():
    """make `from ray.rllib.agents import ppo; ppo.PPOTrainer` resolve to StubPPOTrainer inside the reference"""
    _install_stubs()
    ppo = types.ModuleType("ray.rllib.agents.ppo")
    ppo.PPOTrainer = StubPPOTrainer
    sys.modules["ray.rllib.agents.ppo"] = ppo
    sys.modules["ray.rllib.agents"].ppo = ppo
    return ppo
