"""SeparateContractSubgameStage — drop-in for environments/two_stage_train.py:17-187
(`SeparateContractEnv` + `SeparateContractSubgameStage`): wraps a base env adapter and a contract,
samples theta at reset and returns contract-transferred rewards.  In the reference the wrapper calls
`contract.compute_transfer` and redistributes rewards in Python; here both are the fused epilogue of the
engine's step kernel, switched on for the base env's handle (ce_set_contract).

Out of scope: SeparateContractNegotiateStage, SeparateContractCombinedStage,
JointEnv, NegotiationSolver — RL-algorithm logic that calls this path but is not it."""
import numpy as np

from .. import spaces
from .map_env import _Base


class SeparateContractEnv(_Base):
    metadata = {"render.modes": ["rgb_array"]}

    def __init__(self, base_env, contract, num_agents, convolutional, env_params=None, null_prob=0.0, **kwargs):
        self.num_agents = num_agents
        self.base_env = base_env
        self.contract = contract
        self.contract_low = self.contract.contract_space.low
        self.contract_high = self.contract.contract_space.high
        self.contract_state = {"a" + str(i): 0 for i in range(self.num_agents)}
        self.convolutional = convolutional
        self.null_prob = null_prob
        if getattr(contract, "engine_contract", None) is None:
            raise NotImplementedError("contract %r has no fused engine epilogue" % type(contract).__name__)
        base_env._contract = contract.engine_spec(null_prob)
        if base_env._engine is not None:
            base_env._engine.set_contract(*base_env._contract)
        if self.convolutional:
            contract_space = spaces.Box(low=np.concatenate((self.contract_low, np.array([0.0]))),
                                        high=np.concatenate((self.contract_high, np.array([3.0]))))
            obs_space = self.base_env.observation_space
            space_dict = {"contract": contract_space}
            if "features" in obs_space.keys():
                space_dict["features"] = obs_space["features"]
            if "image" in obs_space.keys():
                space_dict["image"] = obs_space["image"]
            self.observation_space = spaces.Dict(space_dict)
        else:
            self.observation_space = spaces.Box(
                low=np.concatenate((self.base_env.observation_space.low, self.contract_low, np.array([0.0]))),
                high=np.concatenate((self.base_env.observation_space.high, self.contract_high, np.array([3.0]))))

    def _theta(self):
        return np.array([float(self.base_env._engine.download("theta")[0])])

    def _with_contract(self, obs, keys):
        theta = self.params["a0"]
        if self.convolutional:
            out = {}
            for i in range(self.num_agents):
                key = "a" + str(i)
                out[key] = obs[key]
                out[key].update({"contract": np.concatenate((theta, np.array([0])))})
            return out
        return {k: np.concatenate((obs[k], theta, np.array([0]))) for k in keys}

    def step(self, acts):
        raw_obs, base_rew, dones, infos = self.base_env.step(acts)
        keys = list(acts.keys())
        rew = self.base_env._engine.download("reward")[0]
        rews = {k: np.float64(rew[int(k[1:])]) for k in keys}
        for k in keys:
            infos[k]["contract_param"] = self.params[k]
        self.obs = {k: raw_obs[k] for k in keys}
        return self._with_contract(self.obs, keys), rews, dones, infos

    def render(self, mode="rgb"):
        return self.base_env.render()

    def reset(self):
        raise NotImplementedError


class SeparateContractSubgameStage(SeparateContractEnv):
    def __init__(self, base_env, contract, num_agents, convolutional, env_params=None, null_prob=0.0, **kwargs):
        super().__init__(base_env, contract, num_agents, convolutional, env_params, null_prob)
        self.action_space = self.base_env.action_space

    def reset(self):
        # base reset + theta sampling happen in one engine call, in the reference's RNG order
        base_obs = self.base_env.reset()
        rand_val = self._theta()
        self.contract_state = {"a" + str(i): 0 for i in range(self.num_agents)}
        self.params = {key: rand_val for key in ["a" + str(i) for i in range(self.num_agents)]}
        self.obs = base_obs
        return self._with_contract(base_obs, ["a" + str(i) for i in range(self.num_agents)])
