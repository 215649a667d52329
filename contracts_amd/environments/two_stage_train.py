"""SeparateContractSubgameStage — drop-in for environments/two_stage_train.py:17-187
(`SeparateContractEnv` + `SeparateContractSubgameStage`): wraps a base env adapter and a contract,
samples theta at reset and returns contract-transferred rewards.  In the reference the wrapper calls
`contract.compute_transfer` and redistributes rewards in Python; here both are the fused epilogue of the
engine's step kernel, switched on for the base env's handle (ce_set_contract).

`JointEnv` (two_stage_train.py:476-617) — one centralised agent 'a0' driving all agents of a base env, with the
global-map, concatenated-views or duplicated-feature observation — is a thin re-keying of the base env's dictionaries
and is provided too.

Out of scope: SeparateContractNegotiateStage, SeparateContractCombinedStage,
NegotiationSolver — RL-algorithm logic that calls this path but is not it."""
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


class JointEnv(_Base):
    """Centralised-control view of a base env (reference two_stage_train.py:476-617): the single agent 'a0' sends one
    action per base agent and receives the summed reward, summed infos and one of three observations —
    `global_obs` (the whole colour map), `concatenated_obs` (all egocentric views stacked on the channel axis) or the
    feature vectors (`duplicate_obs`: all of them concatenated; otherwise the first still-active agent's)."""

    def __init__(self, base_env, num_agents=2, duplicate_obs=False, concatenated_obs=False, global_obs=False, **kwargs):
        self.base_env = base_env
        self.num_agents = num_agents
        self.duplicate_obs, self.concatenated_obs, self.global_obs = duplicate_obs, concatenated_obs, global_obs
        self._agents = ["a%d" % i for i in range(num_agents)]
        if global_obs or concatenated_obs:
            self.observation_space = (base_env.global_observation_space if global_obs
                                      else base_env.concatenated_observation_space)
            self.action_space = base_env.global_action_space
        else:
            bo, ba = base_env.observation_space, base_env.action_space
            self.observation_space = bo if not duplicate_obs else spaces.Box(
                low=np.concatenate([bo.low] * num_agents), high=np.concatenate([bo.high] * num_agents))
            self.action_space = spaces.Box(low=np.concatenate([ba.low] * num_agents), high=np.concatenate([ba.high] * num_agents))
            self.curr_agent_lst = list(self._agents)

    def _pixel_obs(self, base_obs):
        if self.global_obs:
            return {"a0": self.base_env.get_global_obs()}
        return {"a0": {"image": np.concatenate([base_obs[k]["image"] for k in base_obs.keys()], axis=-1)}}

    def reset(self):
        base_obs = self.base_env.reset()
        if self.global_obs or self.concatenated_obs:
            return self._pixel_obs(base_obs)
        self.agent_obs = dict(base_obs)
        self.curr_agent_lst = list(self._agents)
        return {"a0": np.concatenate([base_obs[k] for k in self._agents])}

    @staticmethod
    def _joint(rews, dones, infos, agents):
        summed = {key: sum(infos[k][key] for k in agents) for key in infos[agents[0]].keys()}
        return ({"a0": sum(rews.values())}, {"a0": dones["__all__"], "__all__": dones["__all__"]}, {"a0": summed})

    def step(self, acts):
        if self.global_obs or self.concatenated_obs:
            action_dict = {k: acts["a0"][i] for i, k in enumerate(self._agents)}
            base_obs, rews, dones, infos = self.base_env.step(action_dict)
            return (self._pixel_obs(base_obs),) + self._joint(rews, dones, infos, self._agents)
        width = self.base_env.action_space.shape[0]
        action_dict = {k: np.array(acts["a0"][i * width:(i + 1) * width]) for i, k in enumerate(self._agents)
                       if k in self.curr_agent_lst}
        base_obs, rews, dones, infos = self.base_env.step(action_dict)
        for k in self.curr_agent_lst:
            self.agent_obs[k] = base_obs[k]
        obs = {"a0": np.concatenate([self.agent_obs[k] for k in self._agents]) if self.duplicate_obs
               else self.agent_obs[self.curr_agent_lst[0]]}
        out = (obs,) + self._joint(rews, dones, infos, list(self.curr_agent_lst))
        self.curr_agent_lst = [k for k in self.curr_agent_lst if not dones.get(k, False)]
        return out

    def render(self, mode="rgb"):
        return self.base_env.render()
