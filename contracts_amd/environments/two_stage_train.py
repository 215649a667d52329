"""SeparateContractSubgameStage — drop-in for environments/two_stage_train.py:17-187
(`SeparateContractEnv` + `SeparateContractSubgameStage`): wraps a base env adapter and a contract,
samples theta at reset and returns contract-transferred rewards.  In the reference the wrapper calls
`contract.compute_transfer` and redistributes rewards in Python; here both are the fused epilogue of the
engine's step kernel, switched on for the base env's handle (ce_set_contract).

`JointEnv` (two_stage_train.py:476-617) — one centralised agent 'a0' driving all agents of a base env, with the
global-map, concatenated-views or duplicated-feature observation — is a thin re-keying of the base env's dictionaries
and is provided too.

`SeparateContractNegotiateStage` / `SeparateContractCombinedStage` (two_stage_train.py:188-470) — propose a contract,
accept or reject it, then play the subgame under it — are host-side protocol around the same engine path; the contract
parameter is written into the engine's `theta` buffer (CE_FLAG_EXTERNAL_THETA: resets draw nothing).  The negotiate
stage rolls the whole subgame inside one outer step with frozen policies: any object with RLlib's
`compute_single_action(obs, policy_id=...)` serves (`trainer_factory`), `ray`'s PPOTrainer when it is installed.

`NegotiationSolver` (two_stage_train.py:619-776) picks the contract at reset instead of learning to propose one: it
scores the null contract and `contract_samples` random ones with the frozen policies' value heads
(`trainer.get_policy(id).model.value_function()` after a forward pass) and keeps the best under the 'max' or the
'majority' rule; the episode then runs on the engine under that parameter."""
import copy
import random
import numpy as np

from .. import spaces
from .map_env import _Base, host_np_draw
from .vector_hook import VectorHookMixin


class SeparateContractEnv(VectorHookMixin, _Base):
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
        # one of the three contracts of contract_list.py: transfer + redistribution run as the step kernel's epilogue.
        # Anything else (a user's Contract subclass) keeps the reference's host protocol: the engine steps the base
        # env, `contract.compute_transfer` is called with the reference's arguments and the wrapper redistributes.
        # — and so does a subclass of a shipped contract that overrides compute_transfer (Contract.fused_epilogue).
        fused = contract.fused_epilogue() if hasattr(contract, "fused_epilogue") else getattr(contract, "engine_contract", None)
        self._host_contract = fused is None
        if not self._host_contract:
            base_env._contract = contract.engine_spec(null_prob)
            if base_env._engine is not None:
                base_env._engine.set_contract(*base_env._contract)
        self._host_transfers = 0
        self._external_theta(False)
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

    def _draw_theta(self):
        """two_stage_train.py:163-166 for a host contract: the same two draws, from the stream the base env's reset
        has just advanced (the process-global np.random, or the env's private stream inside the engine)"""
        def draw(rs):
            if rs.rand() > self.null_prob:
                return rs.uniform(low=self.contract_low, high=self.contract_high)
            return self.contract_low
        return host_np_draw(self.base_env, draw)

    def _begin_episode(self):
        self._host_transfers = 0
        self.transferred_reward_dict = {"a" + str(i): [] for i in range(self.num_agents)}

    def _external_theta(self, on):
        """stages that take the contract parameter from an agent's action: the engine draws nothing at reset"""
        self.base_env._external_theta = bool(on)
        if self.base_env._engine is not None:
            self.base_env._engine.set_flags(external_theta=bool(on))

    def _set_theta(self, value):
        self.base_env._ensure_engine().upload("theta", np.array([float(np.asarray(value).reshape(-1)[0])]))

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

    def _step_host_contract(self, acts):
        """SeparateContractEnv.step of the reference (two_stage_train.py:62-121) around a user-defined contract: a
        transfer is a number (split evenly over the other acting agents) or (value, {recipient: proportion})"""
        raw_obs, base_rew, dones, infos = self.base_env.step(acts)
        self.obs = {key: raw_obs[key] for key in acts.keys()}
        transfers = self.contract.compute_transfer(self.obs, acts, base_rew, self.params, infos)
        rews = {key: base_rew[key] for key in acts.keys()}
        others = len(acts.keys()) - 1
        total = 0
        for i in range(self.num_agents):
            payer = "a" + str(i)
            if payer not in acts.keys():
                continue
            due = transfers[payer]
            if type(due) is tuple:
                amount, shares = due
                rews[payer] -= amount
                total += amount
                for j in range(self.num_agents):
                    payee = "a" + str(j)
                    if payee in shares.keys() and payee in rews.keys():
                        rews[payee] += amount * shares[payee]
            else:
                rews[payer] -= due
                total += due
                for j in range(self.num_agents):
                    if i != j and "a" + str(j) in acts.keys():
                        rews["a" + str(j)] += due / others
        self._host_transfers += total
        self.base_env.metrics["transfers"] = self._host_transfers  # the adapter rebuilds its metrics every step
        for k, v in rews.items():
            self.transferred_reward_dict[k].append(v)
        if hasattr(self.base_env, "compute_sustainability") and dones["__all__"]:
            self.base_env.metrics["transfer_sustainability"] = self.base_env.compute_sustainability(
                copy.deepcopy(self.transferred_reward_dict))
            self.base_env.metrics["transfer_equality"] = self.base_env.compute_equality(
                copy.deepcopy(self.transferred_reward_dict))
        keys = list(acts.keys())
        for k in keys:
            infos[k]["contract_param"] = self.params[k]
        return self._with_contract(self.obs, keys), rews, dones, infos

    def step(self, acts):
        if self._host_contract:
            return self._step_host_contract(acts)
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
        rand_val = self._draw_theta() if self._host_contract else self._theta()
        self._begin_episode()
        self.contract_state = {"a" + str(i): 0 for i in range(self.num_agents)}
        self.params = {key: rand_val for key in ["a" + str(i) for i in range(self.num_agents)]}
        self.obs = base_obs
        return self._with_contract(base_obs, ["a" + str(i) for i in range(self.num_agents)])


class JointEnv(VectorHookMixin, _Base):
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


class _ProposalStages(SeparateContractEnv):
    """what the negotiate and the combined stage share: contract_state 2 = propose, 3 = accept / reject, 0 = play"""

    def _stage_reset(self, first_state):
        base_obs = self.base_env.reset()
        self._set_theta(0.0)
        self.obs = copy.deepcopy(base_obs)
        self.last_seen_obs = copy.deepcopy(base_obs)
        self.params = None
        self.contract_state = {"a" + str(i): first_state for i in range(self.num_agents)}
        self._begin_episode()
        zeros = np.zeros(self.contract_low.shape)
        return self._stage_obs(self.obs, {k: zeros for k in self.contract_state})

    def _stage_obs(self, source, params):
        out = {}
        for i in range(self.num_agents):
            key = "a" + str(i)
            tail = np.concatenate((params[key], np.array([self.contract_state[key]])))
            if self.convolutional:
                out[key] = source[key]
                out[key].update({"contract": tail})
            else:
                out[key] = np.concatenate((source[key], tail))
        return out

    def _propose(self, proposal, acts):
        self.params = {key: proposal for key in acts.keys()}
        self.contract_state = {"a" + str(i): 3 for i in range(self.num_agents)}

    def _decide(self, acts):
        """two random responders (all of them when there are at most two) accept jointly with the product of their
        acceptance probabilities; a rejected proposal plays as the null contract"""
        responders = random.sample(range(1, self.num_agents), 2) if self.num_agents > 3 else range(1, self.num_agents)
        prob = 1
        for term in [acts["a" + str(i)][-1] for i in responders]:
            prob *= term
        decision = 1 if random.random() < prob else 0
        null = np.zeros(shape=self.contract_low.shape)
        for i in range(self.num_agents):
            key = "a" + str(i)
            self.contract_state[key] = 0
            self.params[key] = self.params[key] if decision == 1 else null
        self._set_theta(self.params["a0"])
        return decision

    def _zero(self):
        return ({"a" + str(i): 0.0 for i in range(self.num_agents)}, {"a" + str(i): {} for i in range(self.num_agents)})


class SeparateContractNegotiateStage(_ProposalStages):
    def __init__(self, base_env, contract, num_agents, horizon, trainer_config, trainer_env, trainer_path, convolutional,
                 shared, env_params=None, trainer_factory=None, **kwargs):
        super().__init__(base_env, contract, num_agents, convolutional)
        self._external_theta(True)
        self.horizon = horizon
        self.frozen_trainer = _frozen_trainer(trainer_factory, trainer_config, trainer_env, trainer_path, "the negotiate stage")
        self.shared = shared
        self.action_space = spaces.Box(low=np.concatenate((self.contract_low, np.array([0.0]))),
                                       high=np.concatenate((self.contract_high, np.array([1.0]))))
        self.metrics = {"contract": -1, "accepted": 0}

    def reset(self):
        self.metrics = {"contract": -1, "accepted": 0}
        return self._stage_reset(2)

    def _policy_obs(self, key):
        tail = np.concatenate((self.params[key], np.array([0])))
        if self.convolutional:
            obs = self.obs[key]
            obs.update({"contract": tail})
            return obs
        return np.concatenate((self.obs[key], tail))

    def step(self, acts):
        rews, infos = self._zero()
        if self.contract_state["a0"] == 2:
            self.metrics["contract"] = acts["a0"][:-1]
            self._propose(acts["a0"][:-1], acts)
            dones = {"__all__": False}
        else:
            self.metrics["accepted"] = self._decide(acts)
            dones = {"__all__": True}  # the negotiation ends here; the subgame is played out inside this step
            env_dones = {"__all__": False}
            active = ["a" + str(i) for i in range(self.num_agents)]
            played = 0
            while not env_dones["__all__"] and played < self.horizon:
                act_dict = {key: self.frozen_trainer.compute_single_action(self._policy_obs(key), policy_id="policy" if self.shared else key)
                            for key in active}
                _, env_rews, env_dones, infos = super().step(act_dict)
                self.last_seen_obs = {"a" + str(i): self.obs["a" + str(i)] if "a" + str(i) in self.obs else self.last_seen_obs["a" + str(i)]
                                      for i in range(self.num_agents)}
                played += 1
                for key in active:
                    rews[key] += env_rews[key]
                active = [key for key in active if not env_dones.get(key, False)]
        return self._stage_obs(self.last_seen_obs, self.params), rews, dones, infos


def _frozen_trainer(trainer_factory, trainer_config, trainer_env, trainer_path, what):
    if trainer_factory is None:
        try:  # pragma: no cover - RLlib is absent in the build image
            from ray.rllib.agents import ppo
            trainer_factory = ppo.PPOTrainer
        except Exception as exc:
            raise ImportError("%s needs frozen subgame policies: install ray[rllib] or pass "
                              "trainer_factory=<callable returning an object with compute_single_action>" % what) from exc
    trainer = trainer_factory(config=trainer_config, env=trainer_env)
    trainer.load_checkpoint(trainer_path)
    return trainer


class NegotiationSolver(SeparateContractEnv):
    """Contract chosen by search over the frozen value functions (reference two_stage_train.py:619-776).

    reset(): base reset, then V_i(s0, c) for the null contract and `contract_samples` draws of
    `Box(contract_low, contract_high).sample()`; decision_rule 'max' keeps the draw with the largest welfare
    sum_i V_i, 'majority' first drops draws that fewer than half of the agents strictly prefer to the null contract
    (the null contract always stays a candidate).  Ties go to the earliest candidate (np.argmax)."""

    def __init__(self, base_env, contract, num_agents, horizon, trainer_config, trainer_env, trainer_path, convolutional,
                 shared, env_params=None, contract_samples=50, decision_rule="majority", trainer_factory=None, **kwargs):
        super().__init__(base_env, contract, num_agents, convolutional)
        self._external_theta(True)
        self.horizon = horizon
        self.frozen_trainer = _frozen_trainer(trainer_factory, trainer_config, trainer_env, trainer_path, "NegotiationSolver")
        self.shared = shared
        self.contract_param_space = spaces.Box(low=contract.contract_space.low, high=contract.contract_space.high)
        self.num_samples = contract_samples
        self.decision_rule = decision_rule
        self.config = trainer_config
        self.action_space = self.base_env.action_space
        self._agents = ["a" + str(i) for i in range(num_agents)]

    def reset(self):
        self.metrics = {"contract": -1, "accepted": 0}
        base_obs = self.base_env.reset()
        self.obs = copy.deepcopy(base_obs)
        self.last_seen_obs = copy.deepcopy(base_obs)
        self.params = None
        self.contract_state = {k: 0 for k in self._agents}
        self._begin_episode()
        self.contract_param = self.negotiate()
        self.params = {k: self.contract_param for k in self._agents}
        self._set_theta(self.contract_param)
        return self._with_contract(self.obs, self._agents)

    def compute_vals(self, obs):
        vals = {}
        for key in obs:
            policy_id = "policy" if self.shared else key
            self.frozen_trainer.compute_single_action(obs[key], policy_id=policy_id)  # forward pass fills the value head
            vals[key] = self.frozen_trainer.get_policy(policy_id).model.value_function().item()
        return vals

    def _candidate_obs(self, param):
        tail = np.concatenate((param, np.array([0])))
        if self.convolutional:
            out = {}
            for key in self._agents:
                out[key] = self.obs[key]
                out[key].update({"contract": tail})
            return out
        return {key: np.concatenate((self.obs[key], tail)) for key in self.contract_state.keys()}

    def negotiate(self):
        params, vals = [self.contract_low], [self.compute_vals(self._candidate_obs(self.contract_low))]
        for _ in range(self.num_samples):
            params.append(self.contract_param_space.sample())
            vals.append(self.compute_vals(self._candidate_obs(params[-1])))
        return self.compute_best_param(vals, params)

    def compute_best_param(self, all_vals, all_params, dec_rule=None):
        rule = self.decision_rule if dec_rule is None else dec_rule
        if rule == "max":
            return all_params[int(np.argmax([sum(v.values()) for v in all_vals]))]
        if rule == "majority":
            null = all_vals[0]
            keep = [0]
            for v in all_vals[1:]:
                ayes = sum(1 for k in v if v[k] > null[k])
                if ayes >= len(v) - ayes:
                    keep.append(all_vals.index(v))  # first candidate with these values, as the reference resolves it
            return self.compute_best_param([all_vals[i] for i in keep], [all_params[i] for i in keep], dec_rule="max")
        return None


class SeparateContractCombinedStage(_ProposalStages):
    def __init__(self, base_env, contract, num_agents, convolutional, **kwargs):
        super().__init__(base_env, contract, num_agents, convolutional)
        self._external_theta(True)
        self.continuous_action_space = hasattr(self.base_env, "continuous_action_space")
        base_space = self.base_env.continuous_action_space if self.continuous_action_space else self.base_env.action_space
        self._base_width = base_space.shape[0]
        self.action_space = spaces.Box(low=np.concatenate((base_space.low, self.contract_low, np.array([0.0]))),
                                       high=np.concatenate((base_space.high, self.contract_high, np.array([1.0]))))

    def reset(self):
        return self._stage_reset(2)

    def step(self, acts):
        state = self.contract_state["a0"]
        if state == 2:
            self._propose(acts["a0"][self._base_width:-1], acts)
        elif state == 3:
            self._decide(acts)
        else:
            if self.continuous_action_space:  # logits -> one sampled discrete action per agent (np.random, as the reference)
                import scipy.special
                base_acts = {key: int(np.argmax(np.random.multinomial(1, scipy.special.softmax(acts[key][:self._base_width].astype(np.float64)))))
                             for key in acts.keys()}
            else:
                base_acts = {key: acts[key][:self._base_width] for key in acts.keys()}
            return super().step(base_acts)
        rews, infos = self._zero()
        return self._stage_obs(self.last_seen_obs, self.params), rews, {"__all__": False}, infos
