"""The three contracts of the hot path, by the reference's names (contract/contract_list.py) so that
`getattr(contract.contract_list, cfg['contract'])(**contract_params)` (utils/ray_config_utils.py:30-32)
keeps working.  Here a contract is first of all a SPECIFICATION: its space and the id of the fused HIP epilogue that
computes the transfer inside the step kernel (ce_grid_kernels.hip / ce_selfdrive_kernels.hip) — the wrapper
(`SeparateContractSubgameStage`) never calls `compute_transfer` for them, and nothing in the engine's path does.

`compute_transfer(obs, acts, rews, params, infos)` is kept as the reference's public method for callers that use a
contract object on its own (inspection, unit tests, a user's own wrapper): a few dictionary operations on one step's
infos, returning the reference's `{agent: amount | (amount, {recipient: share})}` mapping."""
import numpy as np

from ..spaces import Box
from .contract import Contract


class CleanupContract(Contract):
    """theta in [0, 0.2]: payment per waste cell cleaned, paid evenly by the others (contract_list.py:7-27)."""
    engine_contract = "cleanup"

    def __init__(self, num_agents, low_val=0, high_val=0.2):
        super().__init__(Box(shape=(1,), low=low_val, high=high_val), np.array([0.0]), num_agents)

    def compute_transfer(self, obs, acts, rews, params, infos=None):
        # contract_list.py:22-27: the cleaner is PAID theta per cleaned square (a negative transfer)
        return {k: -params[k][0] * infos[k]["cleaned_squares"] for k in acts.keys()}


class HarvestFeaturemodLocalContract(Contract):
    """theta in [0, 10]: charged when eating an apple with < 4 apples within radius^2 5 (contract_list.py:29-54)."""
    engine_contract = "harvest_local"

    def __init__(self, num_agents, low_val=0, high_val=10.0):
        super().__init__(Box(shape=(1,), low=low_val, high=high_val), np.array([0.0]), num_agents)

    def compute_transfer(self, obs, acts, rews, params, infos=None):
        # contract_list.py:45-54: feature_obs[8] = apples close to the agent now, eaten_close_apples = ate in a sparse spot
        out = {}
        for k in acts.keys():
            sparse = infos[k]["feature_obs"][8] < 4 and infos[k]["eaten_close_apples"] > 0
            out[k] = params[k][0] if sparse else 0
        return out


class SelfdriveContractDistprop(Contract):
    """theta in [0, 100]: per-unit-distance subsidy at the ambulance's merge (contract_list.py:56-102)."""
    engine_contract = "selfdrive_distprop"

    def __init__(self, num_agents):
        super().__init__(Box(shape=(1,), low=0, high=100.0), np.array([0.0]), num_agents)

    def compute_transfer(self, obs, acts, rews, params, infos=None):
        # contract_list.py:69-102.  The relative positions of the other cars sit at obs['a0'][2 + i]; the loop bound
        # len(obs) // 2 = n + 2 also visits two velocity slots (>= 0, never "behind"), which only adds zero entries for
        # agent names that do not exist — kept, the wrapper ignores them.
        width = len(list(obs.values())[0]) // 2
        theta = params["a0"][0] if "a0" in params else 0.0
        out = {"a0": 0}
        if "a0" in acts.keys() and infos["a0"]["just_passed"]:
            rel = obs["a0"]
            behind = ["a%d" % i for i in range(1, width) if rel[2 + i] < 0]
            if behind:
                gap = {k: -rel[2 + int(k[-1])] for k in behind}
                total = 0
                for k in behind:
                    total += gap[k]
                out["a0"] = (theta * total, {k: gap[k] / total for k in behind})
            for i in range(1, width):
                k = "a%d" % i
                if k in acts.keys() and k not in behind:
                    out[k] = (theta * rel[2 + i], {"a0": 1})
        for i in range(1, width):
            out.setdefault("a%d" % i, 0)
        return out
