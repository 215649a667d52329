"""The three contracts of the hot path, by the reference's names (contract/contract_list.py) so that
`getattr(contract.contract_list, cfg['contract'])(**contract_params)` (utils/ray_config_utils.py:30-32)
keeps working.  Here a contract is a SPECIFICATION: its space and the id of the fused HIP epilogue that
computes the transfer inside the step kernel (ce_grid_kernels.hip / ce_selfdrive_kernels.hip).  There is
deliberately no host implementation of `compute_transfer`: the product has no CPU path."""
import numpy as np

from ..spaces import Box
from .contract import Contract


class _FusedContract(Contract):
    def compute_transfer(self, obs, acts, rews, params, infos=None):
        raise NotImplementedError(
            "%s.compute_transfer is fused into the engine's step kernel (contract id %r); wrap the base env "
            "in SeparateContractSubgameStage and call step()" % (type(self).__name__, self.engine_contract))


class CleanupContract(_FusedContract):
    """theta in [0, 0.2]: payment per waste cell cleaned, paid evenly by the others (contract_list.py:7-27)."""
    engine_contract = "cleanup"

    def __init__(self, num_agents, low_val=0, high_val=0.2):
        super().__init__(Box(shape=(1,), low=low_val, high=high_val), np.array([0.0]), num_agents)


class HarvestFeaturemodLocalContract(_FusedContract):
    """theta in [0, 10]: charged when eating an apple with < 4 apples within radius^2 5 (contract_list.py:29-54)."""
    engine_contract = "harvest_local"

    def __init__(self, num_agents, low_val=0, high_val=10.0):
        super().__init__(Box(shape=(1,), low=low_val, high=high_val), np.array([0.0]), num_agents)


class SelfdriveContractDistprop(_FusedContract):
    """theta in [0, 100]: per-unit-distance subsidy at the ambulance's merge (contract_list.py:56-102)."""
    engine_contract = "selfdrive_distprop"

    def __init__(self, num_agents):
        super().__init__(Box(shape=(1,), low=0, high=100.0), np.array([0.0]), num_agents)
