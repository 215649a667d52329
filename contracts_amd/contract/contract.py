"""Contract base class — same constructor and attributes as the reference's contract/contract.py:1-9."""


class Contract:
    #: which fused epilogue of the engine implements `compute_transfer` (None = not accelerated)
    engine_contract = None

    def __init__(self, contract_space, default_contract, num_agents, features_compute=None):
        self.contract_space = contract_space
        self.default_contract = default_contract
        self.features_compute = features_compute
        self.num_agents = num_agents

    def compute_transfer(self, obs, acts, params, infos=None):
        raise NotImplementedError
