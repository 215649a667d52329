"""Contract specification base class.

Interface parity with the reference's `contract/contract.py:1-9`: a contract object exposes
`contract_space`, `default_contract`, `num_agents`, `features_compute` and a `compute_transfer` hook.
In this package a contract is a *specification* consumed by the engine (which fused epilogue to run, over
which parameter interval); the arithmetic itself lives in the HIP step kernels.
"""
import numpy as np


class Contract:
    #: name of the engine's fused epilogue implementing this contract (None = not accelerated)
    engine_contract = None

    def __init__(self, contract_space, default_contract, num_agents, features_compute=None):
        if num_agents is None or int(num_agents) < 1:
            raise ValueError("a contract needs the number of agents, got %r" % (num_agents,))
        low, high = np.asarray(contract_space.low), np.asarray(contract_space.high)
        if low.shape != high.shape or np.any(high < low):
            raise ValueError("malformed contract space: low=%r high=%r" % (low, high))
        self.num_agents = int(num_agents)
        self.contract_space = contract_space
        self.default_contract = np.asarray(default_contract, dtype=np.float64)
        self.features_compute = features_compute

    # ---- what the engine needs -------------------------------------------------------
    def engine_spec(self, null_prob=0.0):
        """(epilogue name, low, high, null_prob) with the Box's float32 bounds read back as float64 —
        the values `np.random.uniform(low=contract_low, high=contract_high)` sees in the reference
        (environments/two_stage_train.py:39-40,164)."""
        return (self.engine_contract, float(self.contract_space.low[0]), float(self.contract_space.high[0]),
                float(null_prob))

    def fused_epilogue(self):
        """name of the step kernel's epilogue that computes THIS object's transfers, or None = host protocol
        (`compute_transfer` called every step, as the reference does).  A subclass of a shipped contract that overrides
        `compute_transfer` no longer is what the epilogue computes, so it falls back to the host protocol instead of having
        its override silently ignored."""
        name = self.engine_contract
        if name is None:
            return None
        for cls in type(self).__mro__:
            if "engine_contract" in cls.__dict__:  # the class that declared the epilogue owns the arithmetic it stands for
                return name if type(self).compute_transfer is cls.__dict__.get("compute_transfer") else None
        return None

    def compute_transfer(self, obs, acts, params, infos=None):
        raise NotImplementedError

    def __repr__(self):
        return "%s(num_agents=%d, theta in [%g, %g], epilogue=%r)" % (
            type(self).__name__, self.num_agents, float(self.contract_space.low[0]),
            float(self.contract_space.high[0]), self.engine_contract)
