"""Registry: the reference's env tags (utils/env_creator_functions.py:12-60) -> HIP-backed classes.
`register_env` is called when RLlib is importable, so `runner.py`-style configs resolve to these."""
from ..environments.cleanup_new import CleanupEnv
from ..environments.feature_envs import CleanupFeatures, HarvestFeatures
from ..environments.harvest_new import HarvestEnv
from ..environments.self_driving_car_accelerate import SelfAcceleratingCarEnv
from ..environments.two_stage_train import (JointEnv, NegotiationSolver, SeparateContractCombinedStage,
                                            SeparateContractNegotiateStage, SeparateContractSubgameStage)

_ACCELERATED = {
    "SelfDrive": SelfAcceleratingCarEnv,
    "HarvestNew": HarvestEnv,
    "CleanupNew": CleanupEnv,
    "ContractWrapperSubgame": SeparateContractSubgameStage,
    "JointEnv": JointEnv,
    "ContractWrapperNegotiate": SeparateContractNegotiateStage,
    "ContractWrapperCombined": SeparateContractCombinedStage,
    "NegotiationSolver": NegotiationSolver,
    "Harvest": HarvestFeatures,   # `harvest`: the feature-vector env of BASELINE config 0 (harvest_features.py)
    "Cleanup": CleanupFeatures,   # `cleanup` (cleanup_features.py)
}


def env_creator(name, config):
    if name in _ACCELERATED:
        return _ACCELERATED[name](**config)
    raise ValueError("Environment not found")


def get_base_env_tag(arg_dict):
    env = arg_dict.get("environment")
    tags = {"selfdrive": "SelfDrive", "harvest": "Harvest", "harvest_new": "HarvestNew", "cleanup": "Cleanup",
            "cleanup_new": "CleanupNew"}
    assert env in tags
    return tags[env]


try:  # pragma: no cover - RLlib is absent in the build image
    from ray.tune.registry import register_env
    for _tag in _ACCELERATED:
        register_env(_tag, lambda config, _t=_tag: env_creator(_t, config))
except Exception:
    pass
