"""Experiment-config plumbing: the reference's JSON job format -> HIP-backed env objects.

The reference's `runner.py:89-135` expands a config file `[global, sweep, job, job, ...]` into one job
dict per (sweep combination x job x seed) and `utils/ray_config_utils.py:28-123` turns a job dict into a
base env + contract (+ wrapper).  This module restates that plumbing for the accelerated classes so a
reference config file can drive them unchanged (BASELINE config 1).  Only the
environment construction is mirrored — RLlib trainer/solver settings in the job dict are carried along
untouched for the caller."""
import copy
import itertools
import json

from ..contract import contract_list
from .env_creator_functions import env_creator, get_base_env_tag

SEED_MULTIPLIER = 73907  # runner.py:129


def expand_config_list(config_dict_list, seeds=1):
    """runner.py:89-135: global defaults, cartesian sweep, default experiment name, seeds"""
    config_dict_list = copy.deepcopy(config_dict_list)
    global_params, iter_params, jobs = config_dict_list[0], config_dict_list[1], config_dict_list[2:]
    for config in jobs:
        for param in global_params:
            if param not in config:
                config[param] = global_params[param]
    combos = [dict(zip(iter_params.keys(), c)) for c in itertools.product(*iter_params.values())]
    expanded = [{**config, **c} for c in combos for config in jobs]
    for config in expanded:
        if "experiment_name" not in config:
            config["experiment_name"] = config["environment"] + "-" + str(config["num_agents"]) + "agents"
    out = []
    for c in expanded:
        for i in range(seeds):
            c["seed"] = int((i + 1) * SEED_MULTIPLIER)
            out.append(copy.deepcopy(c))
    return out


def load_config_file(path, seeds=1):
    with open(path, "r") as f:
        return expand_config_list(json.load(f), seeds)


def _env_config(job):
    env_config = {"num_agents": job.get("num_agents"), "env_params": job.get("env_params", {})}
    env_config.update(job.get("env_args", {}))
    return env_config


def build_env(job, rng="global", device=0):
    """ray_config_utils.py:28-72,126-214: (top-level env, base env, contract) for one job dict, the first training
    stage of the job: `joint` -> JointEnv over the base env, `separate` -> the bare base env (the no-contract baseline),
    `combined` -> SeparateContractCombinedStage, otherwise SeparateContractSubgameStage."""
    contract_params = dict(job.get("contract_params", {}))
    contract_params["num_agents"] = job.get("num_agents")
    contract = getattr(contract_list, job.get("contract"))(**contract_params) if job.get("contract") else None
    env_config = _env_config(job)
    base_env = env_creator(get_base_env_tag(job), dict(env_config, rng=rng, device=device))
    convolutional = bool(job.get("env_args") and job["env_args"].get("image_obs"))
    if job.get("joint"):
        return env_creator("JointEnv", dict(env_config, base_env=base_env)), base_env, contract
    if job.get("separate") or contract is None:
        return base_env, base_env, contract
    tag = "ContractWrapperCombined" if job.get("combined") else "ContractWrapperSubgame"
    top = env_creator(tag, dict(env_config, base_env=base_env, contract=contract, convolutional=convolutional))
    return top, base_env, contract


def build_second_stage(job, base_env, contract, checkpoint_paths, trainer_config=None, trainer_factory=None):
    """ray_config_utils.py:217-278 (get_neg_config / get_solver_config): the contract-negotiation stage of a job over
    an already built base env — `solver: true` -> NegotiationSolver with the class defaults (50 sampled contracts,
    'majority' rule; the job's `solver_samples` is run_solver.py's number of evaluation episodes, not a constructor
    argument), otherwise SeparateContractNegotiateStage.  `trainer_config` is the
    frozen first-stage trainer config the reference copies into env_config; `trainer_factory` builds the frozen
    policies when RLlib's PPOTrainer is not what should be used."""
    convolutional = bool(job.get("env_args") and job["env_args"].get("image_obs"))
    env_config = dict(_env_config(job), base_env=base_env, contract=contract, convolutional=convolutional,
                      horizon=job.get("horizon"), trainer_config=trainer_config, trainer_env="ContractWrapperSubgame",
                      trainer_path=checkpoint_paths[0], shared=job.get("shared_policy"), trainer_factory=trainer_factory)
    if job.get("solver"):
        return env_creator("NegotiationSolver", env_config)
    return env_creator("ContractWrapperNegotiate", env_config)
