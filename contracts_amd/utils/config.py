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


def build_env(job, rng="global", device=0):
    """ray_config_utils.py:28-72,126-214: (top-level env, base env, contract) for one job dict.
    `separate: true` returns the bare base env (the no-contract baseline); otherwise the base env is
    wrapped in SeparateContractSubgameStage (the first training stage of every contracting config)."""
    contract_params = dict(job.get("contract_params", {}))
    contract_params["num_agents"] = job.get("num_agents")
    contract = getattr(contract_list, job.get("contract"))(**contract_params) if job.get("contract") else None
    env_config = {"num_agents": job.get("num_agents"), "env_params": job.get("env_params", {})}
    env_config.update(job.get("env_args", {}))
    base_env = env_creator(get_base_env_tag(job), dict(env_config, rng=rng, device=device))
    convolutional = bool(job.get("env_args") and job["env_args"].get("image_obs"))
    if job.get("joint") or job.get("combined"):
        raise NotImplementedError("joint / combined stages are outside the accelerated hot path (SURVEY.md §8f)")
    if job.get("separate") or contract is None:
        return base_env, base_env, contract
    top = env_creator("ContractWrapperSubgame", dict(env_config, base_env=base_env, contract=contract,
                                                     convolutional=convolutional))
    return top, base_env, contract
