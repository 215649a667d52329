"""Batched second-stage evaluation — what the reference's `run_solver.py:18-73` does one episode at a time.

The reference resets a `NegotiationSolver` (which picks a contract parameter), then rolls one whole episode with the
frozen subgame policies, `solver_samples` times in sequence, and logs the episode reward and the contract of each.
Here the K (contract, seed) pairs are K env replicas of ONE engine handle under CE_FLAG_EXTERNAL_THETA — every replica
carries its own contract parameter — so an evaluation sweep is `horizon` kernel launches instead of K x horizon Python
steps ("batched evaluation of K sampled contracts x E envs on device").

`act_fn(obs, theta, t)` is the frozen policy: `obs` is the uint8 observation stack [K, n, 15, 15, 3] (grid kinds; divide
by 255 for the reference's float view) or the feature rows [K, n, F] (feature kinds), `theta` the per-replica contract
parameters [K]; it returns the integer actions [K, n].  For 'selfdrive' `obs` is the float64 rows [K, n, 2n + 7] and
the actions are accelerations; a car that is done stops acting (its key leaves the reference's dictionaries, so its
action is ignored and its reward is not counted) and a replica whose episode ended idles while the others finish.
Rewards are accumulated in the reference's order (step by step,
agent by agent) so the totals are the same doubles `run_solver` would log."""
import numpy as np

from . import _lib
from .engine import BatchedEnv

_CONTRACT_KIND = {"CleanupContract": "cleanup", "HarvestFeaturemodLocalContract": "harvest_local",
                  "SelfdriveContractDistprop": "selfdrive_distprop"}


def evaluate_contracts(kind, num_agents, contract, thetas, seeds, act_fn, horizon=1000, device=0, **engine_kwargs):
    """Rolls one episode per (theta, seed) pair, all of them together.

    contract: an engine contract name ('cleanup' / 'harvest_local'), a reference class name, or a contract object
    with `.engine_contract`.  Seeds follow the reference's seeding: replica k behaves like a process that called
    np.random.seed(seeds[k]), constructed the env and reset it.  Returns a dict with `ep_rewards` [K] (sum over agents
    and steps of the transferred rewards, run_solver.py:63-64), `agent_rewards` [K, n], `contract_param` [K], `steps`
    and the reference's summary statistics (`mean reward`, `std reward`, `mean contract`, `std contract`, :69-70)."""
    thetas = np.ascontiguousarray(thetas, np.float64).reshape(-1)
    seeds = np.ascontiguousarray(seeds, np.uint64).reshape(-1)
    if thetas.shape != seeds.shape:
        raise ValueError("one seed per contract parameter")
    name = getattr(contract, "engine_contract", contract)
    name = _CONTRACT_KIND.get(name, name)
    K = len(thetas)
    env = BatchedEnv(kind, K, num_agents, contract=name, horizon=horizon, external_theta=True, auto_reset=False,
                     device=device, **engine_kwargs)
    try:
        env.seed(seeds)
        env.upload("theta", thetas)
        env.reset()
        feat, cars = kind in _lib.FEAT_KINDS, kind == "selfdrive"
        obs_field = "obs_f64" if cars else "features" if feat else "obs"
        acting = np.ones((K, num_agents), bool)  # selfdrive: run_solver's active_agents, per replica
        agent_rewards = np.zeros((K, num_agents))
        ep_rewards = np.zeros(K)
        steps = 0
        done = np.zeros(K, bool)
        while not done.all() and steps < horizon:
            acts = np.asarray(act_fn(env.download(obs_field), thetas, steps)).reshape(K, num_agents)
            if cars:
                env.step(acts.astype(np.float32), (acting & ~done[:, None]).astype(np.uint8))
                r = np.where(acting & ~done[:, None], env.download("reward"), 0.0)
            else:
                env.step(acts.astype(np.uint8))
                r = env.download("reward")
            for a in range(num_agents):  # ep_rewards += r[key] for key in env_obs, in key order
                ep_rewards += r[:, a]
            agent_rewards += r
            if cars:
                acting &= ~env.download("done_agents").astype(bool)
            done = env.download("done").astype(bool)
            steps += 1
        faults = env.download("error_flags").copy()
        if cars:  # replicas that finished early were sent all-inactive steps: that is the idle bit, not a fault
            faults[done] &= ~np.uint32(_lib.FAULT_STEP_AFTER_DONE)
        if faults.any():
            raise _lib.EngineError("env faults: %s" % {int(i): int(faults[i]) for i in np.nonzero(faults)[0][:8]})
    finally:
        env.close()
    return {"ep_rewards": ep_rewards, "agent_rewards": agent_rewards, "contract_param": thetas.copy(), "steps": steps,
            "mean reward": float(np.mean(ep_rewards)), "std reward": float(np.std(ep_rewards)),
            "mean contract": float(np.mean(thetas)), "std contract": float(np.std(thetas))}
