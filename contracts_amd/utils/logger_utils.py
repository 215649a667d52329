"""MetricsCallback — the per-episode metrics hook of the reference's utils/logger_utils.py:95-150: at the end of every
episode the env's `metrics` dictionary is copied into `episode.custom_metrics`, which is how total_apples_eaten,
transfers, equality, ... reach RLlib's result dict.  The wandb / csv logger around it is out of scope.

The reference reads `base_env._unwrapped_env[.base_env].metrics` — one Python env per BaseEnv.  Here a BaseEnv may be
the batched hook (`contracts_amd.vector_env.BatchedBaseEnv`, E sub-envs per handle), so the sub-env of the finished
episode is looked up by the `env_index` RLlib passes; the single-env layouts of the reference keep working."""

try:  # pragma: no cover - RLlib is absent in the build image
    from ray.rllib.agents.callbacks import DefaultCallbacks as _Callbacks
except Exception:
    _Callbacks = object


def episode_env_metrics(base_env, env_index=None):
    """wrapper metrics merged over base-env metrics, as the reference does (m1.update(m2))"""
    env = None
    if hasattr(base_env, "get_sub_environments"):
        subs = base_env.get_sub_environments()
        if len(subs):
            env = subs[0 if env_index is None else env_index]
    if env is None:
        env = getattr(base_env, "_unwrapped_env", None)
    if env is None:
        raise AssertionError("no sub-environment to read metrics from")
    inner = getattr(getattr(env, "base_env", None), "metrics", None)
    outer = getattr(env, "metrics", None)
    assert inner is not None or outer is not None
    metrics = dict(inner) if inner is not None else dict(outer)
    if inner is not None and outer is not None:
        metrics.update(outer)
    return metrics


class MetricsCallback(_Callbacks):
    def on_episode_start(self, *, worker=None, base_env=None, policies=None, episode=None, **kwargs):
        episode.custom_metrics = {}

    def on_episode_step(self, *, worker=None, base_env=None, policies=None, episode=None, **kwargs):
        pass

    def on_episode_end(self, *, worker=None, base_env=None, policies=None, episode=None, env_index=None, **kwargs):
        for k, v in episode_env_metrics(base_env, env_index).items():
            episode.custom_metrics[k] = v
