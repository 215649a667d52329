"""Host-side (numpy) mirror of the engine's synthetic action generator (ce_synth_actions): uniform i.i.d. actions from
a splitmix64 counter hash keyed (key, GLOBAL env index, t, agent) — SURVEY.md §8(d).  Device planes and host planes of
the same (key, env range, t range) are identical, which is what lets a CPU run replay exactly the inputs of a GPU run."""
import numpy as np

_M1, _M2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xD1B54A32D192ED03)
_M3, _M4 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def synth_hash(key, env, t, agent):
    """vectorised synth_hash of ce_device.h (uint64 wrap-around arithmetic); arguments broadcast"""
    with np.errstate(over="ignore"):
        env = np.asarray(env, np.uint64)
        t = np.asarray(t, np.uint64)
        agent = np.asarray(agent, np.uint64)
        z = np.uint64(key) ^ (env * _M1) ^ (((t << np.uint64(32)) | agent) * _M2)
        z = z + _M1
        z = (z ^ (z >> np.uint64(30))) * _M3
        z = (z ^ (z >> np.uint64(27))) * _M4
        return z ^ (z >> np.uint64(31))


def synth_actions_u8(key, env_base, num_envs, num_agents, t0, T, num_actions):
    """[T, E, n] uint8 action ids: ((hash >> 32) * num_actions) >> 32"""
    t = np.arange(t0, t0 + T, dtype=np.uint64)[:, None, None]
    e = (np.uint64(env_base) + np.arange(num_envs, dtype=np.uint64))[None, :, None]
    a = np.arange(num_agents, dtype=np.uint64)[None, None, :]
    h = synth_hash(key, e, t, a) >> np.uint64(32)
    return ((h * np.uint64(num_actions)) >> np.uint64(32)).astype(np.uint8)


def synth_actions_f32(key, env_base, num_envs, num_agents, t0, T):
    """[T, E, n] float32 accelerations in [-0.1, 0.1): 24 hash bits -> float32, as k_synth_f32"""
    t = np.arange(t0, t0 + T, dtype=np.uint64)[:, None, None]
    e = (np.uint64(env_base) + np.arange(num_envs, dtype=np.uint64))[None, :, None]
    a = np.arange(num_agents, dtype=np.uint64)[None, None, :]
    bits = (synth_hash(key, e, t, a) >> np.uint64(40)).astype(np.float32)
    return (bits * np.float32(1.0 / 16777216.0)) * np.float32(0.2) - np.float32(0.1)
