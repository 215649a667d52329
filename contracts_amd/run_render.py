"""Rendered episodes of the grid envs — the loop of the reference's run_render.py:13-77 without the Ray trainer: a
policy is any callable `obs -> action` (or an object with RLlib's compute_single_action); frames are
`full_map_to_colors()` of every step (map + agents + the step's FIRE / CLEAN beams, composed from device state),
upscaled 20x and written with environments.env_utils.make_video_from_rgb_imgs."""
from .environments.env_utils import make_video_from_rgb_imgs


def _act(policy, obs, key):
    if hasattr(policy, "compute_single_action"):
        return policy.compute_single_action(obs, policy_id=key)
    return policy(obs)


def run_rendering(env, policy, store_path, num_renders=10, horizon=None, upscale=20):
    """env: a grid adapter (or a contract wrapper around one); returns the paths of the files written"""
    paths = []
    base = env if hasattr(env, "full_map_to_colors") else env.base_env
    horizon = horizon or getattr(base, "horizon", 1000)
    for j in range(num_renders):
        obs = env.reset()
        dones = {"__all__": False}
        active = list(obs.keys())
        imgs, ep_rewards, steps, info = [], 0, 0, {}
        while not dones["__all__"] and steps < horizon:
            imgs.append(base.full_map_to_colors())
            obs, r, dones, info = env.step({k: _act(policy, obs[k], k) for k in active})
            for key in dones:  # clear inactive agents (run_render.py:57-59)
                if dones[key] and key in active:
                    active.remove(key)
            steps += 1
            ep_rewards += sum(r[k] for k in obs)
        height, width, _ = imgs[0].shape
        name = "trajectory_{}_r{}".format(j, ep_rewards)
        if info.get("a0") and "contract_param" in info["a0"]:
            name += "_c{}".format(info["a0"]["contract_param"])
        paths.append(make_video_from_rgb_imgs(imgs, store_path, video_name=name, resize=(width * upscale, height * upscale)))
    return paths
