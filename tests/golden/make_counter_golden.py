#!/usr/bin/env python3
"""Reference fixtures for the counter-RNG mode (CE_FLAG_RNG_COUNTER): `p_*.npz`.

The mode's stream is the engine's own, so no `np.random.seed` trace of the reference reproduces it — but the REFERENCE can be
run over that stream: this script imports it from /root/reference (ref_harness.py), routes the process-global `np.random`
functions its hot path calls (map_env.py:546,685,821,831, cleanup_new.py:326,339, harvest_new.py:294,
two_stage_train.py:163-164 — and the `rand` names cleanup_new / harvest_new bound at import) to `counter_stream.LegacyDraws`
over `counter_stream.CounterWords`, and records traces with the same `run_grid_trace` that makes the `g*` fixtures, the
constructor, every `reset()` and every `step()` being one operation on the stream.  What then has to agree with the fixture —
oracle (tests/test_oracle_golden.py) and HIP (tests/test_gpu_parity.py) in `rng="counter"` mode — is everything the `g*`
fixtures compare, with the (key0, key1, generation) row in the place of the MT19937 fingerprint.

Build-container only.  Usage:  python tests/golden/make_counter_golden.py [name ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from counter_stream import CounterWords, patched  # noqa: E402
from make_golden import CLEANUP_MID, HARVEST_SMALL, run_grid_trace  # noqa: E402
from ref_harness import load_reference  # noqa: E402

CLEANER = [.1, .1, .15, .1, .05, .1, .1, .3]  # CLEAN-heavy: the waste density crosses both spawn thresholds


def jobs():
    S0 = 73907
    big = (7 << 32) | 123456789  # a 64-bit seed: both key words in use
    return {
        # two episodes across a reset(): waste shuffle (> 512 words in one step) + the reset's shuffles, orientations, theta
        "p1_cleanup_n4_cleaner": dict(kind="cleanup", n=4, seed=S0 + 40, T=[1000, 150], episodes=2, store_obs_steps=40,
                                      action_p=CLEANER),
        "p3_cleanup_n8_cleaner": dict(kind="cleanup", n=8, seed=S0 + 41, T=[1000, 100], episodes=2, store_obs_steps=40,
                                      action_p=CLEANER),
        "p2_harvest_n8": dict(kind="harvest", n=8, seed=S0 + 42, T=[1000, 60], episodes=2, store_obs_steps=40),
        "p4_cleanup_n8_fire": dict(kind="cleanup", n=8, seed=big, act_seed=S0 + 43, T=300, firing=True, store_obs_steps=30),
        "p4_harvest_n5_fire": dict(kind="harvest", n=5, seed=S0 + 44, T=250, firing=True, store_obs_steps=30),
        # short horizon: nine episodes, i.e. nine resets drawing on the persistent spawn / waste lists
        "p5_cleanup_n8_short": dict(kind="cleanup", n=8, seed=S0 + 45, T=[25] * 9, episodes=9, store_obs_steps=10,
                                    action_p=CLEANER, extra_env_kwargs=dict(horizon=25)),
        "p5_harvest_n3_short_nocontract": dict(kind="harvest", n=3, seed=S0 + 46, T=[40] * 5, episodes=5, store_obs_steps=10,
                                               contract=False, extra_env_kwargs=dict(horizon=40)),
        "p6_cleanup_n1": dict(kind="cleanup", n=1, seed=S0 + 47, T=200, store_obs_steps=10, action_p=CLEANER),
        # a hand-made layout (the reference's ascii_map argument) on the counter stream: 90 waste cells, four episodes
        "p7_cleanup_custom_n4": dict(kind="cleanup", n=4, seed=S0 + 48, T=[60, 60, 60, 30], episodes=4, store_obs_steps=10, action_p=CLEANER,
                                     extra_env_kwargs=dict(ascii_map=CLEANUP_MID, horizon=60)),
        "p7_harvest_custom_n5": dict(kind="harvest", n=5, seed=S0 + 49, T=[80, 50], episodes=2, store_obs_steps=10, firing=True,
                                     extra_env_kwargs=dict(ascii_map=HARVEST_SMALL, horizon=80)),
    }


def main():
    R = load_reference()
    import environments.cleanup_new as ref_cleanup
    import environments.harvest_new as ref_harvest
    only = set(sys.argv[1:])
    for name, kw in jobs().items():
        if only and name not in only:
            continue
        stream = CounterWords()
        with patched(stream, modules=(ref_cleanup, ref_harvest)):
            out = run_grid_trace(R, stream=stream, **kw)
        out["words_consumed"] = np.int64(stream.consumed)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        gens = out["mt"][:, 2]
        print("%-32s steps=%5d  generations=%6d  words=%8d  %7.1f KB" % (name, len(out["actions"]), int(gens[-1]), stream.consumed,
                                                                          os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
