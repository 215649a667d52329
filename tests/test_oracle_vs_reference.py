"""Live cross-check of the CPU oracle against the upstream reference, imported where it lies (/root/reference) through
tests/golden/ref_harness.py.  Runs only where the reference is present (the build container); on the GPU box it is
skipped and the committed fixtures carry the pin.  Seeds here are NOT the fixtures' seeds: every run of the suite in
the container re-derives fresh short traces from the reference itself and replays them through the oracle."""
import os
import sys

import numpy as np
import pytest

import golden_check as gc

sys.path.insert(0, gc.GOLDEN_DIR)
from ref_harness import reference_available  # noqa: E402

pytestmark = pytest.mark.skipif(not reference_available(), reason="upstream reference not present (GPU box)")


@pytest.fixture(scope="module")
def R():
    from ref_harness import load_reference
    return load_reference()


@pytest.mark.parametrize("kind,n,firing,contract,seed", [("cleanup", 3, False, True, 910001), ("cleanup", 6, True, False, 910002),
                                                         ("harvest", 4, False, True, 910003), ("harvest", 2, True, True, 910004)])
def test_oracle_matches_live_grid_reference(R, kind, n, firing, contract, seed):
    import make_golden
    from oracle.pyoracle import Oracle
    g = make_golden.run_grid_trace(R, kind=kind, n=n, seed=seed, T=45, firing=firing, contract=contract, store_obs_steps=45,
                                   action_p=[.1, .1, .15, .1, .05, .1, .1, .3] + ([0.0] if firing else []) if kind == "cleanup" else None)
    kind2, n2, kw = gc.grid_kwargs(g)
    orc = Oracle(kind2, 1, n2, **kw)
    gc.replay_grid(g, orc)
    orc.close()


@pytest.mark.parametrize("kind,n,contract,seed", [("harvest_features", 3, True, 920001), ("cleanup_features", 4, True, 920002)])
def test_oracle_matches_live_feature_reference(R, kind, n, contract, seed):
    import make_feat_golden
    from oracle.pyoracle import Oracle
    g = make_feat_golden.run_trace(R, kind, n, seed, [40, 15], contract=contract, horizon=40,
                                   action_p=[.1, .1, .15, .1, .05, .1, .1, .3] if kind == "cleanup_features" else None)
    kind2, n2, kw = gc.feat_kwargs(g)
    orc = Oracle(kind2, 1, n2, **kw)
    gc.replay_feat(g, orc)
    orc.close()


# ---- the counter-RNG mode: the reference run over the engine's stream (tests/golden/make_counter_golden.py) ----
class _MTWords:
    """MT19937 behind the counter harness's word-source interface: with THIS source the patched reference must reproduce an
    ordinary np.random.seed trace — i.e. the patching reaches every draw site of the hot path and LegacyDraws is numpy"""

    def __init__(self):
        self.bg = np.random.MT19937()

    def seed(self, seed):
        self.bg._legacy_seeding(int(seed))

    def next32(self):
        return int(self.bg.random_raw())

    def op(self):
        import contextlib
        return contextlib.nullcontext()

    def state_row(self):
        import hashlib
        st = self.bg.state["state"]
        return np.array([st["pos"], int(hashlib.sha256(st["key"].tobytes()).hexdigest()[:8], 16)], np.int64)


@pytest.mark.parametrize("name", ["g7_cleanup_n8_s1", "g7_harvest_n8_s2", "g4_harvest_n5_fire"])
def test_patched_reference_over_mt19937_words_reproduces_the_plain_fixture(R, name):
    """the harness that records the p_* fixtures, fed MT19937 words instead of the counter stream, regenerates a committed
    g* fixture key for key — generator position and fingerprint after every step included"""
    import counter_stream as cs
    import environments.cleanup_new as ref_cleanup
    import environments.harvest_new as ref_harvest
    import make_golden
    want = gc.load(name)
    kw = dict(kind=str(want["kind"]), n=int(want["n"]), seed=int(want["seed"]), T=len(want["actions"]),
              firing=bool(int(want["firing"])), store_obs_steps=len(want["obs"]),
              action_p=[.1, .1, .15, .1, .05, .1, .1, .3] if name == "g7_cleanup_n8_s1" else None)
    words = _MTWords()
    with cs.patched(words, modules=(ref_cleanup, ref_harvest)):
        got = make_golden.run_grid_trace(R, stream=words, **kw)
    for k in want.files:
        a, b = want[k], got[k]
        assert a.shape == np.asarray(b).shape and (a == b).all(), k


@pytest.mark.parametrize("kind,n,firing,contract,seed", [("cleanup", 5, False, True, (3 << 32) | 930001), ("harvest", 6, True, True, 930002),
                                                         ("cleanup", 2, True, False, 930003)])
def test_oracle_counter_mode_matches_live_reference_over_the_counter_stream(R, kind, n, firing, contract, seed):
    import counter_stream as cs
    import environments.cleanup_new as ref_cleanup
    import environments.harvest_new as ref_harvest
    import make_golden
    from oracle.pyoracle import Oracle
    stream = cs.CounterWords()
    with cs.patched(stream, modules=(ref_cleanup, ref_harvest)):
        g = make_golden.run_grid_trace(R, kind=kind, n=n, seed=seed, act_seed=seed & 0xffff, T=[30, 30, 12], episodes=3,
                                       firing=firing, contract=contract, store_obs_steps=72, extra_env_kwargs=dict(horizon=30),
                                       action_p=[.1, .1, .15, .1, .05, .1, .1, .3] + ([0.0] if firing else []) if kind == "cleanup" else None,
                                       stream=stream)
    kind2, n2, kw = gc.grid_kwargs(g)
    orc = Oracle(kind2, 1, n2, rng="counter", **kw)
    gc.replay_grid(g, orc)
    orc.close()
