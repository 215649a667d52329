"""ce_download_obs_f64: the observation leg of a dict-protocol tick (copy into a page-locked block in parts + float64
conversion on the host pool) against the oracle's views / 255 — whole batch and slices, every part count, both grid
kinds; and its argument checks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n", [("cleanup", 8), ("harvest", 3)])
def test_download_obs_f64_matches_the_oracle_views(kind, n):
    from contracts_amd.engine import BatchedEnv
    from oracle.pyoracle import Oracle
    E = 1100
    env, orc = BatchedEnv(kind, E, n, horizon=50, auto_reset=True), Oracle(kind, E, n, horizon=50, auto_reset=True)
    seeds = np.arange(E, dtype=np.uint64) * 7 + 3
    rs = np.random.RandomState(2)
    for o in (env, orc):
        o.seed(seeds)
        o.reset()
    for _ in range(5):
        a = rs.randint(0, env.num_actions, size=(E, n)).astype(np.uint8)
        env.step(a)
        orc.step(a)
    want = orc.obs.reshape(E, n, 15, 15, 3) / 255.0  # the reference's float image (cleanup_new.py:258, harvest_new.py:229)
    staging = env.host_alloc((E, env.b.obs_env_stride), np.uint8)
    for parts in (1, 2, 4, 7, 16, 50):
        for threads in (1, 5, 16):
            out = np.full((E, n, 15, 15, 3), -1.0)
            env.download_obs_f64(staging, out, threads, parts=parts)
            assert np.array_equal(out, want), (parts, threads)
    lo, cnt = 301, 77  # a slice on its own: staging and out hold just the slice
    out = np.full((cnt, n, 15, 15, 3), -1.0)
    env.download_obs_f64(staging[:cnt], out, 4, parts=3, env_begin=lo, env_count=cnt)
    assert np.array_equal(out, want[lo:lo + cnt])
    out1 = np.full((1, n, 15, 15, 3), -1.0)  # fewer envs than parts
    env.download_obs_f64(staging[:1], out1, 8, parts=4, env_begin=E - 1, env_count=1)
    assert np.array_equal(out1, want[E - 1:])


def test_download_obs_f64_argument_checks():
    from contracts_amd import _lib
    from contracts_amd.engine import BatchedEnv
    env = BatchedEnv("cleanup", 16, 2)
    env.seed(seed0=1)
    env.reset()
    staging = env.host_alloc((16, env.b.obs_env_stride), np.uint8)
    out = np.zeros((16, 2, 15, 15, 3))
    with pytest.raises(_lib.EngineError):
        env.download_obs_f64(staging[:8], out, 4)  # staging too small
    with pytest.raises(_lib.EngineError):
        env.download_obs_f64(staging, out[:8], 4)  # out too small
    with pytest.raises(_lib.EngineError):
        env.download_obs_f64(staging, out, 4, env_begin=10, env_count=7)  # slice out of range
    with pytest.raises(_lib.EngineError):
        env.download_obs_f64(staging, out, 0)  # no threads
    feat = BatchedEnv("harvest_features", 16, 2)
    feat.seed(seed0=1)
    feat.reset()
    with pytest.raises(_lib.EngineError):
        feat.download_obs_f64(staging, out, 4)  # feature kinds have no image observations


def test_downloads_on_another_stream_are_ordered_after_a_reset():
    """ce_reset on one stream followed at once by ce_download_async / ce_download_obs_f64 on ANOTHER non-blocking stream: the
    copies must see the reset's results (the engine orders the second stream behind the reset itself, as it does for steps —
    ADVICE r04: the two download entry points skipped that).  A large batch, so that the reset kernel is still running when the
    host issues the copies; a few steps first so that the buffers hold something else."""
    import torch
    from contracts_amd.engine import BatchedEnv
    E, n = 60000, 8
    env = BatchedEnv("cleanup", E, n, horizon=1000)
    env.seed(seed0=5)
    env.reset()
    a = np.full((E, n), 4, np.uint8)
    for _ in range(3):
        env.step(a)
    env.synchronize()
    assert env.download("timestep").min() == 3
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ts = env.host_alloc((E,), np.int32)
    staging = env.host_alloc((E, env.b.obs_env_stride), np.uint8)
    out = np.full((E, n, 15, 15, 3), -1.0)
    ts[:] = -1
    env.reset(stream=s1.cuda_stream)                       # asynchronous, ~0.2 ms of kernel
    env.download_async("timestep", ts, stream=s2.cuda_stream)   # issued while it runs
    env.download_obs_f64(staging, out, 8, parts=4, stream=s2.cuda_stream)
    env.synchronize(s2.cuda_stream)
    assert (ts == 0).all(), "the copy on the second stream overtook the reset"
    torch.cuda.synchronize()
    want = env.download("obs").reshape(E, n, 15, 15, 3) / 255.0   # the reset observation, read after everything has drained
    assert np.array_equal(out, want)
    env.close()
