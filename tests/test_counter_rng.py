"""The counter-RNG mode (CE_FLAG_RNG_COUNTER, include/contracts_engine.h) on the CPU side: the oracle's Philox4x32-10 against
the published known-answer vectors and an independent numpy restatement, the stream definition (generations of 512 words,
a fresh generation per operation), the mode's scope — and the harness that pins the mode to the REFERENCE: the reference's own
code run with its np.random draw sites routed to this stream (tests/golden/counter_stream.py, make_counter_golden.py -> p_*
fixtures, replayed by tests/test_oracle_golden.py on the CPU and tests/test_gpu_parity.py::test_golden_grid_counter_rng on the
GPU).  Here: that harness's draw algorithms equal numpy's RandomState call for call, and its word stream is the documented one."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po

# Random123 kat_vectors, "philox4x32 10": counter, key, expected
KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def philox_np(key, counters):
    """Philox4x32-10 over an [N, 4] array of counters (numpy uint64 arithmetic, written from the paper's round function)"""
    c = np.asarray(counters, np.uint64).copy()
    k0, k1 = int(key[0]), int(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = (p1 >> np.uint64(32)) ^ c[:, 1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[:, 3] ^ np.uint64(k1)
        c = np.stack([n0, p1 & mask, n2, p0 & mask], axis=1)
        k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
    return c.astype(np.uint32)


def temper(y):
    y = np.asarray(y, np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9d2c5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xefc60000)
    y ^= y >> np.uint32(18)
    return y


def test_philox_known_answers():
    L = po.lib()
    for ctr, key, want in KAT:
        c, k, out = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
        L.orc_philox4x32_10(k.ctypes.data, c.ctypes.data, out.ctypes.data)
        assert tuple(int(x) for x in out) == want
        assert tuple(int(x) for x in philox_np(key, [ctr])[0]) == want


@pytest.mark.parametrize("seed", [0, 1, 12345, 0xffffffff])
def test_counter_stream_definition(seed):
    """stream word 512 (g - 1) + 4 q + j = temper(philox(counter = (q, g, 0, 0), key = (seed, 0))[j]), g = 1, 2, ..."""
    L = po.lib()
    count = 512 * 3 + 17
    out = np.zeros(count, np.uint32)
    L.orc_counter_stream(C.c_uint64(seed), out.ctypes.data, count)
    want = []
    for g in (1, 2, 3, 4):
        ctrs = np.zeros((128, 4), np.uint64)
        ctrs[:, 0] = np.arange(128)
        ctrs[:, 1] = g
        want.append(temper(philox_np((seed, 0), ctrs).reshape(-1)))
    assert np.array_equal(out, np.concatenate(want)[:count])


def _rollout(kind, n, E, T, seed0, same_actions=False, **kw):
    o = po.Oracle(kind, E, n, rng="counter", auto_reset=True, **kw)
    o.seed(seed0=seed0)
    o.reset()
    rs = np.random.RandomState(3)
    na = 9 if kw.get("firing") or kind == "cleanup" else 8
    gens = [o.rng[:, 2].copy()]
    for _ in range(T):
        a = rs.randint(0, na, size=(1 if same_actions else E, n)).astype(np.uint8)
        o.step(np.ascontiguousarray(np.broadcast_to(a, (E, n))))
        gens.append(o.rng[:, 2].copy())
    return o, np.array(gens)


def test_oracle_counter_mode_state_is_key_and_generation():
    o, gens = _rollout("cleanup", 4, 6, 40, 77, contract="cleanup", horizon=12)
    assert o.rng.shape == (6, 4)
    assert np.array_equal(o.rng[:, 0], 77 + np.arange(6)) and not o.rng[:, 1].any() and not o.rng[:, 3].any()
    d = np.diff(gens.astype(np.int64), axis=0)
    # every step opens at least one generation; a step that ends an episode also resets (shuffles, orientations, the spawn
    # draws and theta: more than 512 words with the step's own) and runs on into the next
    assert (d >= 1).all() and (d <= 3).all() and (d >= 2).any()


def test_oracle_counter_mode_is_deterministic_and_seed_dependent():
    a, _ = _rollout("harvest", 5, 4, 30, 5, same_actions=True, contract="harvest_local", horizon=1000)
    b, _ = _rollout("harvest", 5, 4, 30, 5, same_actions=True, contract="harvest_local", horizon=1000)
    c, _ = _rollout("harvest", 5, 4, 30, 6, same_actions=True, contract="harvest_local", horizon=1000)
    assert a.grid.tobytes() == b.grid.tobytes() and a.agents.tobytes() == b.agents.tobytes()
    assert a.grid.tobytes() != c.grid.tobytes()
    # env 1 of seed0 = 5 is env 0 of seed0 = 6: the stream depends on the env's seed only
    assert a.grid[1].tobytes() == c.grid[0].tobytes() and a.agents[1].tobytes() == c.agents[0].tobytes()


def test_counter_mode_differs_from_the_reference_stream_but_not_in_distribution():
    """not the reference's stream, the same process: under one random policy the apple stock and the rewards of 96 harvest
    envs after 200 steps agree between the two modes within sampling error"""
    E, n, T = 96, 5, 200
    stock, reward = {}, {}
    for mode in ("mt19937", "counter"):
        o = po.Oracle("harvest", E, n, rng=mode, horizon=1000)
        o.seed(seed0=1000)
        o.reset()
        rs = np.random.RandomState(8)
        total = np.zeros(E)
        for _ in range(T):
            o.step(rs.randint(0, 8, size=(E, n)).astype(np.uint8))
            total += o.base_reward.sum(axis=1)
        stock[mode] = (o.grid.reshape(E, -1) == 2).sum(axis=1).astype(float)  # CE_CELL_APPLE
        reward[mode] = total
    for stat in (stock, reward):
        a, b = stat["mt19937"], stat["counter"]
        assert not np.array_equal(a, b)
        se = np.sqrt(a.var() / E + b.var() / E)
        assert abs(a.mean() - b.mean()) < 4 * se, (a.mean(), b.mean(), se)


def test_counter_stream_key_is_the_whole_64_bit_seed():
    L = po.lib()
    seed = (5 << 32) | 77
    out = np.zeros(8, np.uint32)
    L.orc_counter_stream(C.c_uint64(seed), out.ctypes.data, 8)
    ctrs = np.zeros((2, 4), np.uint64)
    ctrs[:, 0] = [0, 1]
    ctrs[:, 1] = 1
    assert np.array_equal(out, temper(philox_np((77, 5), ctrs).reshape(-1)))
    o = po.Oracle("harvest", 2, 3, rng="counter")
    o.seed(np.array([seed, seed + 1], np.uint64))
    assert o.rng[:, :2].tolist() == [[77, 5], [78, 5]]


def test_counter_mode_belongs_to_the_grid_kinds():
    for kind, n in (("selfdrive", 4), ("harvest_features", 2), ("cleanup_features", 2)):
        with pytest.raises(RuntimeError):
            po.Oracle(kind, 2, n, rng="counter")


# ---- the harness that runs the REFERENCE over this stream (tests/golden/counter_stream.py, make_counter_golden.py) ----
import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import counter_stream as cs  # noqa: E402


def _mt19937_words(seed, count):
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)  # what np.random.seed(int) / RandomState(int) do
    return bg.random_raw(count)


@pytest.mark.parametrize("seed", [0, 7, 73907])
def test_legacy_draws_are_numpys_legacy_algorithms(seed):
    """LegacyDraws — what the p_* fixtures' generator routes the reference's np.random calls to — fed the raw MT19937 words
    returns exactly what np.random.RandomState returns, call for call, on the argument shapes of the reference's draw sites
    (map_env.py:546,685,821,831, cleanup_new.py:326,339, harvest_new.py:294, two_stage_train.py:163-164)"""
    rs = np.random.RandomState(seed)
    mine = cs.LegacyDraws(cs.WordList(_mt19937_words(seed, 40000)))
    lo32, hi32 = np.array([0.0], np.float32), np.array([0.2], np.float32)  # contract_low / contract_high as gym casts them
    for rnd in range(12):
        for length in (0, 1, 2, 3, 5, 8, 9, 20, 119):
            a, b = [[i, 2 * i] for i in range(length)], [[i, 2 * i] for i in range(length)]
            rs.shuffle(a)
            mine.shuffle(b)
            assert a == b
        movers = list(zip(["a%d" % i for i in range(7)], [[i, i + 1] for i in range(7)]))
        m2 = list(movers)
        rs.shuffle(movers)
        mine.shuffle(m2)
        assert movers == m2
        arr, arr2 = np.arange(24).reshape(12, 2), np.arange(24).reshape(12, 2)
        rs.shuffle(arr)
        mine.shuffle(arr2)
        assert np.array_equal(arr, arr2)
        for k in (222, 155, 1):
            assert np.array_equal(rs.rand(k), mine.rand(k))
        assert rs.rand() == mine.rand()
        assert [rs.randint(4) for _ in range(9)] == [mine.randint(4) for _ in range(9)]
        assert np.array_equal(rs.randint(0, 7, size=5), mine.randint(0, 7, size=5))
        u, v = rs.uniform(low=lo32, high=hi32), mine.uniform(low=lo32, high=hi32)
        assert u.dtype == v.dtype and u.shape == v.shape and np.array_equal(u, v)
        assert rs.uniform(-0.5, 2.25) == mine.uniform(-0.5, 2.25)
        assert np.array_equal(rs.random_sample(3), mine.random_sample(3))
    assert rs.randint(0, 2 ** 32, dtype=np.uint32) == mine.words.next32()  # both sides consumed the same number of words


def test_counter_words_is_the_documented_stream():
    """CounterWords (the Python side of the p_* fixtures) == the published Philox vectors, and == the oracle's stream"""
    for ctr, key, want in KAT:
        assert tuple(int(x) for x in cs.philox4x32_10(key, [ctr])[0]) == want
    L = po.lib()
    for seed in (0, 73907, (7 << 32) | 123456789):
        w = cs.CounterWords(seed)
        mine = [w.next32() for _ in range(512 * 2 + 9)]
        out = np.zeros(len(mine), np.uint32)
        L.orc_counter_stream(C.c_uint64(seed), out.ctypes.data, len(mine))
        assert mine == [int(x) for x in out]
        assert (w.k0, w.k1, w.gen) == (seed & 0xffffffff, seed >> 32, 3)
    w = cs.CounterWords(5)
    with w.op():
        first = w.next32()
    with w.op():  # an operation drops what the previous one left unread
        assert w.gen == 2 and w.next32() == int(cs.temper(cs.philox4x32_10((5, 0), [(0, 2, 0, 0)]))[0, 0]) != first


def test_patched_routes_every_global_draw_or_refuses():
    w = cs.CounterWords(11)
    before = np.random.get_state()[1].copy()
    with cs.patched(w):
        np.random.seed(12)
        assert (w.k0, w.gen) == (12, 0)
        with w.op():
            x = [1, 2, 3, 4, 5]
            np.random.shuffle(x)
            np.random.rand(3), np.random.randint(4), np.random.uniform(0, 1)
        assert w.consumed >= 4 + 6 + 1 + 2
        with pytest.raises(AssertionError):
            np.random.choice(3)
        with pytest.raises(AssertionError):
            np.random.random()
    assert np.array_equal(np.random.get_state()[1], before)  # restored, and the global MT19937 was never touched
