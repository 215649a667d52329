"""CPU: the host-side pieces of the vector hook's recycled dict protocol — the threaded format conversions of the C-ABI
(`ce_obs_u8_to_f64`, `ce_i16_to_f64`: host functions, no device involved) and the CPython marshalling loops
(`contracts_amd/csrc/ce_pydict.c`) — against plain numpy / Python restatements."""
import numpy as np
import pytest


def test_obs_u8_to_f64_is_numpy_division():
    """the reference's float image is uint8 / 255 in float64 (cleanup_new.py:258, harvest_new.py:229)"""
    from contracts_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(0)
    E, n, row, agent = 37, 5, 48, 720
    pitched = rs.randint(0, 256, size=(E, n * agent)).astype(np.uint8)
    for threads in (1, 3, 16):
        out = np.full((E, n, 15, 15, 3), -1.0)
        assert L.ce_obs_u8_to_f64(pitched.ctypes.data, out.ctypes.data, E, n, n * agent, agent, row, threads) == 0
        dense = pitched.reshape(E, n, 15, row)[..., :45].reshape(E, n, 15, 15, 3)
        assert np.array_equal(out, dense / 255)
    assert L.ce_obs_u8_to_f64(None, out.ctypes.data, E, n, n * agent, agent, row, 1) != 0
    assert L.ce_obs_u8_to_f64(pitched.ctypes.data, out.ctypes.data, E, n, n * agent, agent, row, 0) != 0
    # every byte value: the table holds exactly numpy's quotient
    allv = np.zeros((1, 720), np.uint8)
    allv[0, :45] = np.arange(45)
    full = np.zeros((6, 720), np.uint8)
    for k in range(6):
        full[k, :45] = np.arange(45 * k, 45 * k + 45).clip(0, 255)
    out = np.zeros((6, 1, 15, 15, 3))
    assert L.ce_obs_u8_to_f64(full.ctypes.data, out.ctypes.data, 6, 1, 720, 720, 48, 2) == 0
    assert np.array_equal(out[:, 0, 0].reshape(6, 45), full[:, :45] / 255)


def test_host_pool_serves_changing_thread_counts_and_concurrent_callers():
    """the conversions share one process-wide worker pool: calls with growing / shrinking thread counts, sizes below one
    piece, and two callers at once all return complete, exact blocks"""
    import threading
    from contracts_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(5)
    n, row, agent = 3, 48, 720
    blocks = {E: rs.randint(0, 256, size=(E, n * agent)).astype(np.uint8) for E in (1, 7, 300, 1500)}

    def convert(E, threads):
        out = np.full((E, n, 15, 15, 3), -1.0)
        assert L.ce_obs_u8_to_f64(blocks[E].ctypes.data, out.ctypes.data, E, n, n * agent, agent, row, threads) == 0
        return out

    want = {E: b.reshape(E, n, 15, row)[..., :45].reshape(E, n, 15, 15, 3) / 255 for E, b in blocks.items()}
    for threads in (2, 9, 4, 33, 1, 16):
        for E in blocks:
            assert np.array_equal(convert(E, threads), want[E]), (E, threads)
    bad = []

    def hammer(E, threads):
        for _ in range(20):
            if not np.array_equal(convert(E, threads), want[E]):
                bad.append((E, threads))

    ts = [threading.Thread(target=hammer, args=a) for a in ((1500, 6), (300, 11), (1500, 3))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_i16_to_f64():
    from contracts_amd import _lib
    L = _lib.load()
    src = np.random.RandomState(1).randint(-300, 300, size=200001).astype(np.int16)
    for threads in (1, 4):
        out = np.zeros(src.size)
        assert L.ce_i16_to_f64(src.ctypes.data, out.ctypes.data, src.size, threads) == 0
        assert np.array_equal(out, src.astype(np.float64))


def test_pydict_parse_actions():
    from contracts_amd import _ce_pydict as pd
    E, n = 33, 4
    keys = tuple("a%d" % i for i in range(n))
    rs = np.random.RandomState(2)
    a = rs.randint(0, 9, size=(E, n))
    adict = {e: {k: (int(a[e, i]) if (e + i) % 2 else np.int64(a[e, i])) for i, k in enumerate(keys)} for e in range(E)}
    out = np.zeros((E, n), np.uint8)
    pd.parse_actions(adict, list(range(E)), keys, out)
    assert np.array_equal(out, a)
    with pytest.raises(KeyError):
        pd.parse_actions({e: adict[e] for e in range(E - 1)}, list(range(E)), keys, out)
    broken = dict(adict)
    broken[5] = {k: 1 for k in keys[:-1]}
    with pytest.raises(KeyError):
        pd.parse_actions(broken, list(range(E)), keys, out)
    broken[5] = dict.fromkeys(keys, 300)
    with pytest.raises(ValueError):
        pd.parse_actions(broken, list(range(E)), keys, out)
    broken[5] = dict.fromkeys(keys, "up")
    with pytest.raises(TypeError):
        pd.parse_actions(broken, list(range(E)), keys, out)
    with pytest.raises(ValueError):
        pd.parse_actions(adict, list(range(E)), keys, np.zeros(3, np.uint8))  # short output buffer


def test_pydict_refresh_loops_touch_only_what_changed():
    from contracts_amd import _ce_pydict as pd
    E, n = 50, 3
    keys = tuple("a%d" % i for i in range(n))
    rs = np.random.RandomState(3)
    # floats
    dicts = [dict.fromkeys(keys, 0.0) for _ in range(E)]
    shadow = np.zeros((E, n))
    sentinel = dicts[7]["a1"]
    new = np.zeros((E, n))
    new[3, 1], new[9, 2], new[9, 0] = 1.25, -0.0, float("nan")
    assert pd.refresh_floats(dicts, keys, new, shadow) == 3  # -0.0 differs from 0.0 bitwise: it is written through
    assert dicts[3] == {"a0": 0.0, "a1": 1.25, "a2": 0.0} and np.signbit(dicts[9]["a2"]) and np.isnan(dicts[9]["a0"])
    assert dicts[7]["a1"] is sentinel  # an unchanged entry keeps its object
    assert pd.refresh_floats(dicts, keys, new, shadow) == 0 and np.array_equal(shadow.view(np.uint64), new.view(np.uint64))
    # ints
    idicts = [dict.fromkeys(keys, 0) for _ in range(E)]
    ish, inew = np.zeros((E, n), np.int32), rs.randint(-60, 3, size=(E, n)).astype(np.int32)
    cnt = pd.refresh_ints(idicts, keys, inew, ish)
    assert cnt == int((inew != 0).sum()) and all(idicts[e][k] == int(inew[e, i]) for e in range(E) for i, k in enumerate(keys))
    assert all(type(v) is int for d in idicts for v in d.values())
    # infos: byte 0 -> 'eaten_apples', byte 1 -> the kind's second key
    infos = [{"cleaned_squares": 0, "eaten_apples": 0, "feature_obs": None} for _ in range(E * n)]
    sh, nw = np.zeros((E, n, 2), np.uint8), (rs.rand(E, n, 2) < 0.2).astype(np.uint8) * rs.randint(1, 6, size=(E, n, 2)).astype(np.uint8)
    pd.refresh_infos(infos, "eaten_apples", "cleaned_squares", nw, sh)
    flat = nw.reshape(-1, 2)
    assert all(infos[i]["eaten_apples"] == flat[i, 0] and infos[i]["cleaned_squares"] == flat[i, 1] for i in range(E * n))
    assert list(infos[0]) == ["cleaned_squares", "eaten_apples", "feature_obs"] and np.array_equal(sh, nw)
    nw2 = np.zeros_like(nw)
    pd.refresh_infos(infos, "eaten_apples", "cleaned_squares", nw2, sh)
    assert all(d["eaten_apples"] == 0 and d["cleaned_squares"] == 0 for d in infos)
    # dones
    dones = [{"__all__": False, "a0": False, "a1": False} for _ in range(E)]
    dsh, dnew = np.zeros(E, np.uint8), np.zeros(E, np.uint8)
    dnew[[4, 40]] = 1
    assert pd.refresh_dones(dones, ("__all__", "a0", "a1"), dnew, dsh) == 2
    assert dones[4] == {"__all__": True, "a0": True, "a1": True} and dones[5]["__all__"] is False
    assert pd.refresh_dones(dones, ("__all__", "a0", "a1"), np.zeros(E, np.uint8), dsh) == 2 and dones[40]["a1"] is False
    with pytest.raises(ValueError):
        pd.refresh_floats(dicts, keys, new[:10], shadow)
