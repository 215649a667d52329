"""CPU: the arithmetic behind the engine's waste-list shuffle (ce_grid_kernels.hip: shuffle_draws), restated in a
few lines of numpy and held to numpy's own legacy `RandomState.shuffle` — the call the reference makes (cleanup_new.py:339).

np.random.shuffle(x) of a list walks i = len-1 .. 1 and draws j = random_interval(i): masked rejection sampling, one 32-bit
stream word per attempt.  The kernel solves 64 cached words at a time as the unique fixed point of
    Y = { k : (w_k & mask(i_k)) > i_k },   i_k = i0 - (k - off) + |Y below k|,   mask(i) = 2^bitlen(i) - 1
iterated from any start until a pass reproduces its input.  This test runs exactly that iteration (same batch size, same start,
two passes per convergence test, the same treatment of the words behind the end of the list) on the words numpy itself would
consume, applies the draws as the swaps they stand for, and compares the shuffled list AND the stream position with numpy's."""
import numpy as np
import pytest


def _ffbh_i32(x):
    """v_ffbh_i32: leading bits equal to the sign bit; -1 (all ones) for 0 and -1"""
    x &= 0xffffffff
    if x in (0, 0xffffffff):
        return 0xffffffff
    s = x >> 31
    for k in range(1, 32):
        if ((x >> (31 - k)) & 1) != s:
            return k
    return 0xffffffff


def _mask(idx):
    return 0xffffffff >> (_ffbh_i32(int(idx)) & 31)


def fixed_point_draws(words, length, batch=64):
    """-> (J, words consumed, passes): J[i] = the draw of index i, as the kernel computes them"""
    pos, i0, J, passes = 0, length - 1, {}, 0
    while i0 >= 1:
        w = words[pos:pos + batch]
        c7 = [int(x) & 127 for x in w]
        d = np.arange(len(w))
        istart = i0 - d

        def rejected(idx):
            return np.array([(c & _mask(i)) > int(i) for c, i in zip(c7, idx)])

        idx = istart.copy()
        Y = rejected(idx)
        while True:
            idx = istart + (np.cumsum(Y) - Y)
            Y1 = rejected(idx)
            idx = istart + (np.cumsum(Y1) - Y1)
            Y = rejected(idx)
            passes += 2
            if (Y == Y1).all():
                break
        real = (~Y) & (idx >= 1)
        for k in np.nonzero(real)[0]:
            J[int(idx[k])] = c7[k] & _mask(idx[k])
        total = int(real.sum())
        if total >= i0:
            pos += int(np.nonzero(real & (idx == 1))[0][0]) + 1
            i0 = 0
        else:
            pos += len(w)
            i0 -= total
    return J, pos, passes


@pytest.mark.parametrize("length", [119, 8, 5, 2, 65, 128, 97])
def test_fixed_point_draws_are_numpys_shuffle(length):
    passes = []
    for seed in range(40):
        rs = np.random.RandomState(1000 * length + seed)
        rs.random_sample(seed % 7)  # an arbitrary stream position (two words per double)
        twin = np.random.RandomState()
        twin.set_state(rs.get_state())
        words = np.frombuffer(twin.bytes(4 * 1024), dtype="<u4").astype(np.int64)  # the next 1024 stream words
        x = list(range(length))
        rs.shuffle(x)
        J, used, p = fixed_point_draws(words, length)
        passes.append(p)
        y = list(range(length))
        for i in range(length - 1, 0, -1):
            y[i], y[J[i]] = y[J[i]], y[i]
        assert y == x, (length, seed)
        # numpy is now `used` words further down the same stream
        after = np.frombuffer(rs.bytes(16), dtype="<u4")
        assert np.array_equal(after, words[used:used + 4].astype(np.uint32)), (length, seed, used)
    if length == 119:  # the shipped list: three batches of 64 words, 14-16 passes on average
        assert 12 <= np.mean(passes) <= 19 and max(passes) <= 40
