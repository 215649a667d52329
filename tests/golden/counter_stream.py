"""The engine's counter-RNG stream (CE_FLAG_RNG_COUNTER, include/contracts_engine.h) as something the REFERENCE can draw from.

Build-container tooling for `make_counter_golden.py`; pure Python + numpy, no import of the engine or of `oracle/`.

Two independent pieces:

* `LegacyDraws(words)` — numpy's legacy `RandomState` algorithms restated over an arbitrary source of 32-bit words: the
  masked-rejection `random_interval` behind `shuffle`, `random_sample`'s 53-bit doubles from two words, `randint`'s
  masked rejection (`w & 3` for `randint(4)`), `uniform = low + (high - low) * double`.  `tests/test_counter_rng.py` pins this
  class to the real thing: fed the raw MT19937 words of `np.random.MT19937`, every call returns what `np.random.RandomState`
  returns on the same arguments.  Draw sites it stands in for: map_env.py:546,685,821,831, cleanup_new.py:326,339,
  harvest_new.py:294, two_stage_train.py:163-164.
* `CounterWords(seed)` — the documented stream: key = (seed low word, seed high word); generation g = 1, 2, ... is the 128
  Philox4x32-10 blocks with counters (q, g, 0, 0), q = 0 .. 127 — 512 words; a stream word is MT19937's tempering of the
  block word; every operation on an env (construct, reset, step) opens a fresh generation and drops what it leaves
  unread; an operation that needs more than 512 words runs on into the next generation.

`patched(stream)` swaps the process-global `np.random` functions the reference's hot path calls (and the `rand` names its
modules bound at import) for the counter stream's, and makes every other global draw function raise, so that an unpatched
call site cannot silently fall back to MT19937.
"""
import contextlib

import numpy as np

GEN_WORDS = 512


def philox4x32_10(key, counters):
    """Philox4x32-10 (Salmon et al., SC'11) over an [N, 4] array of counters"""
    c = np.asarray(counters, np.uint64).copy()
    k0, k1 = int(key[0]), int(key[1])
    m0, m1, lo = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xffffffff)
    s32 = np.uint64(32)
    for _ in range(10):
        p0, p1 = m0 * c[:, 0], m1 * c[:, 2]
        c = np.stack([(p1 >> s32) ^ c[:, 1] ^ np.uint64(k0), p1 & lo, (p0 >> s32) ^ c[:, 3] ^ np.uint64(k1), p0 & lo], axis=1)
        k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
    return c.astype(np.uint32)


def temper(y):
    y = np.asarray(y, np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9d2c5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xefc60000)
    y ^= y >> np.uint32(18)
    return y


class CounterWords:
    def __init__(self, seed=0):
        self.seed(seed)

    def seed(self, seed):
        seed = int(seed)
        assert 0 <= seed < 1 << 64
        self.k0, self.k1 = seed & 0xffffffff, seed >> 32
        self.gen, self.pos, self.buf = 0, GEN_WORDS, None
        self.consumed = 0  # words handed out since the seed (diagnostic only)

    def _next_generation(self):
        self.gen += 1
        ctr = np.zeros((GEN_WORDS // 4, 4), np.uint64)
        ctr[:, 0] = np.arange(GEN_WORDS // 4)
        ctr[:, 1] = self.gen
        self.buf = [int(w) for w in temper(philox4x32_10((self.k0, self.k1), ctr).reshape(-1))]
        self.pos = 0

    def next32(self):
        if self.pos >= GEN_WORDS:
            self._next_generation()
        w = self.buf[self.pos]
        self.pos += 1
        self.consumed += 1
        return w

    @contextlib.contextmanager
    def op(self):
        """one operation on the env: a fresh generation, the unread rest dropped at the end"""
        self._next_generation()
        yield self
        self.pos = GEN_WORDS

    def state_row(self):
        """what the engine keeps per env in this mode: key0, key1, generation"""
        return np.array([self.k0, self.k1, self.gen], np.int64)


class WordList:
    """a fixed list of words as a source (the pinning test feeds raw MT19937 output through this)"""

    def __init__(self, words):
        self.words, self.pos = [int(w) for w in words], 0

    def next32(self):
        w = self.words[self.pos]
        self.pos += 1
        return w


class LegacyDraws:
    def __init__(self, words):
        self.words = words

    # numpy/random/src/legacy + mtrand.pyx, restated ------------------------------------------------------------------
    def _interval(self, top):
        """random_interval: uniform on [0, top] by masked rejection, one 32-bit word per attempt (top < 2**32)"""
        if top == 0:
            return 0
        mask = top
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        assert top <= 0xffffffff
        while True:
            v = self.words.next32() & mask
            if v <= top:
                return v

    def _double(self):
        a, b = self.words.next32() >> 5, self.words.next32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def shuffle(self, x):
        n = len(x)
        if isinstance(x, np.ndarray) and x.ndim > 1:
            for i in reversed(range(1, n)):
                j = self._interval(i)
                if i != j:
                    x[[i, j]] = x[[j, i]]
            return
        for i in reversed(range(1, n)):
            j = self._interval(i)
            x[i], x[j] = x[j], x[i]

    def random_sample(self, size=None):
        if size is None:
            return self._double()
        out = np.empty(size, np.float64)
        flat = out.reshape(-1)
        for i in range(flat.size):
            flat[i] = self._double()
        return out

    def rand(self, *shape):
        return self.random_sample(shape if shape else None)

    def randint(self, low, high=None, size=None):
        if high is None:
            low, high = 0, low
        span = int(high) - int(low) - 1
        assert 0 <= span < 0xffffffff

        def one():
            return int(low) + self._interval(span)
        if size is None:
            return one()
        out = np.empty(size, np.int64)
        flat = out.reshape(-1)
        for i in range(flat.size):
            flat[i] = one()
        return out

    def uniform(self, low=0.0, high=1.0, size=None):
        lo = np.asarray(low, np.float64)
        scale = np.asarray(high, np.float64) - lo
        if size is None and lo.ndim == 0 and scale.ndim == 0:
            return float(lo) + float(scale) * self._double()
        shape = np.broadcast(lo, scale).shape if size is None else size
        out = np.empty(shape, np.float64)
        lo_b, sc_b = np.broadcast_to(lo, out.shape).reshape(-1), np.broadcast_to(scale, out.shape).reshape(-1)
        flat = out.reshape(-1)
        for i in range(flat.size):
            flat[i] = lo_b[i] + sc_b[i] * self._double()
        return out


_FORBIDDEN = ("random", "ranf", "sample", "choice", "permutation", "multinomial", "normal", "standard_normal", "random_integers",
              "bytes", "binomial", "beta", "exponential", "poisson", "get_state", "set_state")


@contextlib.contextmanager
def patched(stream, modules=()):
    """np.random.{seed, shuffle, rand, random_sample, randint, uniform} draw from `stream` (a CounterWords); `modules` are
    reference modules that did `from numpy.random import rand`"""
    draws = LegacyDraws(stream)
    saved = {name: getattr(np.random, name) for name in ("seed", "shuffle", "rand", "random_sample", "randint", "uniform") + _FORBIDDEN}
    saved_mod = [(m, m.rand) for m in modules if hasattr(m, "rand")]

    def refuse(name):
        def f(*a, **k):
            raise AssertionError("np.random.%s is not routed to the counter stream" % name)
        return f
    try:
        np.random.seed = stream.seed
        for name in ("shuffle", "rand", "random_sample", "randint", "uniform"):
            setattr(np.random, name, getattr(draws, name))
        for name in _FORBIDDEN:
            setattr(np.random, name, refuse(name))
        for m, _ in saved_mod:
            m.rand = draws.rand
        yield draws
    finally:
        for name, f in saved.items():
            setattr(np.random, name, f)
        for m, f in saved_mod:
            m.rand = f
