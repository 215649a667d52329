"""CPU only: the oracle built with AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile: liboracle_asan.so)
replays reference fixtures of every family — out-of-bounds indexing, uninitialised scratch or signed overflow in the
checker would make every parity claim that rests on it worthless.  (GPU sanitizers are not available on this pool;
SURVEY.md §5.)  Runs in a child process: the ASan runtime must be loaded before the interpreter."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import golden_check as gc
from oracle.pyoracle import Oracle
for name in ("g7_cleanup_n8_s1", "g7_harvest_n8_s0", "g8_cleanup_n9", "g9b_harvest_n4_inequity_contract_done"):
    g = gc.load(name)
    kind, n, kw = gc.grid_kwargs(g)
    o = Oracle(kind, 3, n, **kw)
    gc.replay_grid(g, o, env=2)
    o.close()
g = gc.load("g5_selfdrive_n4")
o = Oracle("selfdrive", 2, 4, contract="selfdrive_distprop")
gc.replay_selfdrive(g, o, env=1)
o.close()
import numpy as np
for kind, n in (("harvest_features", 2), ("cleanup_features", 5)):
    o = Oracle(kind, 4, n, contract="harvest_local" if kind.startswith("harvest") else "cleanup", horizon=40, auto_reset=True)
    o.seed(seed0=11); o.reset()
    rs = np.random.RandomState(1)
    for t in range(90):
        o.step(rs.randint(7, size=(4, n)).astype(np.uint8))
    o.close()
print("sanitized replay ok")
"""


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/liboracle_asan.so"])
    env = dict(os.environ, LD_PRELOAD=libasan, CONTRACTS_ORACLE_LIB=lib, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "here": HERE}], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0 and "sanitized replay ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
