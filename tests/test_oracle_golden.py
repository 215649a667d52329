"""CPU: the oracle (oracle/oracle.c) against every golden fixture produced by the reference.
This is the pin that makes the oracle trustworthy as the GPU parity checker."""
import pytest

import golden_check as gc
from oracle.pyoracle import Oracle


@pytest.mark.parametrize("name", [f for f in gc.fixtures("g") if "selfdrive" not in f])
def test_oracle_grid_golden(name):
    g = gc.load(name)
    kind, n, kw = gc.grid_kwargs(g)
    gc.replay_grid(g, Oracle(kind, 2, n, **kw), env=1)


@pytest.mark.parametrize("name", gc.fixtures("p"))
def test_oracle_counter_rng_golden(name):
    """p_* fixtures: the REFERENCE run with np.random routed to the counter stream (tests/golden/make_counter_golden.py) — the
    oracle in rng="counter" mode reproduces every field of every step, and the (key0, key1, generation) row after every
    construct / reset / step"""
    g = gc.load(name)
    assert str(g["rng_mode"]) == "counter"
    kind, n, kw = gc.grid_kwargs(g)
    gc.replay_grid(g, Oracle(kind, 2, n, rng="counter", **kw), env=1)


@pytest.mark.parametrize("name", gc.fixtures("g5_selfdrive"))
def test_oracle_selfdrive_golden(name):
    g = gc.load(name)
    gc.replay_selfdrive(g, Oracle("selfdrive", 2, int(g["n"]), contract="selfdrive_distprop",
                               collision_on=bool(int(g["collision_on"]))), env=1)


@pytest.mark.parametrize("name", gc.fixtures("render_"))
def test_oracle_render_golden(name):
    """beam cells (MapEnv.beam_pos) and the composed full_map_to_colors image, every step"""
    g = gc.load(name)
    gc.replay_render(g, Oracle(str(g["kind"]), 2, int(g["n"]), horizon=int(g["horizon"]), firing=True, beam_trace=True), env=1)
