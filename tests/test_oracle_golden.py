"""CPU: the oracle (oracle/oracle.c) against every golden fixture produced by the reference.
This is the pin that makes the oracle trustworthy as the GPU parity checker."""
import pytest

import golden_check as gc
from oracle.pyoracle import Oracle


@pytest.mark.parametrize("name", [f for f in gc.fixtures("g") if "selfdrive" not in f] + gc.fixtures("m"))
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


def test_oracle_custom_layout_rules():
    """ce_config.ascii_map (include/contracts_engine.h): within the frame and the tables of the kind's shipped layout, walled in,
    at least one cell of every list the kind uses, enough spawn points for the agents"""
    ok = ["@@@@@@", "@HB P@", "@RB P@", "@@@@@@"]
    Oracle("cleanup", 1, 2, ascii_map=ok).close()
    Oracle("harvest", 1, 1, ascii_map=["@@@@", "@AP@", "@@@@"]).close()
    for kind, n, rows in (("cleanup", 3, ok),                                        # more agents than spawn points
                          ("cleanup", 1, ["@@@@@@", "@HB P ", "@RB P@", "@@@@@@"]),   # open edge
                          ("cleanup", 1, ["@@@@@@", "@ B P@", "@@@@@@"]),             # no waste cell
                          ("cleanup", 1, ["@@@@@@", "@HA P@", "@@@@@@"]),             # harvest's apple letter
                          ("harvest", 1, ["@@@@", "@ P@", "@@@@"]),                   # no apple cell
                          ("harvest", 1, ["@" * 40] * 3),                              # wider than the frame
                          ("cleanup", 1, ["@" * 18] + ["@" + "B" * 16 + "@"] * 8 + ["@HP" + "@" * 15] + ["@" * 18]),  # 128 apple cells > 103
                          ("harvest_features", 2, ["@@@@", "@AP@", "@@@@"])):         # grid kinds only
        with pytest.raises(RuntimeError):
            Oracle(kind, 1, n, ascii_map=rows)
    with pytest.raises(ValueError):
        Oracle("cleanup", 1, 1, ascii_map=["@@@", "@@"])
