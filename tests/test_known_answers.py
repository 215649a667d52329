"""Named known-answer scenarios — one reference step from hand-placed agents,
recorded from the reference (tests/golden/make_ka.py).  The fixture doubles as documentation of the
reference's conflict-resolution semantics; a few of its facts are asserted literally below."""
import numpy as np
import pytest

import golden_check as gc
from test_fuzz_states import run_fuzz


def _ka():
    g = gc.load("ka_cleanup")
    return g, [str(x) for x in g["names"]]


def _idx(names, prefix):
    return [i for i, x in enumerate(names) if x.startswith(prefix)]


def test_fixture_documents_reference_semantics():
    g, names = _ka()
    pos_in, pos_out = g["in_agents"][:, :, :2], g["out_agents"][:, :, :2]
    (i,) = _idx(names, "KA2 ")
    assert np.array_equal(pos_in[i], pos_out[i])                       # walls block
    (i,) = _idx(names, "KA5 ")
    assert np.array_equal(pos_in[i], pos_out[i])                       # head-on swap: both stay
    (i,) = _idx(names, "KA6 ")
    assert pos_out[i][0].tolist() == [5, 8] and pos_out[i][1].tolist() == [5, 9]   # chain: both move
    (i,) = _idx(names, "KA7 ")
    assert pos_out[i].tolist() == [[5, 8], [6, 8], [6, 7], [5, 7]]     # 4-cycle rotates everyone
    for i in _idx(names, "KA4"):
        assert np.array_equal(pos_in[i], pos_out[i])                   # occupied contested cell: nobody moves
    winners = {tuple(pos_out[i][0]) for i in _idx(names, "KA3 ")}
    assert winners == {(5, 8), (5, 7)}                                 # the shuffle decides who gets the cell
    shared = [i for i in _idx(names, "KA8 ") if len({tuple(p) for p in pos_out[i]}) < 4]
    assert shared, "KA8: two agents end on one cell for some seeds"
    for i in _idx(names, "KA8 "):
        assert pos_out[i][0].tolist() == [5, 8]                        # the blocked occupant stays
    (i,) = _idx(names, "KA10a")
    assert g["base_rew"][i].tolist() == [-1, 0, -50, 0]                # beam absorbed by the first agent it meets
    (i,) = _idx(names, "KA10c")
    assert g["base_rew"][i].tolist() == [-1, 0, -50, 0]                # shared cell: the later agent is hit
    (i,) = _idx(names, "KA9a")
    assert g["second"][i].tolist() == [1, 0, 0, 0]                     # one H cleaned, then the beam stops


def test_harvest_fixture_documents_reference_semantics():
    g = gc.load("ka_harvest")
    names = [str(x) for x in g["names"]]
    close = {int(n.split(" with ")[1].split()[0]): int(g["second"][i][0]) for i, n in enumerate(names) if n.startswith("KA14")}
    assert close == {0: 1, 1: 1, 2: 1, 3: 0, 4: 0, 5: 0}                # eaten_close: fewer than 4 apples (itself included) within d^2 <= 5
    (i,) = _idx(names, "KA10h")
    assert g["base_rew"][i].tolist() == [-1, -50, 0, 0]                # FIRE passes over apples, the first agent absorbs it
    under = [i for i in _idx(names, "KA11c")]
    t = tuple(g["in_agents"][under[0]][0][:2])
    assert all(g["out_grid"][i][t] != 2 for i in under)                # never respawns under an agent


@pytest.mark.parametrize("kind", ["cleanup", "harvest"])
def test_oracle_known_answers(kind):
    from oracle.pyoracle import Oracle
    g = gc.load("ka_" + kind)
    names = [str(x) for x in g["names"]]
    orc = Oracle(kind, len(names), 4, firing=True)

    def put(field, arr):
        getattr(orc, field)[...] = arr
        orc.import_state()

    run_fuzz(g, orc, lambda f: getattr(orc, f), put)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cleanup", "harvest"])
def test_engine_known_answers(kind):
    from contracts_amd.engine import BatchedEnv
    g = gc.load("ka_" + kind)
    names = [str(x) for x in g["names"]]
    env = BatchedEnv(kind, len(names), 4, firing=True)
    run_fuzz(g, env, env.download, env.upload)
    env.close()
