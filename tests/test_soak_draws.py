"""The random draws of the soak (tests/soak_lib.py) are what the recorded runs under profiles/r05_soak_*.txt are reproduced from: the default stream
must not move when a flavour is added, and every flavour must draw what it says."""
import numpy as np

from tests import soak_lib


def _draw(seed, n, **kw):
    rng = np.random.default_rng(seed)
    return [soak_lib.draw_scene(rng, seed, s, 100, **kw) for s in range(n)]


def test_default_stream_is_pinned():
    d = _draw(36, 3)
    assert [(x["dyn"], x["T"], x["N"], x["kw"]["iter_num"], x["kw"]["max_obs_num"], round(x["speed"], 6)) for x in d] == \
        [("diff", 10, 30, 4, 34, 2.542654), ("diff", 15, 31, 3, 35, 3.306172), ("diff", 20, 18, 4, 17, 3.587156)]
    assert all(x["kw"]["max_edge_num"] == 4 and np.shape(x["car"].G)[0] == 4 for x in d)


def test_flavours_draw_what_they_say():
    large = _draw(50, 4, large=True)
    assert all(x["T"] in (20, 25, 30) and 100 <= x["N"] < 420 for x in large)
    ex = _draw(60, 12, exotic=True, robots=True)
    assert all(x["T"] in (5, 12, 20, 40) and x["kw"]["max_edge_num"] in (5, 6, 8) for x in ex)
    assert {np.shape(x["car"].G)[0] for x in ex} - {4} and all(x["car"].cone_type == "Rpositive" for x in ex)      # other bodies; circle robots only on request
    assert any(x["car"].cone_type == "norm2" for x in _draw(62, 12, exotic=True, circle_robot=True))
    assert max(o.vertex.shape[1] for x in ex for o in x["scene"] if o.vertex is not None) > 4
    circ = _draw(100, 3, circles=True)
    for x in circ:
        kinds = [o.cone_type for o in x["scene"][:x["N"]]]
        assert kinds.count("norm2") >= len(kinds) // 2
    tight, plain = _draw(70, 2, tight=True), _draw(70, 2)
    assert [(a["dyn"], a["T"], a["N"]) for a in tight] == [(a["dyn"], a["T"], a["N"]) for a in plain]               # same draws, half the clearance
