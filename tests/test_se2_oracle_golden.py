"""CPU: oracle/se2_oracle.c (RRTConnect over the build-defined SE(2) space among line segments, BASELINE.json
configs[3]) against tests/golden/se2_golden.json, the independent numpy restatement
(tests/golden/make_golden_se2.py).  PARITY UNPINNED against oxmpl: the reference has no SE(2) space."""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle_py as orc
from helpers import unhex, hexf, bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def se2_golden():
    with open(os.path.join(ROOT, "tests", "golden", "se2_golden.json")) as f:
        return json.load(f)


def test_so2_se2_and_segment_kats(se2_golden):
    L = orc.lib()
    for k in se2_golden["kat"]["random"]:
        a, b = [unhex(v) for v in k["a"]], [unhex(v) for v in k["b"]]
        t, seg = unhex(k["t"]), [unhex(v) for v in k["seg"]]
        assert hexf(L.orc_so2_normalise(a[2])) == k["so2_normalise"]
        assert hexf(L.orc_so2_distance(a[2], b[2])) == k["so2_distance"]
        assert hexf(L.orc_so2_interpolate(a[2], b[2], t)) == k["so2_interpolate"]
        assert hexf(orc.se2_distance(a, b)) == k["se2_distance"]
        assert [hexf(v) for v in orc.se2_interpolate(a, b, t)] == k["se2_interpolate"]
        assert hexf(orc.point_segment_distance(a[0], a[1], seg)) == k["point_segment"]
        assert -math.pi <= L.orc_so2_normalise(a[2]) < math.pi          # so2_state.rs:8
        assert 0.0 <= L.orc_so2_distance(a[2], b[2]) <= math.pi + 1e-15
    for e in se2_golden["kat"]["so2_edges"]:
        a, b = unhex(e["a"]), unhex(e["b"])
        assert hexf(L.orc_so2_normalise(a)) == e["normalise"]
        assert hexf(L.orc_so2_distance(a, b)) == e["distance"]
        assert hexf(L.orc_so2_interpolate(a, b, 0.5)) == e["interp_half"]
    assert hexf(orc.point_segment_distance(1.0, 2.0, [3.0, 4.0, 3.0, 4.0])) == se2_golden["kat"]["degenerate_segment"]
    b, pb = orc._d([0.0, 10.0, 0.0, 10.0])
    assert hexf(L.orc_se2_extent(pb)) == se2_golden["kat"]["extent"]
    # the doc examples of so2_state.rs:23-32,49-53
    assert abs(L.orc_so2_normalise(3.0 * math.pi / 2.0) + math.pi / 2.0) < 1e-9
    assert abs(L.orc_so2_normalise(5.0 * math.pi) + math.pi) < 1e-9
    # the one exact known answer the reference's own unit tests hold for this arithmetic
    # (so2_state.rs:80-87, test_so2_state_normalise: assert_eq!(state2.value, -PI / 2.0))
    assert L.orc_so2_normalise(3.0 * math.pi / 2.0) == -math.pi / 2.0


def make_oracle(P, seed, pid):
    o = orc.OracleSE2Connect(P["bounds_xy"], P["theta_bounds"], P["max_distance"], P["goal_bias"], P["fraction"],
                             P["max_nodes"], seed, pid)
    o.set_segments([[unhex(v) for v in s] for s in P["segments"]], P["clearance"])
    o.setup(P["start"], P["goal"], P["goal_r"])
    return o


@pytest.mark.parametrize("key", ["soup256", "gap"])
def test_se2_connect_oracle_matches_numpy_restatement(se2_golden, key):
    P = se2_golden[key]["params"]
    for r in se2_golden[key]["runs"]:
        o = make_oracle(P, r["seed"], r["pid"])
        assert o.solve(P["max_iterations"]) == orc.SOLVED
        assert [o.num_nodes(0), o.num_nodes(1)] == r["n"] and o.iterations == r["iterations"]
        assert "%016x" % o.checksum == r["checksum"] and [o.end_node(0), o.end_node(1)] == r["end"]
        for w in (0, 1):
            s, p = o.tree(w)
            head = np.array([[unhex(v) for v in row] for row in r["states"][w]])
            assert np.array_equal(bits(s[:len(head)]), bits(head)) and list(p[:len(head)]) == r["parents"][w]
        want = np.array([[unhex(v) for v in row] for row in r["path"]]).reshape(-1, 3)
        got = o.path()
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want))
        # properties in the spirit of oxmpl/tests/rrt_connect_rvss_tests.rs:170-185 with the SE(2) pieces
        assert orc.se2_distance(got[0], P["start"]) < 1e-9 and orc.se2_distance(got[-1], P["goal"]) <= P["goal_r"]
        for a, b in zip(got, got[1:]):
            assert o.is_valid(a) and o.is_valid(b) and o.check_motion(a, b)
        assert np.all(got[:, 2] >= -math.pi) and np.all(got[:, 2] < math.pi)


def test_se2_oracle_rejects_bad_spaces():
    with pytest.raises(ValueError):
        orc.OracleSE2Connect([(0.0, 1.0), (0.0, 1.0)], (1.0, 1.0), 0.5, 0.05)          # so2_state_space.rs:59-64
    with pytest.raises(ValueError):
        orc.OracleSE2Connect([(0.0, float("inf")), (0.0, 1.0)], (-1.0, 1.0), 0.5, 0.05)
    o = orc.OracleSE2Connect([(0.0, 1.0), (0.0, 1.0)], (-10.0, 10.0), 0.5, 0.05)
    lo, hi = np.zeros(1), np.zeros(1)
    orc.lib().orc_se2c_theta_bounds(o.h, lo.ctypes.data_as(orc._dp), hi.ctypes.data_as(orc._dp))
    assert (lo[0], hi[0]) == (-math.pi, math.pi)                                          # clamped, so2_state_space.rs:67
