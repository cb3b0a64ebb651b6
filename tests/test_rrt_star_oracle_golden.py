"""CPU: the RRT* oracle (oracle/rrt_oracle.c, orc_rrts_*) against tests/golden/rrt_star_golden.json, the
independent numpy restatement of oxmpl's RRTStar::solve (tests/golden/make_golden_rrt_star.py).
PARITY UNPINNED against oxmpl itself (oxmpl/tests/rrt_star_rvss_tests.rs asserts properties only; they are
re-asserted here on the oracle's output)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as orc
from helpers import unhex, bits, params_spheres, params_boxes, is_path_valid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def star_golden():
    with open(os.path.join(ROOT, "tests", "golden", "rrt_star_golden.json")) as f:
        return json.load(f)


def make_oracle(P, seed, pid, stop=True):
    o = orc.OracleRRTStar(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], P["search_radius"], P["fraction"],
                          P["max_nodes"], stop, seed, pid)
    if P["spheres"]:
        o.set_spheres(*params_spheres(P))
    if P["boxes"]:
        o.set_boxes(*params_boxes(P))
    o.setup(P["start"], P["goal_c"], P["goal_r"])
    return o


def check_run(o, P, r):
    assert o.num_nodes == r["n"] and o.iterations == r["iterations"] and o.accepted == r["accepted"]
    assert "%016x" % o.checksum == r["checksum"] and o.goal_node == r["goal_node"]
    states, parents = o.tree()
    head = np.array([[unhex(v) for v in row] for row in r["states"]])
    assert np.array_equal(bits(states[:len(head)]), bits(head))
    assert list(parents) == r["parents"]
    assert np.array_equal(bits(o.costs()), bits(np.array([unhex(c) for c in r["cost"]])))
    want = np.array([[unhex(v) for v in row] for row in r["path"]]).reshape(-1, P["dim"])
    got = o.path()
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("key", ["wall", "wall_ref", "config1", "config2"])
def test_rrt_star_oracle_matches_numpy_restatement(star_golden, key):
    P = star_golden[key]["params"]
    for r in star_golden[key]["runs"]:
        o = make_oracle(P, r["seed"], r["pid"], stop=key != "config2")
        o.solve(P["max_iterations"])
        check_run(o, P, r)


def test_rrt_star_reference_properties(star_golden):
    """oxmpl/tests/rrt_star_rvss_tests.rs:167-185: path starts at the start, ends in the goal, every edge valid"""
    P = star_golden["wall_ref"]["params"]
    r = star_golden["wall_ref"]["runs"][0]
    o = make_oracle(P, r["seed"], r["pid"])
    assert o.solve(10 ** 6) == orc.SOLVED
    path = o.path()
    assert orc.distance(path[0], P["start"]) < 1e-9
    assert orc.distance(path[-1], P["goal_c"]) <= P["goal_r"]
    lo, hi = params_boxes(P)

    def valid(p):
        return not any(all(lo[b][k] <= p[k] <= hi[b][k] for k in range(2)) for b in range(len(lo)))

    assert is_path_valid(path, [tuple(b) for b in P["bounds"]], P["fraction"], valid, orc.maximum_extent, orc.num_steps,
                         orc.interpolate, orc.distance)
    # costs: the root is free, every other node costs at least the straight line from the start
    c = o.costs()
    states, parents = o.tree()
    assert c[0] == 0.0 and parents[0] == -1
    d0 = np.sqrt(((states - states[0]) ** 2).sum(axis=1))
    assert np.all(c >= d0 - 1e-9)
