"""CPU: the disc goal sampler (OXHIP_GOAL_SAMPLE_UNIFORM_DISC; reference: oxmpl/tests/rrt_rvss_tests.rs:55-66).
(i)   the C oracle equals the independent pure-Python restatement (tests/golden/make_golden_disc.py -> rrt_disc_golden.json) bit
      for bit: sin / cos KATs, trees, paths, checksums, RNG positions on the reference's wall scene and the README scene;
(ii)  ox_sincos against this host's libm over [0, 2 PI): never more than one ulp apart (how often they differ is printed);
(iii) the same planners with libm's sin / cos -- what a rustc-built oxmpl would call -- stay within north_star's 1e-6 relative
      bound of the ox_sincos runs on these fixtures (same node counts and parents; coordinates to 1e-6 relative).
PARITY UNPINNED against oxmpl itself, as for the rest of the path."""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle_py as orc
from helpers import hexf, params_boxes, params_spheres

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def disc_golden():
    with open(os.path.join(ROOT, "tests", "golden", "rrt_disc_golden.json")) as f:
        return json.load(f)


def _planner(P, goal_bias, seed, pid, stop=True):
    o = orc.OracleRRT(2, P["bounds"], P["max_distance"], goal_bias, P["fraction"], P["max_nodes"], stop, seed, pid)
    o.set_goal_sampler(1)
    if P["spheres"]:
        o.set_spheres(*params_spheres(P))
    if P["boxes"]:
        o.set_boxes(*params_boxes(P))
    o.setup(P["start"], P["goal_c"], P["goal_r"])
    return o


def test_sincos_kats(disc_golden):
    from helpers import unhex
    for row in disc_golden["sincos"]:
        s, c = orc.sincos(unhex(row["x"]))
        assert (hexf(s), hexf(c)) == (row["sin"], row["cos"])


@pytest.mark.parametrize("scene", ["wall", "config1"])
def test_oracle_equals_the_python_restatement(disc_golden, scene):
    P = disc_golden[scene]["params"]
    for run in disc_golden[scene]["runs"]:
        o = _planner(P, run["goal_bias"], run["seed"], run["pid"])
        o.solve(run["max_iterations"])
        assert (o.num_nodes, o.iterations, o.accepted, o.goal_node) == (run["n"], run["iterations"], run["accepted"], run["goal_node"])
        assert "%016x" % o.checksum == run["checksum"]
        states, parents = o.tree()
        m = len(run["first_parents"])
        assert [[hexf(v) for v in row] for row in states[:m]] == run["first_states"]
        assert list(parents[:m]) == run["first_parents"]
        assert [[hexf(v) for v in row] for row in o.path()] == run["path"]
        if run["goal_bias"] < 1.0:
            # the reference's own assertions (rrt_rvss_tests.rs:168-180): the path starts at the start and ends in the goal disc
            path = o.path()
            assert orc.distance(path[0], P["start"]) < 1e-9 and orc.distance(path[-1], P["goal_c"]) <= P["goal_r"]


def test_ox_sincos_is_within_one_ulp_of_libm():
    rng = np.random.default_rng(17)
    xs = np.concatenate([rng.random(200000) * 2.0 * math.pi, np.arange(9) * (math.pi / 4)])
    diff_s = diff_c = 0
    for x in xs:
        s, c = orc.sincos(float(x))
        ls, lc = math.sin(x), math.cos(x)
        assert abs(s - ls) <= abs(np.spacing(ls)) and abs(c - lc) <= abs(np.spacing(lc)), x
        diff_s += s != ls
        diff_c += c != lc
    print("ox_sincos != libm on %d (sin) / %d (cos) of %d angles, never by more than one ulp" % (diff_s, diff_c, len(xs)))
    assert diff_s < 0.1 * len(xs) and diff_c < 0.1 * len(xs)


def test_libm_runs_stay_within_1e_6_relative(disc_golden):
    P = disc_golden["wall"]["params"]
    try:
        for goal_bias in (0.05, 0.5):
            for seed in range(2):
                a = _planner(P, goal_bias, seed, 9)
                a.solve(5000)
                orc.set_sincos_libm(True)
                b = _planner(P, goal_bias, seed, 9)
                b.solve(5000)
                orc.set_sincos_libm(False)
                assert (a.num_nodes, a.iterations, a.goal_node) == (b.num_nodes, b.iterations, b.goal_node)
                sa, pa = a.tree()
                sb, pb = b.tree()
                assert np.array_equal(pa, pb)
                assert np.allclose(sa, sb, rtol=1e-6, atol=1e-9)
                assert np.allclose(a.path(), b.path(), rtol=1e-6, atol=1e-9)
    finally:
        orc.set_sincos_libm(False)
