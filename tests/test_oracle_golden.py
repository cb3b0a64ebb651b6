"""CPU: the C oracle against the golden fixtures produced by the independent numpy
restatement (tests/golden/make_golden.py) and against published ChaCha vectors.
PARITY UNPINNED against oxmpl itself (no reference-held vectors exist)."""
import math
import struct

import numpy as np
import pytest

from oracle import oracle_py as orc
from helpers import unhex, hexf, bits, params_spheres, params_boxes, is_path_valid


def test_chacha_published_vectors():
    # RFC 7539 section 2.3.2 (ChaCha20 block function)
    key = struct.unpack("<8I", bytes(range(32)))
    out = orc.chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)
    assert ["%08x" % w for w in out[:8]] == ["e4e7f110", "15593bd1", "1fdd0f50", "c47120a3",
                                              "c7f4d1c7", "0368c033", "9aaa2204", "4e6cd4c3"]
    # all-zero key/nonce keystream block 0, 20 / 12 / 8 rounds
    for rounds, head in ((20, "76b8e0ada0f13d90405d6ae55386bd28"), (12, "9bf49a6a0755f953811fce125f2683d5"),
                         (8, "3e00ef2f895f40d67f5bb8e81f09a5a1")):
        out = orc.chacha_block([0] * 8, 0, 0, rounds)
        assert struct.pack("<16I", *out).hex().startswith(head)


def test_rng_stream_and_transforms(golden):
    g = golden["rng"]
    r = orc.Rng(g["seed"], g["stream"])
    assert ["%016x" % r.next_u64() for _ in range(40)] == g["u64"]
    for p, want in g["p_int"].items():
        assert "%016x" % orc.lib().orc_bernoulli_p_int(float(p)) == want
    r = orc.Rng(1234, 0)
    for case in g["range"]:
        got = [hexf(r.random_range(case["lo"], case["hi"])) for _ in case["v"]]
        assert got == case["v"]
        for h in got:
            assert case["lo"] <= unhex(h) < case["hi"]
    b = g["bools"]
    r = orc.Rng(b["seed"], b["stream"])
    assert [int(r.random_bool(b["p"])) for _ in b["v"]] == b["v"]
    # p == 1.0 never draws (rand Bernoulli ALWAYS_TRUE); p == 0.0 draws one u64 and is false
    r1, r2 = orc.Rng(5, 5), orc.Rng(5, 5)
    assert r1.random_bool(1.0) is True
    assert r1.next_u64() == r2.next_u64()
    r3, r4 = orc.Rng(5, 5), orc.Rng(5, 5)
    assert r3.random_bool(0.0) is False
    r4.next_u64()
    assert r3.next_u64() == r4.next_u64()


def test_space_kats(golden):
    for k in golden["space"]["kat"]:
        a = [unhex(v) for v in k["a"]]
        b = [unhex(v) for v in k["b"]]
        assert hexf(orc.distance(a, b)) == k["distance"]
        assert hexf(orc.distance(b, a)) == k["distance"]  # (a-b)^2 == (b-a)^2 exactly
        assert [hexf(v) for v in orc.interpolate(a, b, unhex(k["t"]))] == k["interpolate"]
    for e in golden["space"]["extent"]:
        assert hexf(orc.maximum_extent(e["bounds"])) == e["extent"]
    for s in golden["space"]["num_steps"]:
        assert orc.num_steps(unhex(s["dist"]), unhex(s["lvsl"])) == s["n"]
    # config 2 -> 6 checks, config 1 -> 4, reference test scenario -> 8
    assert orc.num_steps(0.5, math.sqrt(300.0) * 0.05) == 6
    assert orc.num_steps(0.5, math.sqrt(800.0) * 0.05) == 4
    assert orc.num_steps(0.5, math.sqrt(200.0) * 0.05) == 8
    # unbounded -> extent 1.0 (real_vector_state_space.rs:104-109)
    assert orc.maximum_extent([(-math.inf, math.inf), (0.0, 1.0)]) == 1.0
    # Rust saturating cast: NaN -> 0
    assert orc.num_steps(float("nan"), 1.0) == 0
    assert orc.num_steps(1.0, 0.0) == 2 ** 64 - 1


def _make(params, seed, pid, max_nodes=None, stop=True):
    p = orc.OracleRRT(params["dim"], params["bounds"], params["max_distance"], params["goal_bias"],
                      params["fraction"], max_nodes or params.get("max_nodes", 10000), stop, seed, pid)
    if params["spheres"]:
        p.set_spheres(*params_spheres(params))
    if params["boxes"]:
        p.set_boxes(*params_boxes(params))
    p.setup(params["start"], params["goal_c"], params["goal_r"])
    return p


def _check_run(p, run, status):
    states, parents = p.tree()
    assert p.num_nodes == run["n"]
    assert p.iterations == run["iterations"]
    assert p.accepted == run["accepted"]
    assert "%016x" % p.checksum == run["checksum"]
    assert p.goal_node == run["goal_node"]
    m = len(run["first_parents"])
    assert [[hexf(v) for v in row] for row in states[:m]] == run["first_states"]
    assert list(parents[:m]) == run["first_parents"]
    assert [[hexf(v) for v in row] for row in p.path()] == run["path"]
    assert status == orc.SOLVED


@pytest.mark.parametrize("key", ["config1", "wall"])
def test_oracle_trees_match_numpy_restatement(golden, key):
    params = golden[key]["params"]
    for run in golden[key]["runs"]:
        p = _make(params, run["seed"], run["pid"])
        st = p.solve(params["max_iterations"])
        _check_run(p, run, st)
        assert p.stop_reason == orc.STOP_GOAL
        # the reference's own assertions (oxmpl/tests/rrt_rvss_tests.rs:168-185)
        path = p.path()
        assert len(path) > 0
        assert orc.distance(path[0], params["start"]) < 1e-9
        assert orc.distance(path[-1], params["goal_c"]) <= params["goal_r"]
        assert is_path_valid(path, params["bounds"], params["fraction"], p.is_valid,
                             orc.maximum_extent, orc.num_steps, orc.interpolate, orc.distance)


def test_oracle_config2_growth(golden):
    params = golden["config2"]["params"]
    for run in golden["config2"]["runs"]:
        p = _make(params, run["seed"], run["pid"], max_nodes=10000, stop=False)
        st = p.solve(run["max_iterations"])
        _check_run(p, run, st)
        assert p.stop_reason == orc.STOP_ITERATIONS


def test_oracle_termination_and_errors():
    bounds = [(0.0, 10.0)] * 2
    p = orc.OracleRRT(2, bounds, 0.5, 0.0, 0.05, max_nodes=50, stop_at_goal=False, seed=1, problem_id=0)
    assert p.solve(10) == orc.PLANNER_UNINITIALISED  # rrt.rs:160-163
    p.setup([1.0, 1.0], [9.0, 9.0], 0.01)
    assert p.solve(10 ** 6) == orc.NO_SOLUTION_FOUND
    assert p.num_nodes == 50 and p.stop_reason == orc.STOP_NODES
    it = p.iterations
    assert p.solve(10) == orc.NO_SOLUTION_FOUND and p.iterations == it  # cap checked before any draw
    # resumed solve continues the same stream: 30+40 iterations == 70 iterations
    a = orc.OracleRRT(2, bounds, 0.5, 0.05, 0.05, 10000, False, 9, 4)
    b = orc.OracleRRT(2, bounds, 0.5, 0.05, 0.05, 10000, False, 9, 4)
    for q in (a, b):
        q.setup([1.0, 1.0], [9.0, 9.0], 0.5)
    a.solve(30); a.solve(40); b.solve(70)
    assert a.checksum == b.checksum and a.num_nodes == b.num_nodes
    # frozen ("steady") iterations never insert
    n0 = a.num_nodes
    a.solve(25, freeze=True)
    assert a.num_nodes == n0 and a.iterations == 95
    for bad, code in (([(0.0, math.inf)] * 2, orc.UNBOUNDED), ([(1.0, 1.0)] * 2, orc.ZERO_VOLUME)):
        with pytest.raises(ValueError, match=str(code)):
            orc.OracleRRT(2, bad, 0.5, 0.0)
    with pytest.raises(ValueError):
        orc.OracleRRT(2, bounds, 0.5, 1.5)  # Bernoulli::new rejects p outside [0,1]


def test_oracle_nearest_ties_lowest_index():
    # exact ties and sqrt-merged near-ties resolve to the lowest index (rrt.rs:192 strict '<')
    nodes = np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
    assert orc.nearest(nodes, [0.0, 0.0])[0] == 0
    assert orc.nearest(nodes[::-1].copy(), [0.0, 0.0])[0] == 0
    # d2 values one ulp apart whose square roots round to the same double: the reference
    # compares post-sqrt values, so the LOWER index wins although its d2 is larger
    y = 2.0 ** -26
    assert 1.0 + y * y != 1.0 and math.sqrt(1.0 + y * y) == 1.0
    i, d = orc.nearest(np.array([[1.0, y], [1.0, 0.0]]), [0.0, 0.0])
    assert (i, d) == (0, 1.0)
    i, d = orc.nearest(np.array([[1.0, 0.0], [1.0, y]]), [0.0, 0.0])
    assert (i, d) == (0, 1.0)


@pytest.mark.parametrize("key", ["connect_config1", "connect_wall"])
def test_oracle_rrt_connect_matches_numpy_restatement(golden, key):
    """RRTConnect (rrt_connect.rs:121-159,199-309): both trees, the merged path and the checksum"""
    params = golden[key]["params"]
    for run in golden[key]["runs"]:
        p = orc.OracleRRTConnect(params["dim"], params["bounds"], params["max_distance"], params["goal_bias"],
                                 params["fraction"], params["max_nodes"], run["seed"], run["pid"])
        if params["spheres"]:
            p.set_spheres(*params_spheres(params))
        if params["boxes"]:
            p.set_boxes(*params_boxes(params))
        assert p.solve(10) == orc.PLANNER_UNINITIALISED
        p.setup(params["start"], params["goal_c"], params["goal_r"])
        assert p.solve(params["max_iterations"]) == orc.SOLVED
        assert [p.num_nodes(0), p.num_nodes(1)] == run["n"]
        assert p.iterations == run["iterations"] and "%016x" % p.checksum == run["checksum"]
        assert [p.end_node(0), p.end_node(1)] == run["end"]
        for w in (0, 1):
            states, parents = p.tree(w)
            m = len(run["parents"][w])
            assert [[hexf(v) for v in row] for row in states[:m]] == run["states"][w]
            assert list(parents[:m]) == run["parents"][w]
        path = p.path()
        assert [[hexf(v) for v in row] for row in path] == run["path"]
        # the reference's assertions (oxmpl/tests/rrt_connect_rvss_tests.rs): start, goal, validity
        o = orc.OracleRRT(params["dim"], params["bounds"], 0.5, 0.0, params["fraction"], 10, True, 0, 0)
        if params["spheres"]:
            o.set_spheres(*params_spheres(params))
        if params["boxes"]:
            o.set_boxes(*params_boxes(params))
        assert orc.distance(path[0], params["start"]) < 1e-9
        assert orc.distance(path[-1], params["goal_c"]) <= params["goal_r"]
        assert is_path_valid(path, params["bounds"], params["fraction"], o.is_valid, orc.maximum_extent,
                             orc.num_steps, orc.interpolate, orc.distance)
        assert p.solve(100) == orc.SOLVED and p.iterations == run["iterations"]  # idempotent once solved
