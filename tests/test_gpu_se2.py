"""GPU parity tests of BASELINE.json configs[3]: RRTConnect over the build-defined SE(2) space among line
segments (rrt_connect_se2.hip) through the C ABI, against the CPU oracle (oracle/se2_oracle.c) and the golden
fixtures (tests/golden/se2_golden.json).  Bit-exact: both trees, checksums, merged paths, and the SO(2) / SE(2)
arithmetic itself (fmod-based normalisation on the device).
PARITY UNPINNED against oxmpl: the reference has no SE(2) space (docs/BACKLOG.md:12-14)."""
import json
import math
import os

import numpy as np
import pytest

from helpers import unhex, bits

pytestmark = pytest.mark.gpu

from oxmpl_amd import capi  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def se2_golden():
    with open(os.path.join(ROOT, "tests", "golden", "se2_golden.json")) as f:
        return json.load(f)


def segs_of(P):
    return np.array([[unhex(v) for v in s] for s in P["segments"]], dtype=np.float64).reshape(-1, 4)


def make_oracle(P, seed, pid, max_nodes=None):
    o = orc.OracleSE2Connect(P["bounds_xy"], P["theta_bounds"], P["max_distance"], P["goal_bias"], P["fraction"],
                             max_nodes or P["max_nodes"], seed, pid)
    o.set_segments(segs_of(P), P["clearance"])
    o.setup(P["start"], P["goal"], P["goal_r"])
    return o


def make_gpu(P, n_problems, seed, first_pid, max_nodes=None, debug_flags=0):
    bounds = list(P["bounds_xy"]) + [tuple(P["theta_bounds"])]
    g = capi.RRTBatch(3, bounds, P["max_distance"], P["goal_bias"], n_problems, max_nodes or P["max_nodes"], P["fraction"],
                      True, seed, first_pid, 0, capi.KERNEL_AUTO, capi.PLANNER_RRT_CONNECT, 0.0, capi.SPACE_SE2,
                      debug_flags=debug_flags)
    g.set_segments(segs_of(P), P["clearance"])
    g.setup(P["start"], P["goal"], P["goal_r"])
    return g


def assert_same(g, p, o, c=None, gc=None):
    c = c or g.counts()
    gc = gc or g.goal_counts()
    assert int(c["nodes"][p]) == o.num_nodes(0) and int(gc["nodes"][p]) == o.num_nodes(1)
    assert int(c["iterations"][p]) == o.iterations and int(c["checksum"][p]) == o.checksum
    assert int(c["goal_node"][p]) == o.end_node(0) and int(gc["end_node"][p]) == o.end_node(1)
    for w, (gs, gp) in enumerate((g.tree(p), g.goal_tree(p))):
        os_, op = o.tree(w)
        assert np.array_equal(bits(gs), bits(os_)) and np.array_equal(gp, op)
    gpath, opath = g.path(p), o.path()
    assert gpath.shape == opath.shape and np.array_equal(bits(gpath), bits(opath))


def test_so2_se2_arithmetic_on_the_device(se2_golden):
    """normalise / distance / interpolate bit for bit, incl. the wrap-around edges and large angles (fmod)"""
    rng = np.random.default_rng(7)
    n = 20000
    a = np.column_stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-7, 7, n)])
    b = np.column_stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-7, 7, n)])
    # edges: +-PI, values one ulp around them, huge and tiny angles
    pi = math.pi
    edge = np.array([pi, -pi, np.nextafter(pi, 0), np.nextafter(-pi, 0), np.nextafter(pi, 4), 3 * pi, -5 * pi, 1e6, -1e9,
                     1e-300, -0.0, 0.0, 2 * pi, -2 * pi, 1e15])
    m = len(edge)
    a[:m * m, 2] = np.repeat(edge, m)
    b[:m * m, 2] = np.tile(edge, m)
    t = rng.uniform(0, 1, n)
    t[:64] = np.linspace(0.0, 1.0, 64)
    out0 = capi.se2_op_batch(0, a, b)
    out1 = capi.se2_op_batch(1, a, b, t)
    L = orc.lib()
    for i in list(range(m * m)) + list(range(m * m, n, 37)):
        assert out0[i, 0] == orc.se2_distance(a[i], b[i]) or (math.isnan(out0[i, 0]))
        assert bits(out0[i, 1:2])[0] == bits(np.array([L.orc_so2_normalise(a[i, 2])]))[0]
        assert bits(out0[i, 2:3])[0] == bits(np.array([L.orc_so2_distance(a[i, 2], b[i, 2])]))[0]
        assert np.array_equal(bits(out1[i]), bits(orc.se2_interpolate(a[i], b[i], float(t[i]))))
    # the reference's own exact unit-test vector (so2_state.rs:80-87): normalise(3 PI / 2) == -PI / 2
    assert capi.se2_op_batch(0, [[0.0, 0.0, 3.0 * math.pi / 2.0]], [[0.0, 0.0, 0.0]])[0, 1] == -math.pi / 2.0
    for k in se2_golden["kat"]["random"]:
        x, y = [unhex(v) for v in k["a"]], [unhex(v) for v in k["b"]]
        o0 = capi.se2_op_batch(0, [x], [y])[0]
        assert [("%016x" % int(v)) for v in bits(o0)] == [k["se2_distance"], k["so2_normalise"], k["so2_distance"]]
        o1 = capi.se2_op_batch(1, [x], [y], [unhex(k["t"])])[0]
        assert [("%016x" % int(v)) for v in bits(o1)] == k["se2_interpolate"]


@pytest.mark.parametrize("key", ["soup256", "gap"])
def test_se2_connect_golden_scenes(se2_golden, key):
    P = se2_golden[key]["params"]
    for r in se2_golden[key]["runs"]:
        g = make_gpu(P, 1, r["seed"], r["pid"])
        st = g.solve(P["max_iterations"])
        assert st[0] == capi.OK
        c, gc = g.counts(), g.goal_counts()
        assert [int(c["nodes"][0]), int(gc["nodes"][0])] == r["n"] and int(c["iterations"][0]) == r["iterations"]
        assert "%016x" % int(c["checksum"][0]) == r["checksum"]
        assert [int(c["goal_node"][0]), int(gc["end_node"][0])] == r["end"]
        want = np.array([[unhex(v) for v in row] for row in r["path"]]).reshape(-1, 3)
        got = g.path(0)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want))
        o = make_oracle(P, r["seed"], r["pid"])
        o.solve(P["max_iterations"])
        assert_same(g, 0, o, c, gc)
        # the checker and the motion check, batched, on the path
        assert g.is_valid(got).all() and g.check_motion(got[:-1], got[1:]).all()
        g.close()


def test_se2_connect_batch_and_checker_parity(se2_golden):
    """64 problems of the 256-segment scene in one launch; is_valid / check_motion against the oracle on random states"""
    P = se2_golden["soup256"]["params"]
    g = make_gpu(P, 64, 11, 500)
    st = g.solve(10 ** 6)
    assert (st == capi.OK).all()
    c, gc = g.counts(), g.goal_counts()
    for p in range(0, 64, 7):
        o = make_oracle(P, 11, 500 + p)
        assert o.solve(10 ** 6) == orc.SOLVED
        assert_same(g, p, o, c, gc)
    rng = np.random.default_rng(3)
    n = 4000
    a = np.column_stack([rng.uniform(0, 10, n), rng.uniform(0, 10, n), rng.uniform(-math.pi, math.pi, n)])
    b = a + np.column_stack([rng.normal(0, 0.4, n), rng.normal(0, 0.4, n), rng.normal(0, 1.5, n)])
    o = make_oracle(P, 0, 0)
    v = g.is_valid(a)
    m = g.check_motion(a, b)
    assert 0.2 < v.mean() < 0.98                        # the scene is neither empty nor full
    for i in range(0, n, 3):
        assert bool(v[i]) == o.is_valid(a[i])
        assert bool(m[i]) == o.check_motion(a[i], b[i])


def test_se2_node_cap_and_argument_validation(se2_golden):
    P = dict(se2_golden["gap"]["params"])
    # goal unreachable: enclosed by four segments -> both trees fill up to the node cap
    box = [(8.0, 4.0, 10.0, 4.0), (8.0, 6.0, 10.0, 6.0), (8.0, 4.0, 8.0, 6.0), (10.0, 4.0, 10.0, 6.0)]
    P["segments"] = [["%016x" % int(bits(np.array([v]))[0]) for v in s] for s in box]
    g = make_gpu(P, 2, 1, 0, max_nodes=300)
    st = g.solve(10 ** 6)
    c, gc = g.counts(), g.goal_counts()
    assert (st == capi.ERR_NO_SOLUTION_FOUND).all() and (c["stop_reason"] == capi.STOP_NODES).all()
    for p in range(2):
        o = make_oracle(P, 1, p, max_nodes=300)
        assert o.solve(10 ** 6) == orc.NO_SOLUTION_FOUND
        assert_same(g, p, o, c, gc)
    for kw, status in ((dict(dim=2, bounds=[(0.0, 1.0)] * 2), capi.ERR_BAD_ARG),                       # (x, y, theta) needs dim 3
                       (dict(bounds=[(0.0, 1.0), (0.0, 1.0), (1.0, 1.0)]), capi.ERR_ZERO_VOLUME),       # so2_state_space.rs:59-64
                       (dict(planner=capi.PLANNER_RRT), capi.ERR_BAD_ARG)):
        args = dict(dim=3, bounds=[(0.0, 1.0), (0.0, 1.0), (-1.0, 1.0)], max_distance=0.5, goal_bias=0.05, n_problems=1,
                    planner=capi.PLANNER_RRT_CONNECT, space=capi.SPACE_SE2)
        args.update(kw)
        with pytest.raises(capi.OxhipError) as ei:
            capi.RRTBatch(**args)
        assert ei.value.status == status


def hexsegs(segs):
    return [["%016x" % int(bits(np.array([float(v)]))[0]) for v in s] for s in segs]


def test_se2_without_the_segment_grid_and_in_chunks(se2_golden):
    """The grid lookup of the motion check switched off (every state against every segment), and a solve cut into launches of 37
    iterations (the block sampler starts over inside its 64-iteration blocks): same trees, checksums and paths"""
    P = se2_golden["soup256"]["params"]
    ref = make_gpu(P, 12, 21, 40)
    assert (ref.solve(10 ** 6) == capi.OK).all()
    c0, gc0 = ref.counts(), ref.goal_counts()
    plain = make_gpu(P, 12, 21, 40, debug_flags=capi.DEBUG_SE2_NO_SEGMENT_GRID)
    assert (plain.solve(10 ** 6) == capi.OK).all()
    small = make_gpu(P, 12, 21, 40, debug_flags=capi.DEBUG_SE2_SMALL_LDS)   # the shape of batches larger than the chip
    assert (small.solve(10 ** 6) == capi.OK).all()
    chunked = make_gpu(P, 12, 21, 40)
    for _ in range(4000):
        st = chunked.solve(37)
        if (st == capi.OK).all():
            break
    assert (st == capi.OK).all()
    for g in (plain, small, chunked):
        c, gc = g.counts(), g.goal_counts()
        for k in ("nodes", "iterations", "checksum", "goal_node"):
            assert np.array_equal(c[k], c0[k]), k
        assert np.array_equal(gc["nodes"], gc0["nodes"]) and np.array_equal(gc["end_node"], gc0["end_node"])
        for p in (0, 5, 11):
            assert np.array_equal(bits(g.path(p)), bits(ref.path(p)))
    for p in (0, 11):
        o = make_oracle(P, 21, 40 + p)
        assert o.solve(10 ** 6) == orc.SOLVED
        assert_same(chunked, p, o)


def test_se2_trees_beyond_the_lds_shadow_and_odd_headings(se2_golden):
    """Both trees grown to 2,000 nodes (the binary32 shadow holds 768: the rest is evaluated exactly, candidates come back from
    HBM); a start heading outside [-PI, PI] (the screen's heading formula does not apply: every node exactly)"""
    P = dict(se2_golden["gap"]["params"])
    box = [(8.0, 4.0, 10.0, 4.0), (8.0, 6.0, 10.0, 6.0), (8.0, 4.0, 8.0, 6.0), (10.0, 4.0, 10.0, 6.0)]
    P["segments"] = hexsegs(box)
    for flags in (0, capi.DEBUG_SE2_SMALL_LDS):
        g = make_gpu(P, 2, 5, 0, max_nodes=2000, debug_flags=flags)
        st = g.solve(10 ** 6)
        c, gc = g.counts(), g.goal_counts()
        assert (st == capi.ERR_NO_SOLUTION_FOUND).all() and (c["stop_reason"] == capi.STOP_NODES).all()
        assert int(max(c["nodes"].max(), gc["nodes"].max())) == 2000
        for p in range(2):
            o = make_oracle(P, 5, p, max_nodes=2000)
            assert o.solve(10 ** 6) == orc.NO_SOLUTION_FOUND
            assert_same(g, p, o, c, gc)
    Q = dict(se2_golden["soup256"]["params"])
    Q["start"] = [Q["start"][0], Q["start"][1], 4.0]
    g = make_gpu(Q, 3, 8, 0)
    assert (g.solve(10 ** 6) == capi.OK).all()
    for p in range(3):
        o = make_oracle(Q, 8, p)
        assert o.solve(10 ** 6) == orc.SOLVED
        assert_same(g, p, o)


def test_se2_crowded_cells_and_soups_beyond_the_lds_table(se2_golden):
    """A cell of the lookup grid that more than eight segments reach (its states meet every segment), and a soup of more than
    256 segments (the table stays in HBM / L2): planner runs and the stand-alone checks against the oracle"""
    P = dict(se2_golden["soup256"]["params"])
    base = segs_of(P)
    rng = np.random.default_rng(12)
    knot = np.column_stack([5.0 + rng.uniform(-0.05, 0.05, 14), 5.0 + rng.uniform(-0.05, 0.05, 14),
                            5.0 + rng.uniform(-0.05, 0.05, 14), 5.0 + rng.uniform(-0.05, 0.05, 14)])
    for segs in (np.vstack([base[:200], knot]), np.vstack([base, knot, base[:60] + 0.013])):
        Q = dict(P)
        Q["segments"] = hexsegs(segs)
        g = make_gpu(Q, 4, 31, 0)
        st = g.solve(200000)
        c, gc = g.counts(), g.goal_counts()
        for p in range(4):
            o = make_oracle(Q, 31, p)
            o.solve(200000)
            assert_same(g, p, o, c, gc)
        n = 1500
        a = np.column_stack([rng.uniform(4.5, 5.5, n), rng.uniform(4.5, 5.5, n), rng.uniform(-3, 3, n)])
        b = a + np.column_stack([rng.normal(0, 0.3, n), rng.normal(0, 0.3, n), rng.normal(0, 1.0, n)])
        m = g.check_motion(a, b)
        o = make_oracle(Q, 0, 0)
        assert 0.02 < m.mean() < 0.98
        for i in range(n):
            assert bool(m[i]) == o.check_motion(a[i], b[i])
