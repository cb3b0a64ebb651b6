"""GPU parity tests of RRT* (planner kind 2) through the C ABI against the CPU oracle (orc_rrts_*) and the golden
fixtures: node count, iteration count, checksum (iteration polynomial over nearest / q_new / verdict + wiring polynomial over
chosen parent / cost bits / rewired count and index sum), tree bits, parents after rewiring, costs, path.
Every test runs on the designs "decoupled" (geometry by rrt_cells.hip or rrt_lanes.hip, wiring by rrt_star_wire.hip) and
"one_kernel" (KERNEL_STREAM: rrt_star.hip).  PARITY UNPINNED against oxmpl itself."""
import json
import os

import numpy as np
import pytest

from helpers import unhex, bits, params_spheres, params_boxes

pytestmark = pytest.mark.gpu

from oxmpl_amd import capi  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DESIGN = {"kernel": capi.KERNEL_AUTO}


@pytest.fixture(autouse=True, params=["decoupled", "decoupled_lanes", "one_kernel"])
def star_design(request):
    # decoupled: geometry by rrt_cells.hip (what KERNEL_AUTO runs for batches of more than 256 problems in R^2 / R^3) + rrt_star_wire.hip;
    # decoupled_lanes: geometry by rrt_lanes.hip (KERNEL_AUTO's choice for small batches and R^4 .. R^6)
    DESIGN["kernel"] = {"decoupled": capi.KERNEL_CELLS, "decoupled_lanes": capi.KERNEL_LANES, "one_kernel": capi.KERNEL_STREAM}[request.param]
    yield request.param
    DESIGN["kernel"] = capi.KERNEL_AUTO


@pytest.fixture(scope="module")
def star_golden():
    with open(os.path.join(ROOT, "tests", "golden", "rrt_star_golden.json")) as f:
        return json.load(f)


def make_oracle(P, seed, pid, stop=True, max_nodes=None):
    o = orc.OracleRRTStar(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], P["search_radius"], P["fraction"],
                          max_nodes or P["max_nodes"], stop, seed, pid)
    if P["spheres"]:
        o.set_spheres(*params_spheres(P))
    if P["boxes"]:
        o.set_boxes(*params_boxes(P))
    o.setup(P["start"], P["goal_c"], P["goal_r"])
    return o


def make_gpu(P, n_problems, seed, first_pid, stop=True, max_nodes=None, **extra):
    g = capi.RRTBatch(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], n_problems, max_nodes or P["max_nodes"],
                      P["fraction"], stop, seed, first_pid, 0, DESIGN["kernel"], capi.PLANNER_RRT_STAR, P["search_radius"], **extra)
    if P["spheres"]:
        g.set_spheres(*params_spheres(P))
    if P["boxes"]:
        g.set_boxes(*params_boxes(P))
    g.setup(P["start"], P["goal_c"], P["goal_r"])
    return g


def assert_same(g, p, o, c=None):
    c = c or g.counts()
    assert int(c["nodes"][p]) == o.num_nodes
    assert int(c["iterations"][p]) == o.iterations
    assert int(c["accepted"][p]) == o.accepted
    assert int(c["checksum"][p]) == o.checksum
    assert int(c["goal_node"][p]) == o.goal_node
    gs, gp = g.tree(p)
    os_, op = o.tree()
    assert np.array_equal(bits(gs), bits(os_)) and np.array_equal(gp, op)
    assert np.array_equal(bits(g.costs(p)), bits(o.costs()))
    gpath, opath = g.path(p), o.path()
    assert gpath.shape == opath.shape and np.array_equal(bits(gpath), bits(opath))


@pytest.mark.parametrize("key", ["wall", "wall_ref", "config1", "config2"])
def test_rrt_star_golden_scenes(star_golden, key):
    P = star_golden[key]["params"]
    stop = key != "config2"
    for r in star_golden[key]["runs"]:
        g = make_gpu(P, 1, r["seed"], r["pid"], stop)
        st = g.solve(P["max_iterations"])
        c = g.counts()
        assert int(c["nodes"][0]) == r["n"] and int(c["iterations"][0]) == r["iterations"]
        assert "%016x" % int(c["checksum"][0]) == r["checksum"] and int(c["goal_node"][0]) == r["goal_node"]
        assert (st[0] == capi.OK) == (r["goal_node"] >= 0)
        gs, gp = g.tree(0)
        head = np.array([[unhex(v) for v in row] for row in r["states"]])
        assert np.array_equal(bits(gs[:len(head)]), bits(head))
        assert list(gp) == r["parents"]
        assert np.array_equal(bits(g.costs(0)), bits(np.array([unhex(v) for v in r["cost"]])))
        want = np.array([[unhex(v) for v in row] for row in r["path"]]).reshape(-1, P["dim"])
        got = g.path(0)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want))
        o = make_oracle(P, r["seed"], r["pid"], stop)
        o.solve(P["max_iterations"])
        assert_same(g, 0, o, c)
        g.close()


def test_rrt_star_batch_of_problems_in_the_sphere_field(star_golden):
    """48 problems of the config-2 field, 1500 iterations each, grown in two solve calls (resume)"""
    P = star_golden["config2"]["params"]
    n_prob = 48
    g = make_gpu(P, n_prob, 42, 100, stop=False)
    g.solve(700)
    g.solve(800)
    c = g.counts()
    for p in range(0, n_prob, 5):
        o = make_oracle(P, 42, 100 + p, stop=False)
        o.solve(1500)
        assert_same(g, p, o, c)
    assert int(c["iterations"].min()) == 1500


@pytest.mark.parametrize("radius", [0.0, 0.3, 2.5, float("inf")])
def test_rrt_star_search_radius_extremes(star_golden, radius):
    """radius 0: no neighbour ever (RRT with costs); inf: every node is a neighbour of every new node"""
    P = dict(star_golden["config1"]["params"], search_radius=radius, max_nodes=400)
    g = make_gpu(P, 4, 9, 0, stop=False)
    g.solve(10 ** 6)
    c = g.counts()
    assert (c["stop_reason"] == capi.STOP_NODES).all() and (c["nodes"] == 400).all()
    for p in range(4):
        o = make_oracle(P, 9, p, stop=False)
        o.solve(10 ** 6)
        assert o.num_nodes == 400
        assert_same(g, p, o, c)
    if radius == 0.0:
        # without neighbours the tree is exactly RRT's
        r = orc.OracleRRT(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], P["fraction"], 400, False, 9, 0)
        r.set_spheres(*params_spheres(P))
        r.setup(P["start"], P["goal_c"], P["goal_r"])
        r.solve(10 ** 6)
        gs, gp = g.tree(0)
        rs, rp = r.tree()
        assert np.array_equal(bits(gs), bits(rs)) and np.array_equal(gp, rp)


@pytest.mark.parametrize("scale,offset", [(1.0, 1.0e3), (1.0, 1.0e6), (1.0e-12, 0.0), (1.0e18, 0.0), (1.0e40, 0.0)])
def test_rrt_star_translated_and_scaled_spaces(star_golden, scale, offset):
    """The sphere-field scene moved or scaled into the regimes of the binary32 screens' error model (nearest scan and
    radius search both screen over an fl32 shadow of the tree and decide in binary64): far from the origin binary32
    separates nothing and every scan takes the binary64 path; beyond 1e15 the screen is switched off; 1e-12 sits at
    the subnormal end.  Trees, parents after rewiring and costs must not move by a bit."""
    B = star_golden["config2"]["params"]
    t = lambda v: [float(x) * scale + offset for x in v]
    P = dict(B)
    P["bounds"] = [t(b) for b in B["bounds"]]
    P["start"], P["goal_c"] = t(B["start"]), t(B["goal_c"])
    P["max_distance"], P["goal_r"], P["search_radius"] = B["max_distance"] * scale, B["goal_r"] * scale, B["search_radius"] * scale
    c, r = params_spheres(B)
    c, r = np.asarray(c) * scale + offset, np.asarray(r) * scale
    n_prob = 4
    g = capi.RRTBatch(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], n_prob, 2000, P["fraction"], False, 9, 300, 0,
                      DESIGN["kernel"], capi.PLANNER_RRT_STAR, P["search_radius"])
    g.set_spheres(c, r)
    g.setup(P["start"], P["goal_c"], P["goal_r"])
    g.solve(600)
    cts = g.counts()
    for p in range(n_prob):
        o = orc.OracleRRTStar(P["dim"], P["bounds"], P["max_distance"], P["goal_bias"], P["search_radius"], P["fraction"],
                              2000, False, 9, 300 + p)
        o.set_spheres(c, r)
        o.setup(P["start"], P["goal_c"], P["goal_r"])
        o.solve(600)
        assert o.num_nodes > 100
        assert_same(g, p, o, cts)
    g.close()


def test_rrt_star_rewiring_shortens_costs(star_golden):
    """a property of the algorithm, checked on the device result: with a useful search radius the goal node's
    cost-to-come is not worse than without rewiring / parent choice (radius 0) on the same sample stream"""
    base = dict(star_golden["wall"]["params"], max_nodes=3000)
    res = {}
    for radius in (0.0, 1.0):
        P = dict(base, search_radius=radius)
        g = make_gpu(P, 8, 5, 0, stop=False)
        g.solve(10 ** 6)
        c = g.counts()
        assert (c["goal_node"] >= 0).all()
        res[radius] = np.array([g.costs(p)[int(c["goal_node"][p])] for p in range(8)])
    assert res[1.0].mean() < res[0.0].mean()


def test_rrt_star_argument_validation():
    with pytest.raises(capi.OxhipError) as ei:
        capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.5, 0.05, 1, 100, planner=capi.PLANNER_RRT_STAR, search_radius=float("nan"))
    assert ei.value.status == capi.ERR_BAD_ARG
    with pytest.raises(capi.OxhipError) as ei:
        capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.5, 0.05, 1, 100, planner=capi.PLANNER_RRT_STAR, kernel=capi.KERNEL_RESIDENT)
    assert ei.value.status == capi.ERR_BAD_ARG
    g = capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.5, 0.05, 1, 100, planner=capi.PLANNER_RRT, search_radius=1.0)
    g.setup([0.1, 0.1], [0.9, 0.9], 0.1)
    with pytest.raises(capi.OxhipError):
        g.costs(0)          # not an RRT* batch


@pytest.mark.parametrize("flags", [0, capi.DEBUG_STAR_TWO_PASS | capi.DEBUG_STAR_ONE_SEGMENT], ids=["default", "two_pass_one_segment"])
def test_rrt_star_wiring_in_many_rounds(star_golden, star_design, flags):
    """the decoupled design wires, per round, the longest prefix of a problem's pending nodes whose neighbour lists fit its
    pool segment: with the segment cut to its minimum (one list's worst case) and every node a neighbour of every later one,
    a 900-node tree needs hundreds of rounds -- and must come out exactly as in one"""
    if not star_design.startswith("decoupled"):
        pytest.skip("the pool belongs to the decoupled design")
    P = dict(star_golden["config2"]["params"], search_radius=float("inf"), max_nodes=900)
    g = make_gpu(P, 3, 77, 5, stop=False, star_pool_share=1, debug_flags=flags)   # (the share is clamped up to the node capacity: 1024 entries)
    g.solve(400)
    g.solve(10 ** 6)
    c = g.counts()
    assert (c["nodes"] == 900).all()
    for p in range(3):
        o = make_oracle(P, 77, 5 + p, stop=False)
        o.solve(10 ** 6)
        assert_same(g, p, o, c)


def test_rrt_star_full_size_config2(star_golden, star_design):
    """BASELINE.json configs[1]'s scene at its full tree size: 24 problems grown to 10,000 nodes with search radius 1 (34
    neighbours per node on average, up to ~580 around the goal); three of them against the oracle -- every node, every
    parent after rewiring, every cost, the checksum -- and all of them through properties (costs decrease along parent
    links by exactly the edge length's contribution being non-negative; every parent precedes nothing it should not)"""
    P = dict(star_golden["config2"]["params"], search_radius=1.0, max_nodes=10000)
    n_prob = 24
    g = make_gpu(P, n_prob, 42, 0, stop=False)
    g.solve(10 ** 9)
    c = g.counts()
    assert (c["nodes"] == 10000).all()
    for p in (0, 7, 23):
        o = make_oracle(P, 42, p, stop=False)
        o.solve(10 ** 9)
        assert_same(g, p, o, c)
    for p in range(n_prob):
        _, par = g.tree(p)
        cost = g.costs(p)
        assert par[0] == -1 and cost[0] == 0.0
        assert (par[1:] >= 0).all() and (par[1:] < 10000).all()
        assert (cost[1:] >= cost[par[1:]]).all()   # cost = parent's cost + a distance (>= : a node may repeat its parent's position)
        st, _ = g.tree(p)
        moved = (st[1:] != st[par[1:]]).any(axis=1)
        assert (cost[1:][moved] > cost[par[1:]][moved]).all()   # ... and strictly more wherever the edge has a length


def test_rrt_star_large_row_instantiation(star_golden, star_design):
    """a 13,000-node RRT* tree: the decoupled design's geometry then comes from the lane-per-query kernel's 32-row instantiation"""
    P = dict(star_golden["config2"]["params"], search_radius=0.6, max_nodes=13000)
    g = make_gpu(P, 2, 5, 9, stop=False)
    g.solve(10 ** 9)
    c = g.counts()
    assert (c["nodes"] == 13000).all()
    if star_design.startswith("decoupled"):   # (AUTO: the geometry comes from rrt_cells.hip in R^2 / R^3)
        assert g.last_timing()["kernel"] == (capi.KERNEL_CELLS if star_design == "decoupled" else capi.KERNEL_LANES)
    o = make_oracle(P, 5, 10, stop=False)
    o.solve(10 ** 9)
    assert_same(g, 1, o, c)
