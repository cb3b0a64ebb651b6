"""GPU parity tests of the PRM path (oxhip_prm_*, through the C ABI) against the CPU oracle
(oracle/prm_oracle.c) and the golden fixtures (tests/golden/prm_golden.json).  Bit-exact: milestone
bit patterns, every node's `edges` list in the reference's order, start connections, goal milestones
and the returned path.

PARITY UNPINNED against oxmpl itself (no reference-held vectors; the reference cannot be built here)."""
import numpy as np
import pytest

from helpers import unhex, bits, params_spheres, params_boxes, is_path_valid
from prm_helpers import csr_checksum, states_checksum, make_oracle_prm, STATUS_NAME

pytestmark = pytest.mark.gpu

from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402


def make_gpu_prm(P, **kw):
    args = dict(max_milestones=P["max_milestones"], lvs_fraction=P["fraction"], seed=P["seed"], stream=P["stream"],
                max_samples=0 if P["max_samples"] >= 10 ** 9 else P["max_samples"])
    args.update(kw)
    g = capi.PRMRoadmap(P["dim"], P["bounds"], P["radius"], **args)
    if P["spheres"]:
        g.set_spheres(*params_spheres(P))
    if P["boxes"]:
        g.set_boxes(*params_boxes(P))
    return g


def assert_same_roadmap(g, o):
    n, e, s = g.sizes()
    assert n == o.num_milestones and s == o.num_samples
    gs, goff, gn = g.roadmap()
    os_, ooff, on = o.roadmap()
    assert np.array_equal(bits(gs), bits(os_))
    assert np.array_equal(goff, ooff)
    assert np.array_equal(gn, on)
    return gs, goff, gn


def assert_same_query(g, o, timeout=0.0):
    st, path = g.solve(timeout)
    ost = o.solve()
    assert st == ost
    if st not in (capi.ERR_INVALID_START_STATE, capi.ERR_UNSAMPLED_STATE_SPACE, capi.ERR_PLANNER_UNINITIALISED):
        sc, gi = g.query_sets()
        assert np.array_equal(sc, o.start_connections())
        assert np.array_equal(gi, o.goal_indices())
    opath = o.path()
    assert path.shape == opath.shape and np.array_equal(bits(path), bits(opath))
    return st, path


@pytest.mark.parametrize("key", ["wall", "r3", "r6", "sample_cap"])
def test_prm_golden_scenes(prm_golden, key):
    P, R = prm_golden[key]["params"], prm_golden[key]["run"]
    q0 = R["queries"][0]
    g = make_gpu_prm(P)
    o = make_oracle_prm(P)
    assert g.solve()[0] == capi.ERR_PLANNER_UNINITIALISED               # prm.rs:229-236
    with pytest.raises(capi.OxhipError) as ei:
        g.construct_roadmap()
    assert ei.value.status == capi.ERR_PLANNER_UNINITIALISED            # prm.rs:97-104
    g.setup(q0["start"], q0["goal_c"], q0["goal_r"])
    o.setup(q0["start"], q0["goal_c"], q0["goal_r"])
    assert g.solve()[0] == capi.ERR_UNSAMPLED_STATE_SPACE               # prm.rs:239-241
    g.construct_roadmap()
    o.construct_roadmap(P["max_milestones"], P["max_samples"])
    gs, goff, gn = assert_same_roadmap(g, o)
    # ... and the committed fixture of the independent numpy restatement
    assert g.sizes() == (R["n"], R["edge_entries"], R["n_samples"])
    assert "%016x" % states_checksum(gs) == R["states_checksum"]
    assert "%016x" % csr_checksum(goff, gn) == R["csr_checksum"]
    for i, want in enumerate(R["edges_head"]):
        assert list(gn[int(goff[i]):int(goff[i + 1])]) == want
    # construct_roadmap again: "already constructed" (prm.rs:106-113)
    g.construct_roadmap()
    assert g.sizes() == (R["n"], R["edge_entries"], R["n_samples"])
    for q in R["queries"]:
        g.set_problem(q["start"], q["goal_c"], q["goal_r"])
        o.set_problem(q["start"], q["goal_c"], q["goal_r"])
        st, path = assert_same_query(g, o)
        assert STATUS_NAME[st] == q["status"]
        want = np.array([[unhex(v) for v in row] for row in q["path"]]).reshape(-1, P["dim"])
        assert path.shape == want.shape and np.array_equal(bits(path), bits(want))
    # setup() clears the roadmap (prm.rs:224) and restarts the stream: the same roadmap comes back
    g.setup(q0["start"], q0["goal_c"], q0["goal_r"])
    assert g.sizes()[0] == 0 and g.solve()[0] == capi.ERR_UNSAMPLED_STATE_SPACE
    g.construct_roadmap()
    assert_same_roadmap(g, o)


def test_prm_reference_test_properties(prm_golden):
    """oxmpl/tests/prm_rvss_tests.rs:162-204 on the device result"""
    P, R = prm_golden["wall"]["params"], prm_golden["wall"]["run"]
    q = R["queries"][0]
    g = make_gpu_prm(P)
    g.setup(q["start"], q["goal_c"], q["goal_r"])
    g.construct_roadmap()
    assert g.sizes()[0] > 0
    st, path = g.solve(5.0)
    assert st == capi.OK and len(path) > 0
    assert orc.distance(path[0], q["start"]) < 1e-9
    assert orc.distance(path[-1], q["goal_c"]) <= q["goal_r"]
    lo, hi = params_boxes(P)

    def valid(p):
        return not any(all(lo[b][k] <= p[k] <= hi[b][k] for k in range(2)) for b in range(len(lo)))

    assert is_path_valid(path, [tuple(b) for b in P["bounds"]], P["fraction"], valid, orc.maximum_extent,
                         orc.num_steps, orc.interpolate, orc.distance)


def _r6_params(n, radius, seed=11, stream=5, n_spheres=16):
    rng = np.random.default_rng(1234)
    centres = rng.uniform(1.0, 9.0, size=(n_spheres, 6))
    radii = rng.uniform(3.0, 4.5, size=n_spheres)
    return dict(dim=6, bounds=[(0.0, 10.0)] * 6, radius=radius, fraction=0.05, seed=seed, stream=stream,
                max_milestones=n, max_samples=10 ** 9, boxes=[],
                spheres=[(list(map(float, c)), float(r)) for c, r in zip(centres, radii)])


def test_prm_r6_4096_milestones_against_oracle():
    """BASELINE.json configs[4] shape (R^6, hyperspheres) at a size the oracle finishes in seconds"""
    P = _r6_params(4096, 3.0)
    g, o = make_gpu_prm(P), make_oracle_prm(P)
    s, gc = [2.0] * 6, [8.0] * 6
    g.setup(s, gc, 2.5)
    o.setup(s, gc, 2.5)
    g.construct_roadmap()
    o.construct_roadmap(4096)
    _, goff, gn = assert_same_roadmap(g, o)
    assert len(gn) > 4096          # a connected-ish roadmap, not a degenerate one
    assert_same_query(g, o)
    t = g.last_timing()
    assert t["redraw_batches"] == 0 and t["candidates"] * 2 >= len(gn)


def test_prm_time_bounded_rounds_give_the_same_roadmap():
    """with a wall-clock timeout the roadmap is built in doubling rounds (clock read in between, prm.rs:118);
    for the same final milestone count it must equal the one-round roadmap and the oracle's"""
    P = dict(dim=3, bounds=[(0.0, 10.0)] * 3, radius=0.9, fraction=0.05, seed=5, stream=9, max_milestones=10000,
             max_samples=10 ** 9, boxes=[([4.0, 0.0, 0.0], [6.0, 7.0, 10.0])],
             spheres=[([2.0, 8.0, 5.0], 1.5), ([8.0, 2.0, 5.0], 1.25)])
    g1, g2, o = make_gpu_prm(P), make_gpu_prm(P, timeout=3600.0), make_oracle_prm(P)
    for x in (g1, g2, o):
        x.setup([1.0, 1.0, 1.0], [9.0, 9.0, 9.0], 0.75)
    g1.construct_roadmap()
    g2.construct_roadmap()
    o.construct_roadmap(10000)
    assert_same_roadmap(g1, o)
    assert_same_roadmap(g2, o)
    assert_same_query(g1, o)
    assert_same_query(g2, o)


def test_prm_rejected_draws_replay_the_batch():
    """bounds where rand's range sampler rejects about a quarter of its draws (v * 4 + 1e16 rounds up to the
    upper bound): later samples start at shifted stream positions, the sequential replay path"""
    P = dict(dim=2, bounds=[(1e16, 1e16 + 4.0), (0.0, 10.0)], radius=3.0, fraction=0.05, seed=21, stream=3,
             max_milestones=1200, max_samples=10 ** 9, boxes=[([1e16 - 1.0, 4.0], [1e16 + 1.0, 4.5])], spheres=[])
    g, o = make_gpu_prm(P), make_oracle_prm(P)
    s, gc = [1e16, 1.0], [1e16 + 2.0, 9.0]
    g.setup(s, gc, 1.0)
    o.setup(s, gc, 1.0)
    g.construct_roadmap()
    o.construct_roadmap(1200)
    gs, _, _ = assert_same_roadmap(g, o)
    assert g.last_timing()["redraw_batches"] >= 2
    assert set(np.unique(gs[:, 0])) <= {1e16, 1e16 + 2.0}
    assert_same_query(g, o)


def test_prm_dense_roadmap_overflows_the_candidate_buffer_once():
    """every pair is within the radius: 4.5 M candidates > the initial buffer, so the pair search runs again
    with the exact size; no obstacle, so every candidate becomes an edge"""
    n = 3000
    P = dict(dim=2, bounds=[(0.0, 1.0), (0.0, 1.0)], radius=2.0, fraction=0.05, seed=2, stream=2, max_milestones=n,
             max_samples=10 ** 9, boxes=[], spheres=[])
    g = make_gpu_prm(P)
    g.setup([0.1, 0.1], [0.9, 0.9], 0.05)
    g.construct_roadmap()
    nn, e, s = g.sizes()
    assert (nn, e, s) == (n, n * (n - 1), n)
    _, off, nb = g.roadmap()
    assert np.array_equal(off, np.arange(n + 1, dtype=np.uint64) * (n - 1))
    want = np.arange(n, dtype=np.uint32)
    for i in (0, 1, 1499, n - 1):
        assert np.array_equal(nb[int(off[i]):int(off[i + 1])], want[want != i])
    assert g.last_timing()["candidates"] == n * (n - 1) // 2
    st, path = g.solve()
    assert st == capi.OK and len(path) >= 2


def test_prm_radius_edge_cases_and_errors():
    P = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], radius=0.0, fraction=0.05, seed=1, stream=1, max_milestones=500,
             max_samples=10 ** 9, boxes=[], spheres=[([5.0, 5.0], 1.0)])
    # radius 0: `dist < 0` never holds -> no edge, no start connection -> NoSolutionFound (prm.rs:266-268)
    g, o = make_gpu_prm(P), make_oracle_prm(P)
    for x in (g, o):
        x.setup([1.0, 1.0], [9.0, 9.0], 1.0)
    g.construct_roadmap()
    o.construct_roadmap(500)
    assert_same_roadmap(g, o)
    assert g.sizes()[1] == 0
    assert assert_same_query(g, o)[0] == capi.ERR_NO_SOLUTION_FOUND
    # invalid start (prm.rs:243-246)
    g.set_problem([5.0, 5.0], [9.0, 9.0], 1.0)
    o.set_problem([5.0, 5.0], [9.0, 9.0], 1.0)
    assert assert_same_query(g, o)[0] == capi.ERR_INVALID_START_STATE
    # the strict radius: a milestone at distance exactly r is not connected, one ulp closer is
    P2 = dict(P, radius=float("inf"), max_milestones=64, spheres=[])
    g2, o2 = make_gpu_prm(P2), make_oracle_prm(P2)
    for x in (g2, o2):
        x.setup([1.0, 1.0], [9.0, 9.0], 1.0)
    g2.construct_roadmap()
    o2.construct_roadmap(64)
    assert_same_roadmap(g2, o2)
    ms, _, _ = g2.roadmap()
    d = float(orc.distance(ms[1], ms[0]))
    for r, connected in ((d, False), (np.nextafter(d, np.inf), True)):
        P3 = dict(P, radius=float(r), max_milestones=2, spheres=[])
        g3, o3 = make_gpu_prm(P3), make_oracle_prm(P3)
        for x in (g3, o3):
            x.setup([1.0, 1.0], [9.0, 9.0], 1.0)
        g3.construct_roadmap()
        o3.construct_roadmap(2)
        assert_same_roadmap(g3, o3)
        assert (g3.sizes()[1] == 2) == connected
    # create-time validation mirrors RealVectorStateSpace::new / sample_uniform errors
    for bounds, status in (([(0.0, float("inf")), (0.0, 1.0)], capi.ERR_UNBOUNDED),
                           ([(1.0, 1.0), (0.0, 1.0)], capi.ERR_ZERO_VOLUME)):
        with pytest.raises(capi.OxhipError) as ei:
            capi.PRMRoadmap(2, bounds, 0.5, 100)
        assert ei.value.status == status


@pytest.mark.parametrize("dim,n_spheres,rmax", [(2, 150, 0.45), (3, 70, 1.2), (3, 64, 0.9)])
def test_prm_many_spheres_filter_is_conservative(dim, n_spheres, rmax):
    """more spheres than one 64-bit filter word, radii comparable to the connection radius: the midpoint filter
    of the edge check must never drop a sphere a motion actually touches (edge lists identical to the oracle)"""
    rng = np.random.default_rng(99 + dim)
    centres = rng.uniform(0.0, 10.0, size=(n_spheres, dim))
    radii = rng.uniform(0.05, rmax, size=n_spheres)
    radii[:3] = [-1.0, 0.0, 1e-300]   # never / only exactly at the centre / denormal-ish radius
    P = dict(dim=dim, bounds=[(0.0, 10.0)] * dim, radius=1.1 if dim == 2 else 1.6, fraction=0.05, seed=77, stream=dim,
             max_milestones=3000, max_samples=10 ** 9, boxes=[([0.0] * dim, [0.4] * dim)],
             spheres=[(list(map(float, c)), float(r)) for c, r in zip(centres, radii)])
    g, o = make_gpu_prm(P), make_oracle_prm(P)
    s, gc = [0.45] * dim, [9.5] * dim
    g.setup(s, gc, 1.0)
    o.setup(s, gc, 1.0)
    g.construct_roadmap()
    o.construct_roadmap(3000)
    _, goff, gn = assert_same_roadmap(g, o)
    t = g.last_timing()
    assert 2 * t["candidates"] > len(gn) > 0      # some in-radius pairs were rejected by check_motion
    assert_same_query(g, o)


@pytest.mark.parametrize("scale,offset", [(1.0, 1.0e3), (1.0, 1.0e6), (1.0e-12, 0.0), (1.0e18, 0.0), (1.0e60, 0.0)])
def test_prm_translated_and_scaled_spaces(scale, offset):
    """the radius search screens pairs in binary32 over an fl32 shadow of the milestones and decides in binary64; far
    from the origin the screen separates nothing (every pair is rechecked), beyond 1e15 it is switched off, 1e-12 is
    the subnormal end: the roadmap must not move by a bit"""
    dim = 3
    rng = np.random.default_rng(5)
    centres = rng.uniform(1.0, 9.0, size=(12, dim)) * scale + offset
    radii = rng.uniform(0.3, 0.9, size=12) * scale
    lo, hi = 0.0 * scale + offset, 10.0 * scale + offset
    P = dict(dim=dim, bounds=[(lo, hi)] * dim, radius=1.3 * scale, fraction=0.05, seed=123, stream=9,
             max_milestones=1500, max_samples=10 ** 9, boxes=[],
             spheres=[(list(map(float, c)), float(r)) for c, r in zip(centres, radii)])
    g, o = make_gpu_prm(P), make_oracle_prm(P)
    s, gc = [0.4 * scale + offset] * dim, [9.6 * scale + offset] * dim
    g.setup(s, gc, 1.0 * scale)
    o.setup(s, gc, 1.0 * scale)
    g.construct_roadmap()
    o.construct_roadmap(1500)
    _, goff, gn = assert_same_roadmap(g, o)
    assert len(gn) > 1500     # a connected-ish roadmap, not a degenerate one
    assert_same_query(g, o)


def test_prm_fully_blocked_space_returns_an_empty_roadmap():
    """no sample is ever valid: construct_roadmap must still return (the reference would run into its timeout)"""
    P = dict(dim=2, bounds=[(0.0, 1.0), (0.0, 1.0)], radius=0.5, fraction=0.05, seed=1, stream=1, max_milestones=50,
             max_samples=10 ** 9, boxes=[([-1.0, -1.0], [2.0, 2.0])], spheres=[])
    g = make_gpu_prm(P)
    g.setup([0.5, 0.5], [0.6, 0.6], 0.1)
    g.construct_roadmap()
    n, e, s = g.sizes()
    assert n == 0 and e == 0 and s == 4096 * 50 + (1 << 22)
    assert g.solve()[0] == capi.ERR_UNSAMPLED_STATE_SPACE


def test_config5_full_size_whole_roadmap_equals_oracle():
    """BASELINE.json configs[4] at its real size: R^6, 32 hyperspheres, 50,000 milestones, connection radius 2 -- every
    milestone bit pattern and every node's edge list against the CPU oracle (~4 s on one core)."""
    sc = scenarios.config5()
    g = scenarios.make_prm(sc, 50000)
    g.construct_roadmap()
    o = orc.OraclePRM(6, sc["bounds"], sc["connection_radius"], lvs_fraction=sc["lvs_fraction"], seed=42, stream=0)
    o.set_spheres(*sc["spheres"])
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    o.construct_roadmap(50000)
    gs, goff, gn = g.roadmap()
    os_, ooff, on = o.roadmap()
    assert gs.shape == (50000, 6) and len(gn) > 500000
    assert np.array_equal(gs.view(np.uint64), os_.view(np.uint64))
    assert np.array_equal(goff, ooff) and np.array_equal(gn, on)
    assert g.sizes()[2] == o.num_samples
    g.close()


def test_radius_screen_margin_sweep():
    """Directed test of the pair search's binary32 threshold screen: the connection radius is swept through the exact
    distance d of one milestone pair, r = d (1 +- eps) for eps from 0 to far beyond the screen's margin; the pair must be
    an edge candidate exactly when distance < r (strict, prm.rs:134) and the whole roadmap must equal the oracle's."""
    sc = scenarios.config5()
    base = scenarios.make_prm(sc, 1500)
    base.construct_roadmap()
    ms, off, nb = base.roadmap()
    base.close()
    i = 700
    j = int(nb[off[i]]) if off[i + 1] > off[i] else 0
    d = orc.distance(ms[i], ms[j])
    for eps in (0.0, 2.0 ** -52, 2.0 ** -40, 2.0 ** -26, 2.0 ** -23, 2.0 ** -21, 2.0 ** -19, 2.0 ** -17, 2.0 ** -15, 2.0 ** -12):
        for sign in (-1.0, 1.0):
            r = d * (1.0 + sign * eps)
            g = scenarios.make_prm(sc, 1500, connection_radius=r)
            g.construct_roadmap()
            o = orc.OraclePRM(6, sc["bounds"], r, lvs_fraction=sc["lvs_fraction"], seed=42, stream=0)
            o.set_spheres(*sc["spheres"])
            o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
            o.construct_roadmap(1500)
            gs, goff, gn = g.roadmap()
            os_, ooff, on = o.roadmap()
            assert np.array_equal(gs.view(np.uint64), os_.view(np.uint64)), (eps, sign)
            assert np.array_equal(goff, ooff) and np.array_equal(gn, on), (eps, sign)
            g.close()


# ---------------------------------------------------------------------------------------- the k-nearest variant
# BASELINE.json configs[4] words the roadmap as "all-pairs k-NN"; the reference connects by radius (prm.rs:131-138).  The variant
# (oxhip_prm_config.knn_k): a new milestone connects to its k nearest EARLIER milestones by (distance, index), visited in ascending
# index order -- defined in oracle/prm_oracle.c and, independently, tests/golden/make_golden_prm_knn.py.
@pytest.fixture(scope="module")
def prm_knn_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prm_knn_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("key", ["wall_k6", "r3_k10", "r6_k8", "wall_k1"])
def test_prm_knn_golden_scenes(prm_knn_golden, key):
    P, R = prm_knn_golden[key]["params"], prm_knn_golden[key]["run"]
    q0 = R["queries"][0]
    g = make_gpu_prm(P, knn_k=P["knn_k"])
    o = make_oracle_prm(P)
    for x in (g, o):
        x.setup(q0["start"], q0["goal_c"], q0["goal_r"])
    g.construct_roadmap()
    o.construct_roadmap(P["max_milestones"], P["max_samples"])
    gs, goff, gn = assert_same_roadmap(g, o)
    assert len(gn) == R["edge_entries"] and "%016x" % csr_checksum(goff, gn) == R["csr_checksum"]
    assert "%016x" % states_checksum(gs) == R["states_checksum"]
    for i, want in enumerate(R["edges_head"]):
        assert list(gn[int(goff[i]):int(goff[i + 1])]) == want
    for q in R["queries"]:
        g.set_problem(q["start"], q["goal_c"], q["goal_r"])
        o.set_problem(q["start"], q["goal_c"], q["goal_r"])
        st, path = assert_same_query(g, o)
        assert STATUS_NAME[st] == q["status"]
    g.close()


@pytest.mark.parametrize("k", [1, 8, 33])
def test_prm_knn_doubling_rounds_and_thin_free_space(k):
    """(i) with a wall clock the roadmap grows in doubling rounds: rows of a later round pick their neighbours among ALL earlier
    milestones; (ii) a free space that is a thin slab breaks the isotropic density estimate behind the candidate radius, so many rows
    fall short and take the exact search -- both must give the oracle's roadmap"""
    slab = dict(dim=3, bounds=[(0.0, 10.0)] * 3, radius=0.7, fraction=0.05, spheres=[], seed=5, stream=3, max_milestones=3000, max_samples=10 ** 9,
                boxes=[([0.0, 0.0, 0.0], [10.0, 10.0, 4.96]), ([0.0, 0.0, 5.0], [10.0, 10.0, 10.0])])
    g = make_gpu_prm(slab, knn_k=k, timeout=3600.0)   # (a timeout makes construction run in doubling rounds)
    o = make_oracle_prm(dict(slab, knn_k=k))
    for x in (g, o):
        x.setup([1.0, 1.0, 4.98], [9.0, 9.0, 4.98], 0.5)
    g.construct_roadmap()
    o.construct_roadmap(slab["max_milestones"])
    assert_same_roadmap(g, o)
    if k == 8:
        assert g.knn_exact_rows() > 0      # the estimate did fall short somewhere: the exact search ran
    assert_same_query(g, o)
    g.close()


def test_prm_knn_config5_full_size_whole_roadmap_equals_oracle():
    """BASELINE.json configs[4] as worded: R^6, 50,000 milestones, k = 8 nearest + edge validity -- the whole roadmap against the oracle"""
    sc = scenarios.config5()
    g = scenarios.make_prm(sc, knn_k=8)
    g.construct_roadmap()
    o = orc.OraclePRM(6, sc["bounds"], sc["connection_radius"], lvs_fraction=0.05, seed=42, stream=0)
    o.set_spheres(*sc["spheres"])
    o.set_knn(8)
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    o.construct_roadmap(sc["max_milestones"])
    assert_same_roadmap(g, o)
    t = g.last_timing()
    print("k-NN PRM 50,000 x R^6, k = 8: phases (ms) %s, candidates %d, exact rows %d" % (["%.3f" % v for v in t["phase_ms"]], t["candidates"], g.knn_exact_rows()))
    assert_same_query(g, o)
    g.close()
