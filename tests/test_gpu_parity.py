"""GPU parity tests: the HIP path (through the C ABI, ctypes) against the CPU oracle and the
golden fixtures.  Bit-exact for every integer, index and f64 bit pattern.

PARITY UNPINNED against oxmpl itself: the oracle is our C restatement (oracle/rrt_oracle.h);
the reference holds no vectors for this path and cannot be built here.
"""
import math
import os

import numpy as np
import pytest

from helpers import unhex, hexf, bits, params_spheres, params_boxes, is_path_valid

pytestmark = pytest.mark.gpu

from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

# KERNEL_AUTO = what a user gets (the lane-per-query kernel where it exists); KERNEL_RESIDENT = the all-binary64 cross-check
KERNELS = [capi.KERNEL_STREAM, capi.KERNEL_RESIDENT, capi.KERNEL_LANES, capi.KERNEL_CELLS, capi.KERNEL_AUTO]
KNAME = {capi.KERNEL_STREAM: "stream", capi.KERNEL_RESIDENT: "resident", capi.KERNEL_LANES: "lanes", capi.KERNEL_CELLS: "cells", capi.KERNEL_AUTO: "auto"}
SCREENED = (capi.KERNEL_LANES, capi.KERNEL_CELLS)   # kernels that count their exact-path events (stamps()[4])


def _batch_or_skip(*args, **kw):
    try:
        return capi.RRTBatch(*args, **kw)
    except capi.OxhipError as e:
        if e.status == capi.ERR_BAD_ARG and ("resident" in str(e) or "cell-grid kernel" in str(e)):
            pytest.skip("this kernel has no instantiation for this shape")
        raise


def _oracle_for(sc, seed, pid, max_nodes, stop):
    p = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                      max_nodes, stop, seed, pid)
    if sc["spheres"] is not None:
        p.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        p.set_boxes(*sc["boxes"])
    p.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return p


def _gpu_for(sc, n_problems, max_nodes, stop, seed, first_pid=0, kernel=0):
    try:
        return scenarios.make_batch(sc, n_problems, max_nodes, stop, seed, first_pid, 0, kernel)
    except capi.OxhipError as e:
        if e.status == capi.ERR_BAD_ARG and ("resident" in str(e) or "cell-grid kernel" in str(e)):
            pytest.skip("this kernel has no instantiation for this shape")
        raise


def _assert_same_problem(gpu, p, o, c=None):
    c = c or gpu.counts()
    assert int(c["nodes"][p]) == o.num_nodes
    assert int(c["iterations"][p]) == o.iterations
    assert int(c["accepted"][p]) == o.accepted
    assert int(c["checksum"][p]) == o.checksum
    assert int(c["goal_node"][p]) == o.goal_node
    gs, gp = gpu.tree(p)
    os_, op = o.tree()
    assert np.array_equal(gp, op)
    assert np.array_equal(bits(gs), bits(os_))
    assert np.array_equal(bits(gpu.path(p)), bits(o.path()))


# --------------------------------------------------------------------------- arithmetic
def test_device_f64_arithmetic_is_ieee():
    rng = np.random.default_rng(1)
    # sqrt: wide exponent range, subnormals, perfect squares +-1 ulp, the 3-doubles-share-a-root case
    mant = rng.random(200000) + 1.0
    expo = rng.integers(-1000, 1000, size=mant.size)
    x = np.ldexp(mant, expo)
    sq = rng.random(50000) * 100.0
    sq = sq * sq
    x = np.concatenate([x, sq, np.nextafter(sq, np.inf), np.nextafter(sq, 0.0),
                        [0.0, 5e-324, 2.2250738585072014e-308, 1.0, 1.0 + 2.0 ** -52, 1.0 + 2.0 ** -51, 4.0, np.inf,
                         300.0, 800.0, 200.0, 1.7976931348623157e308]])
    got = capi.f64_op_batch(0, x)
    assert np.array_equal(bits(got), bits(np.sqrt(x)))
    # division
    a = np.ldexp(rng.random(200000) + 1.0, rng.integers(-300, 300, size=200000)) * rng.choice([-1.0, 1.0], 200000)
    b = np.ldexp(rng.random(200000) + 1.0, rng.integers(-300, 300, size=200000))
    got = capi.f64_op_batch(1, a, b)
    assert np.array_equal(bits(got), bits(a / b))
    a2 = np.array([0.5, 0.5, 1.0, 6.0, 0.0, 1.0, 0.0])
    b2 = np.array([0.08660254037844389, 0.14142135623730953, 3.0, 6.0, 1.0, 0.0, 0.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        want = a2 / b2
    assert np.array_equal(bits(capi.f64_op_batch(1, a2, b2)), bits(want))
    # ceil
    c = np.concatenate([rng.random(10000) * 20.0 - 10.0, [5.773502691896258, 6.0, -0.0, 0.0, 1e300, 0.9999999999999999]])
    assert np.array_equal(bits(capi.f64_op_batch(2, c)), bits(np.ceil(c)))
    # interpolate must NOT be contracted into an FMA: from + (to - from) * t, three roundings
    f = rng.random(200000) * 20.0 - 10.0
    t = rng.random(200000) * 20.0 - 10.0
    tt = rng.random(200000)
    want = f + (t - f) * tt
    got = capi.f64_op_batch(3, f, t, tt)
    assert np.array_equal(bits(got), bits(want))
    # (a-b)^2
    got = capi.f64_op_batch(4, f, t)
    assert np.array_equal(bits(got), bits((f - t) * (f - t)))


def test_device_rng_stream_matches_oracle(golden):
    g = golden["rng"]
    got = capi.rng_u64_batch(g["seed"], g["stream"], 40)
    assert ["%016x" % int(v) for v in got] == g["u64"]
    for seed, stream, n in [(0, 0, 8), (42, 1023, 1000), (2 ** 64 - 1, 2 ** 63 + 5, 777)]:
        r = orc.Rng(seed, stream)
        want = np.array([r.next_u64() for _ in range(n)], dtype=np.uint64)
        assert np.array_equal(capi.rng_u64_batch(seed, stream, n), want)


def test_distance_interpolate_batch(golden):
    for dim in (1, 2, 3, 6):
        ks = [k for k in golden["space"]["kat"] if len(k["a"]) == dim]
        a = np.array([[unhex(v) for v in k["a"]] for k in ks])
        b = np.array([[unhex(v) for v in k["b"]] for k in ks])
        t = np.array([unhex(k["t"]) for k in ks])
        assert [hexf(v) for v in capi.distance_batch(a, b)] == [k["distance"] for k in ks]
        got = capi.interpolate_batch(a, b, t)
        assert [[hexf(v) for v in row] for row in got] == [k["interpolate"] for k in ks]
    rng = np.random.default_rng(5)
    for dim in range(1, 9):
        a = rng.random((4096, dim)) * 20 - 10
        b = rng.random((4096, dim)) * 20 - 10
        t = rng.random(4096)
        want = np.array([orc.distance(x, y) for x, y in zip(a, b)])
        assert np.array_equal(bits(capi.distance_batch(a, b)), bits(want))
        want = np.array([orc.interpolate(x, y, tt) for x, y, tt in zip(a, b, t)])
        assert np.array_equal(bits(capi.interpolate_batch(a, b, t)), bits(want))


# ------------------------------------------------------------------------ nearest neighbour
def test_nn_argmin_random_and_ragged():
    rng = np.random.default_rng(7)
    for dim in (1, 2, 3, 6, 8):
        sizes = [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 10000]
        trees = [rng.random((n, dim)) * 10.0 for n in sizes]
        qs = rng.random((len(sizes), dim)) * 10.0
        idx, md = capi.nn_argmin_batch(trees, qs)
        for t, q, i, d in zip(trees, qs, idx, md):
            wi, wd = orc.nearest(t, q)
            assert (int(i), hexf(d)) == (wi, hexf(wd))


def test_nn_argmin_ties_resolve_to_lowest_index():
    y = 2.0 ** -26  # d2 = 1 + 2^-52 and d2 = 1 share sqrt == 1.0: the reference keeps the LOWER index
    cases = [
        (np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [0.0, -1.0]]), [0.0, 0.0]),
        (np.array([[1.0, y], [1.0, 0.0]]), [0.0, 0.0]),
        (np.array([[1.0, 0.0], [1.0, y]]), [0.0, 0.0]),
        (np.array([[3.0, 3.0]] * 700), [1.0, 1.0]),  # all duplicates -> index 0
    ]
    # near-ties hidden at every position of a large tree (inside one lane, across lanes, across waves)
    rng = np.random.default_rng(11)
    base = rng.random((3000, 2)) * 10.0 + 5.0
    for pos_a, pos_b in [(0, 1), (5, 2999), (64, 128), (255, 256), (1234, 1234 + 256), (2999, 17)]:
        t = base.copy()
        t[pos_a] = [1.0, y]
        t[pos_b] = [1.0, 0.0]
        cases.append((t, [0.0, 0.0]))
    trees = [c[0] for c in cases]
    qs = np.array([c[1] for c in cases])
    idx, md = capi.nn_argmin_batch(trees, qs)
    for t, q, i, d in zip(trees, qs, idx, md):
        wi, wd = orc.nearest(t, q)
        assert (int(i), hexf(d)) == (wi, hexf(wd))
    assert int(idx[1]) == 0 and int(idx[2]) == 0 and int(idx[3]) == 0


# ------------------------------------------------------------------ validity / motion check
@pytest.mark.parametrize("scn", ["config1", "wall", "config2"])
def test_is_valid_and_check_motion_match_oracle(scn):
    sc = getattr(scenarios, scn)()
    gpu = _gpu_for(sc, 1, 100, True, 0)
    o = _oracle_for(sc, 0, 0, 100, True)
    rng = np.random.default_rng(3)
    lo = np.array([b[0] for b in sc["bounds"]])
    hi = np.array([b[1] for b in sc["bounds"]])
    pts = rng.random((4000, sc["dim"])) * (hi - lo) + lo
    if sc["spheres"] is not None:  # points exactly on / next to sphere surfaces
        c, r = sc["spheres"]
        on = c.copy()
        on[:, 0] += r
        pts = np.concatenate([pts, on, np.nextafter(on, np.inf), np.nextafter(on, -np.inf)])
    if sc["boxes"] is not None:    # box faces are inclusive
        blo, bhi = sc["boxes"]
        pts = np.concatenate([pts, blo, bhi, np.nextafter(blo, -np.inf), np.nextafter(bhi, np.inf)])
    want = np.array([o.is_valid(p) for p in pts])
    assert np.array_equal(gpu.is_valid(pts), want)
    assert 0 < want.sum() < len(want)
    a = rng.random((3000, sc["dim"])) * (hi - lo) + lo
    step = rng.random((3000, 1)) * 1.2
    d = rng.standard_normal((3000, sc["dim"]))
    b = a + d / np.linalg.norm(d, axis=1, keepdims=True) * step
    b[:100] = a[:100]  # from == to -> num_steps == 0 -> is_valid(to)
    want = np.array([o.check_motion(x, y) for x, y in zip(a, b)])
    assert np.array_equal(gpu.check_motion(a, b), want)
    assert 0 < want.sum() < len(want)
    gpu.close()


# --------------------------------------------------------------------------- planner parity
@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
@pytest.mark.parametrize("key", ["config1", "wall"])
def test_rrt_matches_golden_fixtures(golden, key, kernel):
    """README quick-start (config 1) and the reference's own wall test scene: identical node
    count, parents, state bits, path and checksum as the committed fixtures."""
    params = golden[key]["params"]
    sc = dict(dim=params["dim"], bounds=params["bounds"], max_distance=params["max_distance"],
              goal_bias=params["goal_bias"], lvs_fraction=params["fraction"], start=params["start"],
              goal_centre=params["goal_c"], goal_radius=params["goal_r"],
              spheres=params_spheres(params) if params["spheres"] else None,
              boxes=params_boxes(params) if params["boxes"] else None)
    # the fixtures finish far below 10,000 nodes, so the resident kernel's capacity gives the same trees
    max_nodes = params["max_nodes"] if kernel == capi.KERNEL_STREAM else 10000
    for run in golden[key]["runs"]:
        gpu = _gpu_for(sc, 1, max_nodes, True, run["seed"], run["pid"], kernel)
        st = gpu.solve(params["max_iterations"])
        assert st[0] == capi.OK
        c = gpu.counts()
        assert int(c["nodes"][0]) == run["n"]
        assert int(c["iterations"][0]) == run["iterations"]
        assert int(c["accepted"][0]) == run["accepted"]
        assert "%016x" % int(c["checksum"][0]) == run["checksum"]
        assert int(c["goal_node"][0]) == run["goal_node"]
        assert int(c["stop_reason"][0]) == capi.STOP_GOAL
        states, parents = gpu.tree(0)
        m = len(run["first_parents"])
        assert [[hexf(v) for v in row] for row in states[:m]] == run["first_states"]
        assert list(parents[:m]) == run["first_parents"]
        path = gpu.path(0)
        assert [[hexf(v) for v in row] for row in path] == run["path"]
        # the reference's own assertions (oxmpl/tests/rrt_rvss_tests.rs:168-185)
        o = _oracle_for(sc, 0, 0, 10, True)
        assert len(path) > 0
        assert orc.distance(path[0], sc["start"]) < 1e-9
        assert orc.distance(path[-1], sc["goal_centre"]) <= sc["goal_radius"]
        assert is_path_valid(path, sc["bounds"], sc["lvs_fraction"], o.is_valid, orc.maximum_extent,
                             orc.num_steps, orc.interpolate, orc.distance)
        gpu.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_rrt_config2_golden_and_oracle_batch(golden, kernel):
    """config 2 (R^3, 64 spheres): a 48-problem batch grown for 1500 iterations each equals the
    oracle problem by problem; problems 0, 1 also equal the committed numpy fixtures."""
    sc = scenarios.config2()
    c, r = params_spheres(golden["config2"]["params"])
    assert np.array_equal(bits(c), bits(sc["spheres"][0])) and np.array_equal(bits(r), bits(sc["spheres"][1]))
    P = 48
    gpu = _gpu_for(sc, P, 10000, False, 42, 0, kernel)
    st = gpu.solve(1500)
    cnt = gpu.counts()
    planners = [_oracle_for(sc, 42, p, 10000, False) for p in range(P)]
    orc.solve_many(planners, 1500, threads=8)
    for p in range(P):
        _assert_same_problem(gpu, p, planners[p], cnt)
        assert st[p] == (capi.OK if planners[p].goal_node >= 0 else capi.ERR_NO_SOLUTION_FOUND)
    for run in golden["config2"]["runs"]:
        if run["pid"] < P:
            p = run["pid"]
            assert "%016x" % int(cnt["checksum"][p]) == run["checksum"]
            states, parents = gpu.tree(p)
            m = len(run["first_parents"])
            assert [[hexf(v) for v in row] for row in states[:m]] == run["first_states"]
            assert [[hexf(v) for v in row] for row in gpu.path(p)] == run["path"]
    gpu.close()
    # sharding: problems 1000..1023 on their own batch equal the oracle's problem ids 1000..1023
    gpu = _gpu_for(sc, 24, 10000, False, 42, 1000, kernel)
    gpu.solve(600)
    planners = [_oracle_for(sc, 42, 1000 + p, 10000, False) for p in range(24)]
    orc.solve_many(planners, 600, threads=8)
    for p in range(24):
        _assert_same_problem(gpu, p, planners[p])
    gpu.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_LANES, capi.KERNEL_AUTO], ids=lambda k: KNAME[k])
@pytest.mark.parametrize("dim,max_nodes", [(3, 15000), (2, 20000), (2, 11500)])
def test_rrt_large_row_instantiations(kernel, dim, max_nodes):   # (AUTO: the cell-grid kernel, whatever the batch size)
    """trees beyond 11,264 nodes run the lane-per-query kernel's 32-row (R^3, up to 16,384 nodes) and 40-row (R^2, up to
    20,480) instantiations: grown to capacity, then frozen iterations at full size, against the oracle"""
    sc = scenarios.config2() if dim == 3 else scenarios.config1()
    P = 3
    gpu = _gpu_for(sc, P, max_nodes, False, 11, 40, kernel)
    gpu.solve(6000)             # resume across the instantiation's whole range
    gpu.solve(10 ** 7)
    gpu.solve(700, freeze=True)
    c = gpu.counts()
    assert (c["nodes"] == max_nodes).all()
    assert gpu.last_timing()["kernel"] == (capi.KERNEL_LANES if kernel == capi.KERNEL_LANES else capi.KERNEL_CELLS)
    planners = [_oracle_for(sc, 11, 40 + p, max_nodes, False) for p in range(P)]
    orc.solve_many(planners, 10 ** 7, threads=3)
    orc.solve_many(planners, 700, freeze=True, threads=3)
    for p in range(P):
        _assert_same_problem(gpu, p, planners[p], c)
    gpu.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_rrt_other_dimensions_and_obstacle_mixes(kernel):
    """dims 1..6, spheres+boxes mixed, goal_bias 0 / 1 / 0.5, more obstacles than one wave."""
    rng = np.random.default_rng(21)
    for dim, gb, ns, nb in [(1, 0.05, 0, 0), (2, 0.5, 3, 2), (3, 1.0, 70, 0), (3, 0.0, 130, 5), (4, 0.1, 10, 1),
                            (5, 0.3, 40, 2), (6, 0.05, 20, 0)]:
        bounds = [(-3.0, 7.0)] * dim
        sc = dict(dim=dim, bounds=bounds, max_distance=0.7, goal_bias=gb, lvs_fraction=0.02,
                  start=[-2.5] * dim, goal_centre=[6.5] * dim, goal_radius=0.4,
                  spheres=(rng.random((ns, dim)) * 6.0 - 1.0, rng.random(ns) * 0.5 + 0.1) if ns else None,
                  boxes=None)
        if nb:
            lo = rng.random((nb, dim)) * 6.0 - 1.0
            sc["boxes"] = (lo, lo + rng.random((nb, dim)) * 0.8 + 0.1)
        P = 6
        if kernel == capi.KERNEL_RESIDENT and dim not in (2, 3):
            continue  # these resident kernels are instantiated for R^2 / R^3 only
        if kernel == capi.KERNEL_LANES and dim not in (2, 3, 4, 5, 6):
            continue  # the lane-per-query kernel: R^2 .. R^6
        if kernel == capi.KERNEL_CELLS and dim not in (2, 3):
            continue  # the cell-grid kernel: R^2, R^3
        gpu = _gpu_for(sc, P, 400, False, 7, 100, kernel)
        gpu.solve(500)
        planners = [_oracle_for(sc, 7, 100 + p, 400, False) for p in range(P)]
        orc.solve_many(planners, 500, threads=6)
        for p in range(P):
            _assert_same_problem(gpu, p, planners[p])
        gpu.close()


@pytest.mark.parametrize("mode", ["product", "whole_tree_path_forced", "whole_tree_path_forced_diagnostic_build"])
@pytest.mark.parametrize("kernel", [capi.KERNEL_LANES, capi.KERNEL_STREAM], ids=lambda k: KNAME[k])
def test_fuzz_seed203_r6_case(kernel, mode):
    """The one mismatch tools/fuzz_parity.py found this round (seed 203, committed as a fixture with the obstacle field it drew):
    R^6, 100 spheres, 9 problems grown to 9000 nodes.  In problem 2, iteration 5868, two nodes are 4.5 apart in d2 at
    E = 2.4: the lane-per-query kernel of that build sent the query to its whole-tree path, whose wave-wide motion check read
    sphere 5's threshold across lanes from inside a divergent region -- with the register spilled (R^4..R^6), lane 5 held a
    stale temporary and an end state inside the sphere was accepted.  debug_flags = DEBUG_PAIR_TO_WHOLE_TREE sends two-lane cases down that
    path again (the product build resolves them in the round), so the path stays covered -- in the product instantiation
    and in the diagnostic one (stamps on), whose audit counts accepted end states that lie inside a sphere."""
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_r6_seed203_case.npz"))
    d = json.loads(str(z["desc"]))
    no_pair, stamped = mode != "product", mode.endswith("diagnostic_build")
    if no_pair:
        if kernel != capi.KERNEL_LANES:
            pytest.skip("the switch concerns the lane-per-query kernel only")
    flags = (capi.DEBUG_PAIR_TO_WHOLE_TREE | (capi.DEBUG_AUDIT if stamped else 0)) if no_pair else 0   # AUDIT: the diagnostic build's end-state audit
    dim = d["dim"]
    bounds = [(d["lo"], d["hi"])] * dim
    gpu = capi.RRTBatch(dim, bounds, d["md"], d["gb"], d["nprob"], d["max_nodes"], d["frac"], d["stop"], d["seed"], d["pid0"], 0, kernel,
                        debug_flags=flags)
    gpu.set_spheres(z["sc"], z["sr"])
    gpu.setup(z["start"], z["goal"], float(z["gr"]))
    if stamped:
        gpu.enable_stamps(True)
    for a in z["schedule"]:
        gpu.solve(int(a))
    planners = []
    for p in range(d["nprob"]):
        o = orc.OracleRRT(dim, bounds, d["md"], d["gb"], d["frac"], d["max_nodes"], d["stop"], d["seed"], d["pid0"] + p)
        o.set_spheres(z["sc"], z["sr"])
        o.setup(z["start"], z["goal"], float(z["gr"]))
        planners.append(o)
    orc.solve_many(planners, int(sum(z["schedule"])), threads=9)
    c = gpu.counts()
    for p in range(d["nprob"]):
        _assert_same_problem(gpu, p, planners[p], c)
    if stamped:
        assert int(gpu.stamps()[50]) == 0          # the diagnostic build's audit: no accepted end state inside a sphere
        assert int(gpu.stamps()[45]) > 0           # summed over the nine workgroups: the whole-tree path was taken (the case sits in problem 2)
    gpu.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_long_motion_checks_are_not_a_stalled_pipeline(kernel):
    """A tiny longest-valid-segment fraction makes every motion check 300,000 interpolated states against 48 boxes
    (boxes are stepped, never filtered): the resident kernels' resolver then works for longer than the watchdog of the
    waves waiting on it, which must read its heartbeat instead of reporting a stalled hand-off."""
    if kernel == capi.KERNEL_AUTO:
        pytest.skip("a batch of three runs the lane-per-query kernel under KERNEL_AUTO: the [lanes] case")
    # (the cell-grid kernel has no hand-off to stall -- one wave does everything -- and steps every lane's motion in lockstep: a third
    #  of the states keeps its case to the time of the others)
    states = 1.0e5 if kernel == capi.KERNEL_CELLS else 3.0e5
    rng = np.random.default_rng(5)
    lo = np.column_stack([rng.uniform(0.5, 9.0, 48), rng.uniform(6.0, 9.5, 48)])
    sc = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], max_distance=0.5, goal_bias=0.05,
              lvs_fraction=0.5 / (states * 0.1 * math.sqrt(200.0)), start=[1.0, 1.0], goal_centre=[9.0, 1.0], goal_radius=0.3,
              spheres=None, boxes=(lo, lo + 0.2))
    P, iters = 3, 24
    gpu = _gpu_for(sc, P, 64, False, 3, 0, kernel)
    gpu.solve(iters)
    planners = [_oracle_for(sc, 3, p, 64, False) for p in range(P)]
    orc.solve_many(planners, iters, threads=3)
    for p in range(P):
        _assert_same_problem(gpu, p, planners[p])
    gpu.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_rrt_termination_resume_and_freeze(kernel):
    sc = scenarios.config2()
    # node cap: stops at max_nodes without drawing further
    gpu = _gpu_for(sc, 4, 300, False, 5, 0, kernel)
    gpu.solve(10 ** 6)
    c = gpu.counts()
    planners = [_oracle_for(sc, 5, p, 300, False) for p in range(4)]
    for o in planners:
        o.solve(10 ** 6)
    for p in range(4):
        assert int(c["nodes"][p]) == 300 and int(c["stop_reason"][p]) == capi.STOP_NODES
        _assert_same_problem(gpu, p, planners[p], c)
    it0 = c["iterations"].copy()
    gpu.solve(100)
    assert np.array_equal(gpu.counts()["iterations"], it0)
    gpu.close()
    # resume: 300 + 500 iterations == 800 iterations; then 256 frozen iterations
    gpu = _gpu_for(sc, 4, 10000, False, 9, 0, kernel)
    gpu.solve(300)
    gpu.solve(500)
    planners = [_oracle_for(sc, 9, p, 10000, False) for p in range(4)]
    for o in planners:
        o.solve(800)
    for p in range(4):
        _assert_same_problem(gpu, p, planners[p])
    n0 = gpu.counts()["nodes"].copy()
    gpu.solve(256, freeze=True)
    for o in planners:
        o.solve(256, freeze=True)
    c = gpu.counts()
    assert np.array_equal(c["nodes"], n0)
    for p in range(4):
        _assert_same_problem(gpu, p, planners[p], c)
    gpu.close()
    # stop_at_goal: solved problems are left alone by a second solve (idempotent)
    gpu = _gpu_for(sc, 8, 10000, True, 42, 0, kernel)
    st = gpu.solve(100000)
    assert (st == capi.OK).all()
    c1 = gpu.counts()
    assert (c1["stop_reason"] == capi.STOP_GOAL).all()
    gpu.solve(1000)
    c2 = gpu.counts()
    assert np.array_equal(c1["iterations"], c2["iterations"]) and np.array_equal(c1["nodes"], c2["nodes"])
    planners = [_oracle_for(sc, 42, p, 10000, True) for p in range(8)]
    for p, o in enumerate(planners):
        assert o.solve(100000) == orc.SOLVED
        _assert_same_problem(gpu, p, o, c2)
    gpu.close()
    # timeout: every motion is invalid (one huge sphere), so the tree never grows and the
    # iteration budget cannot be reached -> wall-clock Timeout (rrt.rs:172-174)
    blocked = dict(sc)
    blocked["spheres"] = (np.array([[5.0, 5.0, 5.0]]), np.array([100.0]))
    gpu = _gpu_for(blocked, 2, 1000, True, 1, 0, kernel)
    st = gpu.solve(10 ** 12, timeout_s=0.05)
    assert (st == capi.ERR_TIMEOUT).all()
    c = gpu.counts()
    assert (c["stop_reason"] == capi.STOP_TIMEOUT).all() and (c["nodes"] == 1).all() and (c["accepted"] == 0).all()
    gpu.close()


def _d2(a, q):
    acc = None
    for x, y in zip(a, q):
        d = x - y
        acc = d * d if acc is None else acc + d * d
    return acc


def _near_tie_pair(q, rng):
    """two states whose squared distances to q differ (by an ulp or two) but whose square roots
    round to the same double: the reference (post-sqrt compare) must keep the LOWER index"""
    for _ in range(200000):
        v = rng.standard_normal(len(q))
        a = [float(x + 0.9 * w / np.linalg.norm(v)) for x, w in zip(q, v)]
        da = _d2(a, q)
        b = list(a)
        for _k in range(40):
            b[-1] = float(np.nextafter(b[-1], np.inf))
            db = _d2(b, q)
            if db != da and math.sqrt(db) == math.sqrt(da):
                return a, b, da, db
    raise AssertionError("no near-tie pair found")


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_planner_near_ties_take_the_exact_path(kernel):
    """Plant, in each problem's tree, two nodes whose d2 to the problem's FIRST query differ but whose
    sqrt coincide, at index pairs that land in one lane, in different lanes and in different waves
    of both kernels; the reference's strict '<' on the post-sqrt value keeps the lower index even
    when its d2 is the larger one.  One frozen iteration; nearest index is part of the checksum."""
    sc = scenarios.config2()
    sc["goal_bias"] = 0.0
    pairs = [(1, 2), (5, 517), (5, 261), (70, 700), (2999, 17), (1030, 6), (64, 128), (511, 512), (2047, 2048)]
    cases = [(ia, ib, swap) for ia, ib in pairs for swap in (False, True)]
    P, n = len(cases), 3000
    gpu = _gpu_for(sc, P, 4096, False, 77, 0, kernel)
    rng = np.random.default_rng(123)
    planners, flipped = [], 0
    for p, (ia, ib, swap) in enumerate(cases):
        r = orc.Rng(77, p)
        assert not r.random_bool(0.0)
        q = [r.random_range(lo, hi) for lo, hi in sc["bounds"]]
        a, b, da, db = _near_tie_pair(q, rng)
        if swap:
            a, b, da, db = b, a, db, da
        # every other node is at least 3 away from q
        far = rng.standard_normal((n, 3))
        far = np.array(q) + far / np.linalg.norm(far, axis=1, keepdims=True) * (3.0 + rng.random((n, 1)) * 4.0)
        tree = far.copy()
        tree[ia], tree[ib] = a, b
        parents = np.concatenate([[-1], rng.integers(0, np.arange(1, n))]).astype(np.int32)
        o = _oracle_for(sc, 77, p, 4096, False)
        assert o.set_tree(tree, parents) == 0
        gpu.set_tree(p, tree, parents)
        planners.append(o)
        lo_idx = min(ia, ib)
        d_lo = da if ia < ib else db
        d_hi = db if ia < ib else da
        if d_lo > d_hi:
            flipped += 1  # a d2-argmin would pick the other node
        assert orc.nearest(tree, q)[0] == lo_idx
    assert flipped >= 3
    gpu.solve(1, freeze=True)
    c = gpu.counts()
    for p, o in enumerate(planners):
        o.solve(1, freeze=True)
        assert int(c["checksum"][p]) == o.checksum, cases[p]
        assert int(c["nodes"][p]) == n
    # and the warm-started trees keep growing identically
    gpu.solve(300)
    for p, o in enumerate(planners):
        o.solve(300)
        _assert_same_problem(gpu, p, o)
    gpu.close()


MARGIN_EPS = [0.0, 2.0 ** -40, 2.0 ** -30, 2.0 ** -26, 2.0 ** -24, 2.0 ** -23, 2.0 ** -22, 2.0 ** -21, 2.0 ** -20, 2.0 ** -19,
              2.0 ** -18, 2.0 ** -17, 2.0 ** -16, 2.0 ** -15, 2.0 ** -14, 2.0 ** -12, 2.0 ** -10, 2.0 ** -6]


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_screen_margin_sweep(kernel):
    """Directed test of the binary32 screens' accept / reject margin.  The query is fixed (goal_bias = 1: every query is
    the goal centre, no draws); the tree holds a nearest node A at distance d and a planted runner-up B at d (1 + eps), eps
    swept from 0 through the screen's margin (~2^-18 .. 2^-15 of d) to far beyond it, with A and B placed in one lane,
    in neighbouring lanes, in different waves and in different register rows, B before A and after A.  Whatever the
    screen decides, nearest index, q_new and verdict (the checksum) must equal the oracle's; and for the register-resident
    binary32-screen kernel the in-kernel counter of exact-path events shows that the screen refuses to decide on the near side
    of its margin and does decide on the far side (so the fast path is what the other tests exercise)."""
    sc = scenarios.config2()
    sc["goal_bias"] = 1.0
    q = np.array(sc["goal_centre"])
    pairs = [(1, 2), (2, 1), (5, 517), (517, 5), (5, 69), (70, 700), (2999, 17), (1030, 6), (511, 512), (2047, 2048), (2900, 2901)]
    n = 3000
    rng = np.random.default_rng(2024)
    amb_by_eps = {}
    for eps in MARGIN_EPS:
        P = len(pairs)
        gpu = _gpu_for(sc, P, 4096, False, 5, 0, kernel)
        planners = []
        for p, (ia, ib) in enumerate(pairs):
            u = rng.standard_normal((2, 3))
            u /= np.linalg.norm(u, axis=1, keepdims=True)
            d = 0.4 + 0.5 * rng.random()
            a = q - np.abs(u[0]) * d                      # inside the bounds: the goal centre sits near the upper corner
            b = q - np.abs(u[1]) * d * (1.0 + eps)
            far = rng.standard_normal((n, 3))
            far = q - np.abs(far / np.linalg.norm(far, axis=1, keepdims=True)) * (3.0 + rng.random((n, 1)) * 4.0)
            tree = far.copy()
            tree[ia], tree[ib] = a, b
            parents = np.concatenate([[-1], rng.integers(0, np.arange(1, n))]).astype(np.int32)
            o = _oracle_for(sc, 5, p, 4096, False)
            assert o.set_tree(tree, parents) == 0
            gpu.set_tree(p, tree, parents)
            planners.append(o)
        if kernel in SCREENED:
            gpu.enable_stamps(True)
        gpu.solve(8, freeze=True)
        c = gpu.counts()
        for p, o in enumerate(planners):
            o.solve(8, freeze=True)
            assert int(c["checksum"][p]) == o.checksum, (eps, pairs[p])
            assert int(c["iterations"][p]) == o.iterations == 8
        if kernel in SCREENED:
            amb_by_eps[eps] = int(gpu.stamps()[4])
        # the planted trees keep growing identically (inserts on, same fixed query: duplicates and near-ties galore)
        gpu.enable_stamps(False) if kernel in SCREENED else None
        gpu.solve(40)
        for p, o in enumerate(planners):
            o.solve(40)
            _assert_same_problem(gpu, p, o)
        gpu.close()
    if kernel in SCREENED:
        for eps, n_amb in amb_by_eps.items():
            if eps <= 2.0 ** -22:
                assert n_amb > 0, ("the screen decided a pair it cannot separate", eps, amb_by_eps)
            if eps >= 2.0 ** -10:
                assert n_amb == 0, ("the screen never takes its fast path", eps, amb_by_eps)


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
@pytest.mark.parametrize("scale,offset,max_nodes", [(1.0, 1.0e3, 600), (1.0, 1.0e6, 600), (1.0, -5.0e4, 3000),
                                                    (1.0e-12, 0.0, 600), (1.0e18, 0.0, 600), (1.0e30, 0.0, 3000),
                                                    (1.0e100, 0.0, 600)])
def test_rrt_translated_and_scaled_spaces(kernel, scale, offset, max_nodes):
    """The same scene moved far from the origin or blown up / shrunk by many orders of magnitude.  For the
    binary32-screen kernel these are the regimes of its error model: a large offset makes binary32 too coarse to
    separate any two nodes (every query takes the exact binary64 path), 1e18 and beyond exceed the range in which
    binary32 squares are trusted at all (1e30: they overflow), 1e-12 exercises the subnormal end.  Results must not
    move by a bit in any of them, for any kernel."""
    base = scenarios.config2()
    t = lambda v: (np.asarray(v, dtype=np.float64) * scale + offset)
    c, r = base["spheres"]
    sc = dict(dim=3, bounds=[(float(t(lo)), float(t(hi))) for lo, hi in base["bounds"]],
              max_distance=base["max_distance"] * scale, goal_bias=0.1, lvs_fraction=base["lvs_fraction"],
              start=list(t(base["start"])), goal_centre=list(t(base["goal_centre"])), goal_radius=base["goal_radius"] * scale,
              spheres=(t(c), np.asarray(r) * scale), boxes=None)
    P, iters = 4, 700
    gpu = _gpu_for(sc, P, max_nodes, False, 31, 40, kernel)
    gpu.solve(iters)
    planners = [_oracle_for(sc, 31, 40 + p, max_nodes, False) for p in range(P)]
    orc.solve_many(planners, iters, threads=4)
    for p in range(P):
        _assert_same_problem(gpu, p, planners[p])
        gs, gp = gpu.tree(p)
        os_, op = planners[p].tree()
        assert np.array_equal(gp, op) and np.array_equal(gs.view(np.uint64), os_.view(np.uint64))
    assert min(o.num_nodes for o in planners) > 100   # the scene still lets the trees grow
    gpu.close()


def test_solve_before_setup_is_planner_uninitialised():
    b = capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.1, 0.0, 1, 10)
    with pytest.raises(capi.OxhipError) as ei:
        b.solve(10)
    assert ei.value.status == capi.ERR_PLANNER_UNINITIALISED  # rrt.rs:160-163
    b.close()


FULL_P, FULL_N, FULL_SAMPLE = 1024, 10000, [0, 1, 255, 256, 511, 777, 1022, 1023]


@pytest.fixture(scope="module")
def full_size_oracle():
    """BASELINE.json configs[1] on the CPU oracle, ALL 1024 problems: counters after the grow phase (1 -> 10,000 nodes)
    and after 64 further frozen iterations; the planners of FULL_SAMPLE are kept for tree / path comparison.  One pass
    serves every kernel kind (chunks of 128 planners bound the memory; ~20 s on the GPU box's 16 threads)."""
    sc = scenarios.config2()
    threads = min(os.cpu_count() or 1, 16)
    keys = ("nodes", "iterations", "accepted", "checksum", "goal_node")
    grown = {k: np.zeros(FULL_P, dtype=np.int64 if k == "goal_node" else np.uint64) for k in keys}
    frozen = {k: np.zeros(FULL_P, dtype=np.int64 if k == "goal_node" else np.uint64) for k in keys}
    kept = {}

    def record(dst, p, pl):
        dst["nodes"][p], dst["iterations"][p], dst["accepted"][p] = pl.num_nodes, pl.iterations, pl.accepted
        dst["checksum"][p], dst["goal_node"][p] = pl.checksum, pl.goal_node

    for p0 in range(0, FULL_P, 128):
        ids = list(range(p0, min(FULL_P, p0 + 128)))
        planners = [_oracle_for(sc, 42, p, FULL_N, False) for p in ids]
        orc.solve_many(planners, 10 ** 7, threads=threads)
        for p, pl in zip(ids, planners):
            record(grown, p, pl)
            if p in FULL_SAMPLE:
                kept[p] = (pl.tree(), pl.path())
        orc.solve_many(planners, 64, freeze=True, threads=threads)
        for p, pl in zip(ids, planners):
            record(frozen, p, pl)
        del planners
    return grown, frozen, kept


@pytest.mark.parametrize("kernel", KERNELS, ids=lambda k: KNAME[k])
def test_full_size_config2_properties(kernel, full_size_oracle):
    """BASELINE.json configs[1] at full size (1024 problems x 10,000 nodes): EVERY problem's node count, iteration count,
    accepted count, goal node and per-iteration checksum equal the oracle's after the grow phase and after 64 frozen
    iterations at n = 10,000; trees / paths bit for bit on a sample; size-independent properties on a stride."""
    sc = scenarios.config2()
    P, N = FULL_P, FULL_N
    grown, frozen, kept = full_size_oracle
    gpu = _gpu_for(sc, P, N, False, 42, 0, kernel)
    gpu.solve(10 ** 7)
    c = gpu.counts()
    assert (c["nodes"] == N).all() and (c["stop_reason"] == capi.STOP_NODES).all()
    assert (c["accepted"] == N - 1).all()
    assert (c["goal_node"] > 0).all()
    assert len(set(int(v) for v in c["checksum"])) == P  # independent streams
    for k in ("nodes", "iterations", "accepted", "checksum"):
        assert np.array_equal(c[k].astype(np.uint64), grown[k]), k
    assert np.array_equal(c["goal_node"].astype(np.int64), grown["goal_node"])
    o = _oracle_for(sc, 0, 0, 10, True)
    for p in range(0, P, 37):
        states, parents = gpu.tree(p)
        assert parents[0] == -1 and (parents[1:] >= 0).all() and (parents[1:] < np.arange(1, N)).all()
        edge = np.sqrt(((states[1:] - states[parents[1:]]) ** 2).sum(axis=1))
        assert edge.max() <= 0.5 * (1 + 1e-12)  # steer never exceeds max_distance
        assert (states >= 0.0).all() and (states <= 10.0).all()
        assert gpu.is_valid(states[1:]).all()   # every inserted state passed the motion check
        path = gpu.path(p)
        assert np.array_equal(bits(path[0]), bits(np.array(sc["start"])))
        assert orc.distance(path[-1], sc["goal_centre"]) <= sc["goal_radius"]
        assert is_path_valid(path, sc["bounds"], sc["lvs_fraction"], o.is_valid, orc.maximum_extent,
                             orc.num_steps, orc.interpolate, orc.distance)
    for p in FULL_SAMPLE:
        (os_, op), opath = kept[p]
        gs, gp = gpu.tree(p)
        assert np.array_equal(gp, op) and np.array_equal(bits(gs), bits(os_))
        assert np.array_equal(bits(gpu.path(p)), bits(opath))
    # steady mode at n = 10,000: 64 frozen iterations, every problem
    gpu.solve(64, freeze=True)
    c = gpu.counts()
    for k in ("nodes", "iterations", "accepted", "checksum"):
        assert np.array_equal(c[k].astype(np.uint64), frozen[k]), k
    gpu.close()


# ------------------------------------------------------------------------------ RRTConnect
def _oracle_connect(sc, seed, pid, max_nodes):
    p = orc.OracleRRTConnect(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                             max_nodes, seed, pid)
    if sc["spheres"] is not None:
        p.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        p.set_boxes(*sc["boxes"])
    p.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return p


def _assert_same_connect(gpu, p, o, c, gc):
    assert int(c["nodes"][p]) == o.num_nodes(0) and int(gc["nodes"][p]) == o.num_nodes(1)
    assert int(c["iterations"][p]) == o.iterations and int(c["checksum"][p]) == o.checksum
    assert int(c["goal_node"][p]) == o.end_node(0) and int(gc["end_node"][p]) == o.end_node(1)
    for which, (gs, gp) in enumerate((gpu.tree(p), gpu.goal_tree(p))):
        os_, op = o.tree(which)
        assert np.array_equal(gp, op) and np.array_equal(bits(gs), bits(os_))
    assert np.array_equal(bits(gpu.path(p)), bits(o.path()))


@pytest.mark.parametrize("key", ["connect_config1", "connect_wall"])
def test_rrt_connect_matches_golden_fixtures(golden, key):
    """RRTConnect (rrt_connect.rs) on the README scene and on the reference's own test scene
    (oxmpl/tests/rrt_connect_rvss_tests.rs, RRTConnect::new(0.5, 0.0)): both trees, merged path, checksum"""
    params = golden[key]["params"]
    sc = dict(dim=params["dim"], bounds=params["bounds"], max_distance=params["max_distance"],
              goal_bias=params["goal_bias"], lvs_fraction=params["fraction"], start=params["start"],
              goal_centre=params["goal_c"], goal_radius=params["goal_r"],
              spheres=params_spheres(params) if params["spheres"] else None,
              boxes=params_boxes(params) if params["boxes"] else None)
    ov = _oracle_for(sc, 0, 0, 10, True)
    for run in golden[key]["runs"]:
        gpu = scenarios.make_batch(sc, 1, params["max_nodes"], True, run["seed"], run["pid"], 0, 0, capi.PLANNER_RRT_CONNECT)
        st = gpu.solve(params["max_iterations"])
        assert st[0] == capi.OK
        c, gc = gpu.counts(), gpu.goal_counts()
        assert [int(c["nodes"][0]), int(gc["nodes"][0])] == run["n"]
        assert int(c["iterations"][0]) == run["iterations"] and "%016x" % int(c["checksum"][0]) == run["checksum"]
        assert [int(c["goal_node"][0]), int(gc["end_node"][0])] == run["end"]
        assert int(c["stop_reason"][0]) == capi.STOP_GOAL
        for which, (states, parents) in enumerate((gpu.tree(0), gpu.goal_tree(0))):
            m = len(run["parents"][which])
            assert [[hexf(v) for v in row] for row in states[:m]] == run["states"][which]
            assert list(parents[:m]) == run["parents"][which]
        path = gpu.path(0)
        assert [[hexf(v) for v in row] for row in path] == run["path"]
        # the reference's assertions (rrt_connect_rvss_tests.rs:154-181)
        assert len(path) > 0 and orc.distance(path[0], sc["start"]) < 1e-9
        assert orc.distance(path[-1], sc["goal_centre"]) <= sc["goal_radius"]
        assert is_path_valid(path, sc["bounds"], sc["lvs_fraction"], ov.is_valid, orc.maximum_extent,
                             orc.num_steps, orc.interpolate, orc.distance)
        it = int(c["iterations"][0])
        gpu.solve(100)  # solved: a second solve is a no-op
        assert int(gpu.counts()["iterations"][0]) == it
        gpu.close()


def test_rrt_connect_batch_matches_oracle():
    """config 2 (R^3, 64 spheres), 64 problems: every problem equals the oracle; then caps, budgets, resume"""
    sc = scenarios.config2()
    P = 64
    gpu = scenarios.make_batch(sc, P, 10000, True, 42, 500, 0, 0, capi.PLANNER_RRT_CONNECT)
    st = gpu.solve(100000)
    assert (st == capi.OK).all()
    c, gc = gpu.counts(), gpu.goal_counts()
    planners = [_oracle_connect(sc, 42, 500 + p, 10000) for p in range(P)]
    for p, o in enumerate(planners):
        assert o.solve(100000) == orc.SOLVED
        _assert_same_connect(gpu, p, o, c, gc)
    # the same batch without the sphere lookup grid of the motion check (every state against every sphere): identical counters
    plain = scenarios.make_batch(sc, P, 10000, True, 42, 500, 0, 0, capi.PLANNER_RRT_CONNECT, debug_flags=capi.DEBUG_SE2_NO_SEGMENT_GRID)
    assert (plain.solve(100000) == capi.OK).all()
    cp, gcp = plain.counts(), plain.goal_counts()
    for k in ("nodes", "iterations", "checksum", "goal_node"):
        assert np.array_equal(c[k], cp[k]), k
    assert np.array_equal(gc["nodes"], gcp["nodes"])
    plain.close()
    gpu.close()
    # a 5-D problem with boxes + spheres on the runtime-dim kernel, goal_bias 0.3
    rng = np.random.default_rng(4)
    sc5 = dict(dim=5, bounds=[(-2.0, 6.0)] * 5, max_distance=0.9, goal_bias=0.3, lvs_fraction=0.03,
               start=[-1.5] * 5, goal_centre=[5.5] * 5, goal_radius=0.3,
               spheres=(rng.random((12, 5)) * 5.0, rng.random(12) * 0.6 + 0.2),
               boxes=(np.array([[1.0] * 5]), np.array([[1.8] * 5])))
    gpu = scenarios.make_batch(sc5, 6, 5000, True, 3, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
    gpu.solve(150)      # budget, then resume: 150 + 100000 == one long run
    gpu.solve(100000)
    c, gc = gpu.counts(), gpu.goal_counts()
    for p in range(6):
        o = _oracle_connect(sc5, 3, p, 5000)
        o.solve(100150)
        _assert_same_connect(gpu, p, o, c, gc)
    gpu.close()
    # node cap: a slab separates start from goal, both trees grow but can never connect
    blocked = dict(sc)
    blocked["spheres"] = None
    blocked["boxes"] = (np.array([[4.5, -1.0, -1.0]]), np.array([[5.5, 11.0, 11.0]]))
    gpu = scenarios.make_batch(blocked, 2, 300, True, 1, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
    st = gpu.solve(10 ** 6)
    c, gc = gpu.counts(), gpu.goal_counts()
    assert (st == capi.ERR_NO_SOLUTION_FOUND).all() and (c["stop_reason"] == capi.STOP_NODES).all()
    assert ((c["nodes"] == 300) | (gc["nodes"] == 300)).all()
    for p in range(2):
        o = _oracle_connect(blocked, 1, p, 300)
        assert o.solve(10 ** 6) == orc.NO_SOLUTION_FOUND
        _assert_same_connect(gpu, p, o, c, gc)
    with pytest.raises(capi.OxhipError):
        gpu.solve(10, freeze=True)
    gpu.close()


def test_rrt_connect_rarely_taken_paths_of_the_one_wave_kernel():
    """rrt_connect.hip (round 3: one wave per problem): trees beyond the LDS mirror (490 nodes in R^3: the scan continues in HBM
    behind a drain of this wave's own stores), an obstacle table too large for LDS (generic loads), more than 64 obstacles with
    boxes among them, motions of more than eight and of more than 64 states (the walked and the lane-per-state motion checks),
    a solve cut into launches that end inside a sampled block, and R^7 (32 instead of 64 iterations sampled at a time)"""
    rng = np.random.default_rng(77)
    # (a) an enclosed goal: both trees run into a node cap of 1,500; 20 spheres + the box walls
    wall = 0.2
    a, b = 7.0 - wall, 10.0 + wall   # six slabs around the cube [7, 10]^3, overlapping along its edges (no gap to step through)
    lo = np.array([[a, a, a], [a, a, 10.0], [a, a, a], [a, 10.0, a], [a, a, a], [10.0, a, a]])
    hi = np.array([[b, b, 7.0], [b, b, b], [b, 7.0, b], [b, b, b], [7.0, b, b], [b, b, b]])
    sca = dict(dim=3, bounds=[(0.0, 10.5)] * 3, max_distance=0.5, goal_bias=0.05, lvs_fraction=0.05, start=[0.5] * 3, goal_centre=[8.5] * 3,
               goal_radius=0.3, spheres=(rng.uniform(1.0, 6.0, (20, 3)), rng.uniform(0.2, 0.6, 20)), boxes=(lo, hi))
    g = scenarios.make_batch(sca, 2, 1500, True, 9, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
    st = g.solve(10 ** 6)
    c, gc = g.counts(), g.goal_counts()
    assert (st == capi.ERR_NO_SOLUTION_FOUND).all() and (c["stop_reason"] == capi.STOP_NODES).all()
    assert int(max(c["nodes"].max(), gc["nodes"].max())) == 1500
    for p in range(2):
        o = _oracle_connect(sca, 9, p, 1500)
        assert o.solve(10 ** 6) == orc.NO_SOLUTION_FOUND
        _assert_same_connect(g, p, o, c, gc)
    g.close()
    # (b) 300 small spheres + 40 boxes (1,440 doubles: the table stays in HBM), (c) 70 spheres + 30 boxes (two batches of 64 obstacles),
    # (d) a fine resolution: 12 states per motion, (e) a very fine one: 1,300 states per motion, (f) R^7
    def field(ns, nb, dim, r):
        cen = rng.uniform(1.5, 8.5, (ns, dim))
        blo = rng.uniform(1.5, 8.0, (nb, dim))
        return (cen, rng.uniform(0.5 * r, r, ns)), (blo, blo + rng.uniform(0.1, 0.5, (nb, dim)))
    for ns, nb, dim, r, frac, md, budget in ((300, 40, 3, 0.25, 0.05, 0.5, 4000), (70, 30, 3, 0.5, 0.05, 0.5, 4000), (64, 0, 3, 0.6, 0.02, 0.4, 3000),
                                             (5, 2, 2, 0.8, 0.0002, 0.37, 600), (40, 10, 7, 1.5, 0.05, 1.1, 3000)):
        sph, box = field(ns, nb, dim, r)
        sc = dict(dim=dim, bounds=[(0.0, 10.0)] * dim, max_distance=md, goal_bias=0.1, lvs_fraction=frac, start=[0.7] * dim,
                  goal_centre=[9.3] * dim, goal_radius=0.4, spheres=sph, boxes=box if nb else None)
        g = scenarios.make_batch(sc, 3, 3000, True, 5, 100, 0, 0, capi.PLANNER_RRT_CONNECT)
        done = 0
        for cut in (37, 101, budget):          # launches ending inside a sampled block
            g.solve(cut - done)
            done = cut
        c, gc = g.counts(), g.goal_counts()
        for p in range(3):
            o = _oracle_connect(sc, 5, 100 + p, 3000)
            o.solve(budget)
            _assert_same_connect(g, p, o, c, gc)
        g.close()
