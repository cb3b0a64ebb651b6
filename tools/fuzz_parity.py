#!/usr/bin/env python3
"""Randomised parity sweep: random spaces, obstacle fields, planner parameters and seeds, every planner / kernel against its
CPU oracle, bit for bit.  Usage: fuzz_parity.py [seconds] [seed] [planner]
Prints one line per failure (with the parameters to reproduce it) and a summary; exit code 1 on any failure.
tests/test_gpu_fuzz.py imports `sweep` and runs a bounded leg of it with fixed seeds inside the -m gpu suite."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

rng = np.random.default_rng(1)   # re-seeded by sweep()
DUMP_DIR = os.environ.get("FUZZ_DUMP_DIR", "")
# the lane-per-query kernel's rarely taken paths, forced one or two at a time (oxhip_debug_flag; results identical by construction)
LANE_FLAGS = [0, 0, 0, 0, capi.DEBUG_PAIR_TO_WHOLE_TREE, capi.DEBUG_ALL_WHOLE_TREE, capi.DEBUG_ONE_LANE_ROUNDS,
              capi.DEBUG_ALL_WHOLE_TREE | capi.DEBUG_SHORT_MEMO, capi.DEBUG_ONE_LANE_ROUNDS | capi.DEBUG_PAIR_TO_WHOLE_TREE]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def random_field(dim, lo, hi):
    ns, nb = int(rng.choice([0, 1, 5, 40, 64, 100])), int(rng.choice([0, 0, 1, 3]))
    w = hi - lo
    sc = rng.uniform(lo, hi, size=(ns, dim))
    sr = rng.uniform(0.02, 0.12, size=ns) * w * (1.0 if dim <= 3 else 2.5)
    if ns and rng.random() < 0.3:
        sr[0] = rng.choice([-1.0, 0.0, 1e-300])
    blo = rng.uniform(lo, hi - 0.2 * w, size=(nb, dim))
    bhi = blo + rng.uniform(0.03, 0.2, size=(nb, dim)) * w
    return (sc, sr), (blo, bhi)


def case_rv(planner):
    dim = int(rng.choice([1, 2, 2, 3, 3, 3, 4, 6, 8]))
    lo = float(rng.choice([0.0, -10.0, 100.0, -1e6]))
    hi = lo + float(rng.choice([1.0, 10.0, 20.0, 1e3]))
    w = hi - lo
    bounds = [(lo, hi)] * dim
    md = float(rng.choice([0.02, 0.05, 0.1, 0.3])) * w
    gb = float(rng.choice([0.0, 0.05, 0.5, 1.0]))
    frac = float(rng.choice([0.01, 0.05, 0.2, 1.0]))
    (sc, sr), (blo, bhi) = random_field(dim, lo, hi)
    start = rng.uniform(lo, hi, size=dim)
    goal = rng.uniform(lo, hi, size=dim)
    gr = float(rng.choice([0.02, 0.1])) * w
    seed, pid0 = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 20))
    nprob = int(rng.choice([1, 3, 9]))
    max_nodes = int(rng.choice([50, 400, 2500, 2500, 9000]))
    iters = int(rng.choice([60, 500, 3000, 3000, 12000]))
    stop = bool(rng.random() < 0.5)
    radius = float(rng.choice([0.0, 0.05, 0.15, 0.5])) * w
    kernels = [capi.KERNEL_STREAM]
    if planner == capi.PLANNER_RRT and dim in (2, 3):
        kernels += [capi.KERNEL_RESIDENT, capi.KERNEL_CELLS, capi.KERNEL_CELLS]
    if planner == capi.PLANNER_RRT and 2 <= dim <= 6:
        kernels += [capi.KERNEL_LANES, capi.KERNEL_LANES, capi.KERNEL_AUTO]   # the default path: weighted up
    if planner == capi.PLANNER_RRT_STAR and 2 <= dim <= 6:
        kernels += [capi.KERNEL_AUTO, capi.KERNEL_AUTO, capi.KERNEL_LANES]   # RRT*: the decoupled design (geometry by rrt_cells.hip / rrt_lanes.hip + rrt_star_wire.hip)
    kernel = int(rng.choice(kernels))
    flags = int(rng.choice(LANE_FLAGS)) if (kernel in (capi.KERNEL_LANES, capi.KERNEL_CELLS, capi.KERNEL_AUTO) and planner != capi.PLANNER_RRT_CONNECT) else 0
    if flags & capi.DEBUG_ALL_WHOLE_TREE:   # every query scans the whole tree with one wave: keep those cases short
        iters, max_nodes = min(iters, 3000), min(max_nodes, 2500)
    desc = dict(planner=planner, kernel=kernel, dim=dim, lo=lo, hi=hi, md=md, gb=gb, frac=frac, ns=len(sr), nb=len(blo),
                seed=seed, pid0=pid0, nprob=nprob, max_nodes=max_nodes, iters=iters, stop=stop, radius=radius, flags=flags)
    if planner == capi.PLANNER_RRT_CONNECT:
        stop = True
    try:
        g = capi.RRTBatch(dim, bounds, md, gb, nprob, max_nodes, frac, stop, seed, pid0, 0, kernel, planner, radius, debug_flags=flags,
                          frozen_split=int(rng.choice([0, 0, 1, 3, 8])))
    except capi.OxhipError as e:
        if e.status == capi.ERR_BAD_ARG:
            return None
        raise
    if len(sr):
        g.set_spheres(sc, sr)
    if len(blo):
        g.set_boxes(blo, bhi)
    g.setup(start, goal, gr)
    schedule = []
    if rng.random() < 0.4:                      # the same budget in two or three solve calls (resume)
        a = int(rng.integers(1, iters))
        schedule.append(a)
        if rng.random() < 0.5 and iters - a > 1:
            b = int(rng.integers(1, iters - a))
            schedule.append(b)
            a += b
        schedule.append(iters - a)
    else:
        schedule.append(iters)
    for a in schedule:
        g.solve(a)
    frozen = int(rng.choice([0, 0, 64, 700, 3000])) if planner == capi.PLANNER_RRT else 0   # then some iterations with inserts off
    if frozen:
        g.solve(frozen, freeze=True)
        if rng.random() < 0.3:
            g.solve(37, freeze=True)
            frozen += 37
    c = g.counts()
    for p in range(nprob):
        if planner == capi.PLANNER_RRT:
            o = orc.OracleRRT(dim, bounds, md, gb, frac, max_nodes, stop, seed, pid0 + p)
        elif planner == capi.PLANNER_RRT_STAR:
            o = orc.OracleRRTStar(dim, bounds, md, gb, radius, frac, max_nodes, stop, seed, pid0 + p)
        else:
            o = orc.OracleRRTConnect(dim, bounds, md, gb, frac, max_nodes, seed, pid0 + p)
        if len(sr):
            o.set_spheres(sc, sr)
        if len(blo):
            o.set_boxes(blo, bhi)
        o.setup(start, goal, gr)
        o.solve(iters)
        if frozen:
            o.solve(frozen, freeze=True)
        ok = int(c["checksum"][p]) == o.checksum and int(c["iterations"][p]) == o.iterations
        if planner == capi.PLANNER_RRT_CONNECT:
            gc = g.goal_counts()
            ok = ok and int(c["nodes"][p]) == o.num_nodes(0) and int(gc["nodes"][p]) == o.num_nodes(1)
            for w_, (gs, gp) in enumerate((g.tree(p), g.goal_tree(p))):
                os_, op = o.tree(w_)
                ok = ok and np.array_equal(bits(gs), bits(os_)) and np.array_equal(gp, op)
        else:
            gs, gp = g.tree(p)
            os_, op = o.tree()
            ok = ok and int(c["nodes"][p]) == o.num_nodes and int(c["goal_node"][p]) == o.goal_node
            ok = ok and np.array_equal(bits(gs), bits(os_)) and np.array_equal(gp, op)
            if planner == capi.PLANNER_RRT_STAR:
                ok = ok and np.array_equal(bits(g.costs(p)), bits(o.costs()))
        gpath, opath = g.path(p), o.path()
        ok = ok and gpath.shape == opath.shape and np.array_equal(bits(gpath), bits(opath))
        if not ok:
            if DUMP_DIR:   # everything tools/fuzz_replay.py needs to run the case again
                os.makedirs(DUMP_DIR, exist_ok=True)
                np.savez(os.path.join(DUMP_DIR, "fuzz_fail_%d_%d.npz" % (seed, pid0)), desc=json.dumps(desc), sc=sc, sr=sr, blo=blo, bhi=bhi,
                         start=start, goal=goal, gr=gr, schedule=np.array(schedule, dtype=np.int64), frozen=frozen, problem=p)
            return dict(desc, problem=p, schedule=schedule, frozen=frozen)
    g.close()
    return True


def case_prm():
    dim = int(rng.choice([2, 3, 6, 8]))
    lo, hi = 0.0, float(rng.choice([1.0, 10.0]))
    w = hi - lo
    (sc, sr), (blo, bhi) = random_field(dim, lo, hi)
    n = int(rng.choice([50, 700, 3000]))
    radius = float(rng.choice([0.0, 0.05, 0.12, 0.3])) * w * (1.0 if dim <= 3 else 2.5)
    seed, stream = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 20))
    frac = float(rng.choice([0.02, 0.05, 0.3]))
    cap = int(rng.choice([0, 0, n + n // 3]))
    desc = dict(planner="prm", dim=dim, hi=hi, n=n, radius=radius, seed=seed, stream=stream, frac=frac, ns=len(sr), nb=len(blo),
                max_samples=cap)
    try:
        g = capi.PRMRoadmap(dim, [(lo, hi)] * dim, radius, n, 0.0, frac, cap, seed, stream)
    except capi.OxhipError as e:
        if e.status == capi.ERR_BAD_ARG:
            return None
        raise
    o = orc.OraclePRM(dim, [(lo, hi)] * dim, radius, lvs_fraction=frac, seed=seed, stream=stream)
    for x in (g, o):
        if len(sr):
            x.set_spheres(sc, sr)
        if len(blo):
            x.set_boxes(blo, bhi)
    start, goal = rng.uniform(lo, hi, size=dim), rng.uniform(lo, hi, size=dim)
    g.setup(start, goal, 0.15 * w)
    o.setup(start, goal, 0.15 * w)
    g.construct_roadmap()
    o.construct_roadmap(n, cap if cap else 2 ** 62)
    gs, goff, gn = g.roadmap()
    os_, ooff, on = o.roadmap()
    ok = np.array_equal(bits(gs), bits(os_)) and np.array_equal(goff, ooff) and np.array_equal(gn, on)
    ok = ok and g.sizes()[2] == o.num_samples
    for _ in range(3):
        s2, g2 = rng.uniform(lo, hi, size=dim), rng.uniform(lo, hi, size=dim)
        g.set_problem(s2, g2, 0.2 * w)
        o.set_problem(s2, g2, 0.2 * w)
        st, path = g.solve()
        ok = ok and st == o.solve()
        opath = o.path()
        ok = ok and path.shape == opath.shape and np.array_equal(bits(path), bits(opath))
    g.close()
    return True if ok else desc


def case_se2():
    nseg = int(rng.choice([0, 4, 60, 256]))
    segs = rng.uniform(0.0, 10.0, size=(nseg, 4))
    segs[:, 2:] = segs[:, :2] + rng.normal(0.0, 0.6, size=(nseg, 2))
    th = [(-np.pi, np.pi), (-1.0, 2.0), (-10.0, 10.0), (0.5, 0.6)][int(rng.integers(0, 4))]
    md, gb = float(rng.choice([0.2, 0.5, 1.5])), float(rng.choice([0.0, 0.05, 0.5]))
    seed, pid0 = int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 2 ** 20))
    clear = float(rng.choice([0.0, 0.05, 0.2]))
    start = [float(rng.uniform(0, 10)), float(rng.uniform(0, 10)), float(rng.uniform(-4, 4))]
    goal = [float(rng.uniform(0, 10)), float(rng.uniform(0, 10)), float(rng.uniform(-4, 4))]
    nprob, iters, max_nodes = int(rng.choice([1, 5])), int(rng.choice([100, 1500])), int(rng.choice([60, 2000]))
    # (round 3) the lookup grid switched off in a quarter of the cases; the solve cut into launches that end inside a sampled block
    flags = capi.DEBUG_SE2_NO_SEGMENT_GRID if rng.integers(0, 4) == 0 else 0
    cuts = sorted(int(v) for v in rng.integers(1, iters, size=int(rng.integers(0, 3))))
    desc = dict(planner="se2", nseg=nseg, th=th, md=md, gb=gb, seed=seed, pid0=pid0, clear=clear, nprob=nprob, iters=iters,
                max_nodes=max_nodes, flags=flags, cuts=cuts)
    g = capi.RRTBatch(3, [(0.0, 10.0), (0.0, 10.0), th], md, gb, nprob, max_nodes, 0.05, True, seed, pid0, 0, 0,
                      capi.PLANNER_RRT_CONNECT, 0.0, capi.SPACE_SE2, debug_flags=flags)
    g.set_segments(segs, clear)
    g.setup(start, goal, 0.4)
    done = 0
    for cut in cuts + [iters]:
        if cut > done:
            g.solve(cut - done)
            done = cut
    c, gc = g.counts(), g.goal_counts()
    for p in range(nprob):
        o = orc.OracleSE2Connect([(0.0, 10.0), (0.0, 10.0)], th, md, gb, 0.05, max_nodes, seed, pid0 + p)
        o.set_segments(segs, clear)
        o.setup(start, goal, 0.4)
        o.solve(iters)
        ok = int(c["checksum"][p]) == o.checksum and int(c["iterations"][p]) == o.iterations
        for w_, (gs, gp) in enumerate((g.tree(p), g.goal_tree(p))):
            os_, op = o.tree(w_)
            ok = ok and np.array_equal(bits(gs), bits(os_)) and np.array_equal(gp, op)
        gpath, opath = g.path(p), o.path()
        ok = ok and gpath.shape == opath.shape and np.array_equal(bits(gpath), bits(opath))
        if not ok:
            return dict(desc, problem=p)
    g.close()
    return True


CASES = [("rrt", lambda: case_rv(capi.PLANNER_RRT)), ("rrt", lambda: case_rv(capi.PLANNER_RRT)),
         ("rrt_connect", lambda: case_rv(capi.PLANNER_RRT_CONNECT)), ("rrt_star", lambda: case_rv(capi.PLANNER_RRT_STAR)),
         ("prm", case_prm), ("se2_connect", case_se2)]


def sweep(budget_s, seed, only=None, verbose=True):
    """run random cases for `budget_s` seconds from `seed`; returns (cases per planner, list of failures)"""
    global rng
    rng = np.random.default_rng(seed)
    cases = [c for c in CASES if only is None or c[0] == only]
    counts, failures = {}, []
    t0 = time.perf_counter()
    i = 0
    slowest = 0.0
    while time.perf_counter() - t0 < budget_s:
        name, fn = cases[i % len(cases)]
        i += 1
        t_case = time.perf_counter()
        r = fn()
        if verbose and time.perf_counter() - t_case > slowest:
            slowest = time.perf_counter() - t_case
            print("slowest case so far: %s %.2f s" % (name, slowest), flush=True)
        if r is None:
            continue
        counts[name] = counts.get(name, 0) + 1
        if r is not True:
            failures.append(r)
            print("FAIL", json.dumps({k: (v if not isinstance(v, tuple) else list(v)) for k, v in r.items()}), flush=True)
        if verbose and i % 50 == 0:
            print("progress", counts, "failures", len(failures), flush=True)
    return counts, failures


if __name__ == "__main__":
    BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    counts, failures = sweep(BUDGET, int(sys.argv[2]) if len(sys.argv) > 2 else 1, sys.argv[3] if len(sys.argv) > 3 else None)
    print(json.dumps({"seconds": BUDGET, "cases": counts, "failures": len(failures)}))
    sys.exit(1 if failures else 0)
