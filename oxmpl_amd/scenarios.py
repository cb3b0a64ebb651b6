"""Synthetic workloads named by BASELINE.json's configs (inputs only; no planner logic).

config 1: README quick-start, R^2, disc obstacle r=2 at the origin, start (-5,-5), goal (5,5) r=0.5
          (/root/reference README.md:147-171, oxmpl-js/examples/simple_2d_planning.js:8-41)
wall    : the reference's own RRT test scene (oxmpl/tests/rrt_rvss_tests.rs:109-159)
config 2: R^3, [0,10]^3, 64 random spheres, start (.5,.5,.5), goal ball (9.5,9.5,9.5) r=0.5
config 4: SE(2) RRTConnect (BASELINE.json configs[3]): [0,10]^2 x [-PI,PI), 64 random quadrilaterals = 256 segments,
          disc robot of radius 0.15, start (0.5,0.5,0), goal ball (9.5,9.5,1.5) r=0.5
config 5: PRM (BASELINE.json configs[4]): R^6, [0,10]^6, 32 random hyperspheres, 50,000 milestones,
          connection radius 2.0, start (1,..,1), goal ball (9,..,9) r=1.5
"""
import struct

import numpy as np

_M64 = 0xFFFFFFFFFFFFFFFF


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & _M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return state, z ^ (z >> 31)


def sphere_field(seed=0x5EED0001, n=64, dim=3, lo=1.0, hi=9.0, rmin=0.3, rmax=0.8, keep_clear=()):
    """n spheres: centres U[lo,hi)^dim, radii U[rmin,rmax) from SplitMix64(seed) through the
    52-bit [1,2)-1 transform; a sphere closer than radius+0.5 to a keep_clear point is redrawn."""
    st = seed

    def u(a, b):
        nonlocal st
        st, z = _splitmix64(st)
        bits = (z >> 12) | 0x3FF0000000000000
        v = struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0
        return v * (b - a) + a

    cs, rs = [], []
    while len(rs) < n:
        c = [u(lo, hi) for _ in range(dim)]
        r = u(rmin, rmax)
        ok = True
        for p in keep_clear:
            acc = 0.0
            for x, y in zip(c, p):
                acc = acc + (x - y) * (x - y)
            if not (acc ** 0.5 > r + 0.5):
                ok = False
        if ok:
            cs.append(c)
            rs.append(r)
    return np.array(cs, dtype=np.float64), np.array(rs, dtype=np.float64)


def config1():
    return dict(dim=2, bounds=[(-10.0, 10.0), (-10.0, 10.0)], max_distance=0.5, goal_bias=0.05, lvs_fraction=0.05,
                start=[-5.0, -5.0], goal_centre=[5.0, 5.0], goal_radius=0.5,
                spheres=(np.array([[0.0, 0.0]]), np.array([2.0])), boxes=None)


def wall():
    return dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], max_distance=0.5, goal_bias=0.0, lvs_fraction=0.05,
                start=[1.0, 5.0], goal_centre=[9.0, 5.0], goal_radius=0.5, spheres=None,
                boxes=(np.array([[4.75, 2.0]]), np.array([[5.25, 8.0]])))


def config2():
    start, goal = [0.5, 0.5, 0.5], [9.5, 9.5, 9.5]
    return dict(dim=3, bounds=[(0.0, 10.0)] * 3, max_distance=0.5, goal_bias=0.05, lvs_fraction=0.05,
                start=start, goal_centre=goal, goal_radius=0.5,
                spheres=sphere_field(keep_clear=[start, goal]), boxes=None)


def polygon_soup(seed=0x5EED0003, n_poly=64, lo=0.8, hi=9.2, wmin=0.15, wmax=0.45, keep_clear=(), margin=0.4):
    """n_poly random quadrilaterals (4 segments each): centre U[lo,hi)^2, half-widths U[wmin,wmax), every vertex
    jittered by up to 30 %; polygons too close to a keep_clear point are redrawn.  SplitMix64, no trigonometry."""
    st = seed

    def u(a, b):
        nonlocal st
        st, z = _splitmix64(st)
        bits = (z >> 12) | 0x3FF0000000000000
        v = struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0
        return v * (b - a) + a

    segs = []
    while len(segs) < 4 * n_poly:
        cx, cy = u(lo, hi), u(lo, hi)
        w, h = u(wmin, wmax), u(wmin, wmax)
        verts = [(cx + sx * w * u(0.7, 1.3), cy + sy * h * u(0.7, 1.3)) for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        size = 1.3 * max(w, h) * (2.0 ** 0.5)
        if any(((cx - p[0]) ** 2 + (cy - p[1]) ** 2) ** 0.5 <= size + margin for p in keep_clear):
            continue
        for k in range(4):
            a, b = verts[k], verts[(k + 1) % 4]
            segs.append((a[0], a[1], b[0], b[1]))
    return np.array(segs, dtype=np.float64)


def config4():
    start, goal = [0.5, 0.5, 0.0], [9.5, 9.5, 1.5]
    pi = 3.141592653589793
    return dict(dim=3, bounds=[(0.0, 10.0), (0.0, 10.0), (-pi, pi)], max_distance=0.5, goal_bias=0.05, lvs_fraction=0.05,
                start=start, goal_centre=goal, goal_radius=0.5, clearance=0.15,
                segments=polygon_soup(keep_clear=[start, goal]), spheres=None, boxes=None)


def make_se2_batch(sc, n_problems, max_nodes=10000, seed=42, first_problem_id=0, device=0):
    """RRTConnect over SE(2) among sc["segments"] (oxmpl_amd.capi.SPACE_SE2), set up and ready to solve."""
    from .capi import RRTBatch, KERNEL_AUTO, PLANNER_RRT_CONNECT, SPACE_SE2
    b = RRTBatch(3, sc["bounds"], sc["max_distance"], sc["goal_bias"], n_problems, max_nodes, sc["lvs_fraction"], True,
                 seed, first_problem_id, device, KERNEL_AUTO, PLANNER_RRT_CONNECT, 0.0, SPACE_SE2)
    b.set_segments(sc["segments"], sc["clearance"])
    b.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return b


def config5():
    start, goal = [1.0] * 6, [9.0] * 6
    return dict(dim=6, bounds=[(0.0, 10.0)] * 6, connection_radius=2.0, lvs_fraction=0.05, max_milestones=50000,
                start=start, goal_centre=goal, goal_radius=1.5,
                spheres=sphere_field(seed=0x5EED0005, n=32, dim=6, lo=1.0, hi=9.0, rmin=2.0, rmax=3.5,
                                     keep_clear=[start, goal]), boxes=None)


def make_prm(sc, max_milestones=None, seed=42, stream=0, device=0, timeout=0.0, connection_radius=None, knn_k=0):
    """Build a PRMRoadmap for a scenario dict and run Planner::setup."""
    from .capi import PRMRoadmap
    g = PRMRoadmap(sc["dim"], sc["bounds"], connection_radius or sc["connection_radius"],
                   max_milestones or sc["max_milestones"], timeout, sc["lvs_fraction"], 0, seed, stream, device, knn_k)
    if sc["spheres"] is not None:
        g.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        g.set_boxes(*sc["boxes"])
    g.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return g


def make_batch(sc, n_problems, max_nodes=10000, stop_at_goal=True, seed=42, first_problem_id=0, device=0, kernel=0,
               planner=0, search_radius=0.0, **extra):
    """Build an RRTBatch for a scenario dict and run Planner::setup on every problem."""
    from .capi import RRTBatch
    b = RRTBatch(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], n_problems, max_nodes,
                 sc["lvs_fraction"], stop_at_goal, seed, first_problem_id, device, kernel, planner, search_radius, **extra)
    if sc["spheres"] is not None:
        b.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        b.set_boxes(*sc["boxes"])
    b.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return b
