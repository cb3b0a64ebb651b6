#!/usr/bin/env python3
"""Generates tests/golden/se2_golden.json: RRTConnect over SE(2) = R^2 x SO(2) among line segments
(BASELINE.json configs[3]) computed by an independent numpy / pure-Python restatement.

The reference has no SE(2) space (docs/BACKLOG.md:12-14); the build defines it from the reference's components:
RealVectorStateSpace (real_vector_state_space.rs:137-186, 233-249) for (x, y), SO2StateSpace
(so2_state_space.rs:97-122, 164-169; so2_state.rs:33-37) for theta, distance = 1.0 * d_xy + 0.5 * d_theta,
extent = extent_xy + 0.5 * PI; planner = rrt_connect.rs:121-159, 166-189, 227-309.  See oracle/se2_oracle.h for
the full statement.  PARITY UNPINNED against oxmpl (nothing to compare with: the space does not exist there).

Run:  python tests/golden/make_golden_se2.py
"""
import json
import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (ChaCha12Rng, distance, f64_bits, hexf, interpolate, maximum_extent, num_steps,  # noqa: E402
                         random_bool, random_range, splitmix64)

PI = math.pi
FNV_P = 0x100000001B3
M64 = (1 << 64) - 1


def rem_euclid(a, b):
    r = math.fmod(a, b)
    return r + abs(b) if r < 0.0 else r


def so2_normalise(v):
    return rem_euclid(v + PI, 2.0 * PI) - PI


def so2_distance(a, b):
    diff = a - b
    diff = rem_euclid(diff + PI, 2.0 * PI) - PI
    return abs(diff)


def so2_interpolate(frm, to, t):
    d = so2_normalise(to) - so2_normalise(frm)
    if d > PI:
        d -= 2.0 * PI
    elif d < -PI:
        d += 2.0 * PI
    return so2_normalise(frm + d * t)


def se2_distance(a, b):
    return distance(a[:2], b[:2]) + 0.5 * so2_distance(a[2], b[2])


def se2_interpolate(frm, to, t):
    xy = interpolate(frm[:2], to[:2], t)
    return [xy[0], xy[1], so2_interpolate(frm[2], to[2], t)]


def point_segment_distance(px, py, seg):
    ax, ay, bx, by = seg
    abx, aby = bx - ax, by - ay
    apx, apy = px - ax, py - ay
    len2 = abx * abx + aby * aby
    t = 0.0
    if len2 > 0.0:
        t = (apx * abx + apy * aby) / len2
    if not (t > 0.0):
        t = 0.0
    if t > 1.0:
        t = 1.0
    cx, cy = ax + abx * t, ay + aby * t
    dx, dy = px - cx, py - cy
    return math.sqrt(dx * dx + dy * dy)


class Soup:
    def __init__(self, segs, clearance):
        self.segs = [tuple(float(v) for v in s) for s in segs]
        self.clearance = clearance

    def is_valid(self, s):
        return all(point_segment_distance(s[0], s[1], g) > self.clearance for g in self.segs)


def se2_extent(bounds_xy):
    return maximum_extent(bounds_xy) + 0.5 * PI


def check_motion(soup, bounds_xy, fraction, frm, to):
    dist = se2_distance(frm, to)
    n = num_steps(dist, se2_extent(bounds_xy) * fraction)
    if n <= 1:
        return soup.is_valid(to)
    for i in range(1, n + 1):
        if not soup.is_valid(se2_interpolate(frm, to, float(i) / float(n))):
            return False
    return True


def tree_dists(rows, n, q):
    """se2_distance(tree[i], q) for i < n, numpy rows"""
    dx = rows[:n, 0] - q[0]
    dy = rows[:n, 1] - q[1]
    dr = np.sqrt(dx * dx + dy * dy)
    diff = rows[:n, 2] - q[2]
    r = np.fmod(diff + PI, 2.0 * PI)
    r = np.where(r < 0.0, r + 2.0 * PI, r)
    return dr + 0.5 * np.abs(r - PI)


def se2_connect_solve(bounds_xy, th_bounds, max_distance, goal_bias, fraction, soup, start, goal, goal_r, seed, pid,
                      max_iterations, max_nodes):
    rng = ChaCha12Rng(seed, pid)
    th_lo, th_hi = max(th_bounds[0], -PI), min(th_bounds[1], PI)
    trees = [np.zeros((max_nodes + 1, 3)), np.zeros((max_nodes + 1, 3))]
    parents = [[-1], [-1]]
    n = [1, 1]
    trees[0][0], trees[1][0] = start, goal
    chk = 0xCBF29CE484222325
    iterations = 0
    end = [-1, -1]

    def extend(w, target):
        d = tree_dists(trees[w], n[w], target)
        nearest = int(np.argmin(d))
        min_dist = float(d[nearest])
        q_near = [float(v) for v in trees[w][nearest]]
        if min_dist > max_distance:
            q_new, res = se2_interpolate(q_near, target, max_distance / min_dist), 1
        else:
            q_new, res = list(target), 2
        if not check_motion(soup, bounds_xy, fraction, q_near, q_new):
            return 0, nearest, q_new
        trees[w][n[w]] = q_new
        parents[w].append(nearest)
        n[w] += 1
        return res, nearest, q_new

    for _ in range(max_iterations):
        if n[0] >= max_nodes or n[1] >= max_nodes:
            break
        grow_start = n[0] <= n[1]
        if random_bool(rng, goal_bias):
            q_rand = list(goal)
        else:
            q_rand = [random_range(rng, bounds_xy[0][0], bounds_xy[0][1]), random_range(rng, bounds_xy[1][0], bounds_xy[1][1]),
                      random_range(rng, th_lo, th_hi)]
        wa = 0 if grow_start else 1
        wb = 1 - wa
        ra, near_a, qa = extend(wa, q_rand)
        for v in (int(grow_start), near_a, *[f64_bits(x) for x in qa], ra):
            chk = ((chk ^ v) * FNV_P) & M64
        iterations += 1
        done = False
        if ra:
            idx_a = n[wa] - 1
            if grow_start and se2_distance(qa, goal) <= goal_r:
                end = [idx_a, -1]
                done = True
            else:
                rb, near_b, qb = extend(wb, qa)
                for v in (near_b, *[f64_bits(x) for x in qb], rb):
                    chk = ((chk ^ v) * FNV_P) & M64
                if rb == 2:
                    end[wa], end[wb] = idx_a, n[wb] - 1
                    done = True
        if done:
            break
    path = []
    if end[0] >= 0:
        i = end[0]
        while i >= 0:
            path.append([float(v) for v in trees[0][i]])
            i = parents[0][i]
        path.reverse()
        if end[1] >= 0:
            i = parents[1][end[1]]
            while i >= 0:
                path.append([float(v) for v in trees[1][i]])
                i = parents[1][i]
    return dict(n=n, iterations=iterations, checksum=chk, end=end, path=path,
                states=[trees[0][:n[0]].copy(), trees[1][:n[1]].copy()], parents=parents)


def polygon_soup(seed, n_poly, lo, hi, wmin, wmax, keep_clear, margin):
    """n_poly random quadrilaterals (4 segments each): centre U[lo,hi)^2, half-widths U[wmin,wmax) with each vertex
    jittered by up to 30 % of them; a polygon whose centre is within margin + its size of a keep_clear point is redrawn.
    SplitMix64 + the 52-bit [1,2)-1 transform, no transcendental functions."""
    st = seed

    def u(a, b):
        nonlocal st
        st, z = splitmix64(st)
        bits = (z >> 12) | 0x3FF0000000000000
        v = struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0
        return v * (b - a) + a

    segs = []
    while len(segs) < 4 * n_poly:
        cx, cy = u(lo, hi), u(lo, hi)
        w, h = u(wmin, wmax), u(wmin, wmax)
        verts = []
        for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            verts.append((cx + sx * w * u(0.7, 1.3), cy + sy * h * u(0.7, 1.3)))
        size = 1.3 * max(w, h) * math.sqrt(2.0)
        if any(distance([cx, cy], p[:2]) <= size + margin for p in keep_clear):
            continue
        for k in range(4):
            a, b = verts[k], verts[(k + 1) % 4]
            segs.append((a[0], a[1], b[0], b[1]))
    return segs


def record(res, head):
    return dict(n=res["n"], iterations=res["iterations"], checksum="%016x" % res["checksum"], end=res["end"],
                path=[[hexf(v) for v in row] for row in res["path"]],
                states=[[[hexf(v) for v in row] for row in t[:head]] for t in res["states"]],
                parents=[[int(x) for x in pp[:head]] for pp in res["parents"]])


def main():
    out = {}
    # ---- arithmetic KATs of the SO(2) / SE(2) / segment primitives
    r = ChaCha12Rng(5, 5)
    kat = []
    for _ in range(24):
        a = [random_range(r, -10.0, 10.0), random_range(r, -10.0, 10.0), random_range(r, -7.0, 7.0)]
        b = [random_range(r, -10.0, 10.0), random_range(r, -10.0, 10.0), random_range(r, -7.0, 7.0)]
        t = random_range(r, 0.0, 1.0)
        seg = [random_range(r, -10.0, 10.0) for _ in range(4)]
        kat.append(dict(a=[hexf(v) for v in a], b=[hexf(v) for v in b], t=hexf(t), seg=[hexf(v) for v in seg],
                        so2_normalise=hexf(so2_normalise(a[2])), so2_distance=hexf(so2_distance(a[2], b[2])),
                        so2_interpolate=hexf(so2_interpolate(a[2], b[2], t)), se2_distance=hexf(se2_distance(a, b)),
                        se2_interpolate=[hexf(v) for v in se2_interpolate(a, b, t)],
                        point_segment=hexf(point_segment_distance(a[0], a[1], seg))))
    edge = [(PI, -PI), (-PI, PI), (3.0, -3.0), (0.0, 0.0), (-PI, -PI), (PI - 1e-16, -PI), (7.0, -7.0), (1e-300, -1e-300)]
    out["kat"] = dict(random=kat,
                      so2_edges=[dict(a=hexf(a), b=hexf(b), normalise=hexf(so2_normalise(a)), distance=hexf(so2_distance(a, b)),
                                      interp_half=hexf(so2_interpolate(a, b, 0.5))) for a, b in edge],
                      degenerate_segment=hexf(point_segment_distance(1.0, 2.0, (3.0, 4.0, 3.0, 4.0))),
                      extent=hexf(se2_extent([(0.0, 10.0), (0.0, 10.0)])))
    # ---- configs[3] scene: [0,10]^2 x [-PI,PI), 64 quadrilaterals = 256 segments, disc robot of radius 0.15
    start, goal = [0.5, 0.5, 0.0], [9.5, 9.5, 1.5]
    segs = polygon_soup(0x5EED0003, 64, 0.8, 9.2, 0.15, 0.45, [start, goal], 0.4)
    soup = Soup(segs, 0.15)
    assert soup.is_valid(start) and soup.is_valid(goal)
    p3 = dict(bounds_xy=[(0.0, 10.0), (0.0, 10.0)], theta_bounds=[-PI, PI], max_distance=0.5, goal_bias=0.05, fraction=0.05,
              start=start, goal=goal, goal_r=0.5, clearance=0.15, segments=[[hexf(v) for v in s] for s in segs],
              max_nodes=20000, max_iterations=200000)
    runs = []
    for seed in range(4):
        res = se2_connect_solve(p3["bounds_xy"], p3["theta_bounds"], 0.5, 0.05, 0.05, soup, start, goal, 0.5, seed, 21, 200000, 20000)
        rec = record(res, 48)
        rec.update(seed=seed, pid=21)
        runs.append(rec)
    out["soup256"] = dict(params=p3, runs=runs)
    # ---- narrow theta bounds + a wall with a gap: sampling in a sub-arc, wrap-around never taken
    wall = [(5.0, 0.0, 5.0, 4.0), (5.0, 5.0, 5.0, 10.0)]
    pw = dict(bounds_xy=[(0.0, 10.0), (0.0, 10.0)], theta_bounds=[-1.0, 2.0], max_distance=0.75, goal_bias=0.0, fraction=0.05,
              start=[1.0, 5.0, -0.5], goal=[9.0, 5.0, 1.5], goal_r=0.5, clearance=0.2, segments=[[hexf(v) for v in s] for s in wall],
              max_nodes=20000, max_iterations=200000)
    sw = Soup(wall, 0.2)
    runs = []
    for seed in range(3):
        res = se2_connect_solve(pw["bounds_xy"], pw["theta_bounds"], 0.75, 0.0, 0.05, sw, pw["start"], pw["goal"], 0.5, seed, 4, 200000, 20000)
        rec = record(res, 48)
        rec.update(seed=seed, pid=4)
        runs.append(rec)
    out["gap"] = dict(params=pw, runs=runs)
    path = os.path.join(HERE, "se2_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)
    for k in ("soup256", "gap"):
        for r in out[k]["runs"]:
            print(k, "seed", r["seed"], "n", r["n"], "iters", r["iterations"], "end", r["end"], "path", len(r["path"]), r["checksum"])


if __name__ == "__main__":
    main()
