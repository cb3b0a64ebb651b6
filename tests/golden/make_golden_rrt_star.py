#!/usr/bin/env python3
"""Generates tests/golden/rrt_star_golden.json: RRT* runs computed by an independent numpy / pure-Python
restatement of oxmpl's RRTStar::solve (oxmpl/src/geometric/planners/rrt_star.rs:170-289: sample, nearest,
steer, check_motion, find_neighbours :121-131, choose parent :225-241, push :244-250, rewire :253-282, goal
:285-288) on the primitives of make_golden.py.  Independent of oracle/rrt_oracle.c (numpy row operations,
Python lists); the two pin each other.  PARITY UNPINNED against a rustc-built oxmpl: the reference's
RRT* tests (oxmpl/tests/rrt_star_rvss_tests.rs) assert properties only.

Run:  python tests/golden/make_golden_rrt_star.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (ChaCha12Rng, Field, check_motion, distance, f64_bits, hexf, interpolate,  # noqa: E402
                         random_bool, random_range, sphere_field)

FNV_P = 0x100000001B3
M64 = (1 << 64) - 1


def row_dists(a, rows):
    """distance(a, row) for every row: sequential sum over k of (a_k - row_k)^2, sqrt (rvss.rs:137-155)"""
    acc = np.zeros(len(rows))
    for k in range(len(a)):
        d = a[k] - rows[:, k]
        acc = acc + d * d
    return np.sqrt(acc)


def rrt_star_solve(dim, bounds, max_distance, goal_bias, search_radius, fraction, field, start, goal_c, goal_r, seed, pid,
                   max_iterations, max_nodes, stop_at_goal=True):
    rng = ChaCha12Rng(seed, pid)
    tree = np.zeros((max_nodes + 1, dim), dtype=np.float64)
    tree[0] = start
    parents, cost = [-1], [0.0]
    n = 1
    h_iter, w_wire = 0xCBF29CE484222325, 0       # checksum = H + W: iteration polynomial + wiring polynomial (oracle/rrt_oracle.c)
    iterations = accepted = rewires = 0
    goal_node = -1
    for _ in range(max_iterations):
        if n >= max_nodes:
            break
        q_rand = list(goal_c) if random_bool(rng, goal_bias) else [random_range(rng, lo, hi) for lo, hi in bounds]
        acc = np.zeros(n)
        for k in range(dim):                       # distance(tree[i], q_rand)
            d = tree[:n, k] - q_rand[k]
            acc = acc + d * d
        dists = np.sqrt(acc)
        nearest = int(np.argmin(dists))
        min_dist = float(dists[nearest])
        q_near = [float(v) for v in tree[nearest]]
        q_new = interpolate(q_near, q_rand, max_distance / min_dist) if min_dist > max_distance else list(q_rand)
        ok = check_motion(field, bounds, fraction, q_near, q_new)
        g = ((0xCBF29CE484222325 ^ nearest) * FNV_P) & M64
        for v in q_new:
            g = ((g ^ f64_bits(v)) * FNV_P) & M64
        g = ((g ^ int(ok)) * FNV_P) & M64
        h_iter = (h_iter * FNV_P + g) & M64
        iterations += 1
        if not ok:
            continue
        accepted += 1
        dn = row_dists(q_new, tree[:n])            # distance(q_new, tree[i]); symmetric in its arguments
        neighbours = [int(i) for i in np.nonzero(dn < search_radius)[0]]
        best_parent, min_cost = nearest, cost[nearest] + float(dn[nearest])
        for i in neighbours:                       # choose parent, ascending index, strict <
            c = cost[i] + float(dn[i])
            if c < min_cost and check_motion(field, bounds, fraction, [float(v) for v in tree[i]], q_new):
                min_cost, best_parent = c, i
        tree[n] = q_new
        parents.append(best_parent)
        cost.append(min_cost)
        new_idx = n
        n += 1
        rew_cnt = rew_sum = 0
        for i in neighbours:                       # rewire
            if i == best_parent:
                continue
            c2 = min_cost + float(dn[i])
            if c2 < cost[i] and check_motion(field, bounds, fraction, q_new, [float(v) for v in tree[i]]):
                parents[i] = new_idx
                cost[i] = c2
                rew_cnt += 1
                rew_sum += i
        rewires += rew_cnt
        w = 0xCBF29CE484222325
        for v in (best_parent, f64_bits(min_cost), rew_cnt, rew_sum):
            w = ((w ^ v) * FNV_P) & M64
        w_wire = (w_wire * FNV_P + w) & M64
        if distance(q_new, goal_c) <= goal_r:
            if goal_node < 0:
                goal_node = new_idx
            if stop_at_goal:
                break
    path = []
    if goal_node >= 0:
        i = goal_node
        while i >= 0:
            path.append([float(v) for v in tree[i]])
            i = parents[i]
        path.reverse()
    chk = (h_iter + w_wire) & M64
    return dict(n=n, iterations=iterations, accepted=accepted, rewires=rewires, checksum=chk, goal_node=goal_node,
                states=tree[:n].copy(), parents=parents, cost=cost, path=path)


def record(res, head):
    return dict(n=res["n"], iterations=res["iterations"], accepted=res["accepted"], rewires=res["rewires"],
                checksum="%016x" % res["checksum"], goal_node=res["goal_node"],
                states=[[hexf(v) for v in row] for row in res["states"][:head]],
                parents=[int(p) for p in res["parents"]], cost=[hexf(c) for c in res["cost"]],
                path=[[hexf(v) for v in row] for row in res["path"]])


def main():
    out = {}
    # ---- the reference's RRT* test scene (oxmpl/tests/rrt_star_rvss_tests.rs:109-165) with a wider search radius
    cw = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], max_distance=0.5, goal_bias=0.0, search_radius=1.0, fraction=0.05,
              start=[1.0, 5.0], goal_c=[9.0, 5.0], goal_r=0.5, spheres=[], boxes=[([4.75, 2.0], [5.25, 8.0])],
              max_nodes=20000, max_iterations=200000)
    fw = Field(2, [], cw["boxes"])
    runs = []
    for seed in range(3):
        res = rrt_star_solve(2, cw["bounds"], 0.5, 0.0, 1.0, 0.05, fw, cw["start"], cw["goal_c"], 0.5, seed, 7, 200000, 20000)
        rec = record(res, 64)
        rec.update(seed=seed, pid=7)
        runs.append(rec)
    out["wall"] = dict(params=cw, runs=runs)
    # the reference test's own parameters: RRTStar::new(0.5, 0.0, 0.25) (rrt_star_rvss_tests.rs:148)
    cr = dict(cw, search_radius=0.25)
    runs = []
    for seed in range(2):
        res = rrt_star_solve(2, cr["bounds"], 0.5, 0.0, 0.25, 0.05, fw, cr["start"], cr["goal_c"], 0.5, seed, 8, 200000, 20000)
        rec = record(res, 64)
        rec.update(seed=seed, pid=8)
        runs.append(rec)
    out["wall_ref"] = dict(params=cr, runs=runs)
    # ---- README scene (config 1) with goal bias
    c1 = dict(dim=2, bounds=[(-10.0, 10.0), (-10.0, 10.0)], max_distance=0.5, goal_bias=0.05, search_radius=1.25,
              fraction=0.05, start=[-5.0, -5.0], goal_c=[5.0, 5.0], goal_r=0.5, spheres=[([0.0, 0.0], 2.0)], boxes=[],
              max_nodes=20000, max_iterations=200000)
    f1 = Field(2, c1["spheres"])
    runs = []
    for seed in range(3):
        res = rrt_star_solve(2, c1["bounds"], 0.5, 0.05, 1.25, 0.05, f1, c1["start"], c1["goal_c"], 0.5, seed, 3, 200000, 20000)
        rec = record(res, 64)
        rec.update(seed=seed, pid=3)
        runs.append(rec)
    out["config1"] = dict(params=c1, runs=runs)
    # ---- config 2 field (R^3, 64 spheres), fixed iteration budget, no stop at goal
    start3, goal3 = [0.5, 0.5, 0.5], [9.5, 9.5, 9.5]
    spheres = sphere_field(0x5EED0001, 64, 3, 1.0, 9.0, 0.3, 0.8, [start3, goal3])
    c2 = dict(dim=3, bounds=[(0.0, 10.0)] * 3, max_distance=0.5, goal_bias=0.05, search_radius=1.0, fraction=0.05,
              start=start3, goal_c=goal3, goal_r=0.5, boxes=[], spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in spheres],
              max_nodes=10000, max_iterations=1200)
    f2 = Field(3, spheres)
    runs = []
    for pid in (0, 5):
        res = rrt_star_solve(3, c2["bounds"], 0.5, 0.05, 1.0, 0.05, f2, start3, goal3, 0.5, 42, pid, 1200, 10000,
                             stop_at_goal=False)
        rec = record(res, 128)
        rec.update(seed=42, pid=pid)
        runs.append(rec)
    out["config2"] = dict(params=c2, runs=runs)
    path = os.path.join(HERE, "rrt_star_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)
    for k, v in out.items():
        for r in v["runs"]:
            print(k, "seed", r["seed"], "n", r["n"], "iters", r["iterations"], "rewires", r["rewires"], "goal", r["goal_node"],
                  "path", len(r["path"]), r["checksum"])


if __name__ == "__main__":
    main()
