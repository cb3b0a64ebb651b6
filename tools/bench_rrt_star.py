#!/usr/bin/env python3
"""Measurement of the RRT* row: BASELINE.json configs[1] scene (R^3, 64 spheres), 1024 independent
problems grown from 1 to max_nodes nodes with RRTStar(0.5, 0.05, search_radius) on one MI355X, next to the CPU
oracle on a bounded sample of the same problems.  Usage: bench_rrt_star.py [problems] [max_nodes] [search_radius] [kernel]
(kernel 0 = KERNEL_AUTO: the decoupled design -- geometry by rrt_cells.hip (R^2 / R^3) or rrt_lanes.hip, wiring by rrt_star_wire.hip; 1 = rrt_star.hip)"""
import concurrent.futures as cf
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

sc = scenarios.config2()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
R = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
KERNEL = int(sys.argv[4]) if len(sys.argv) > 4 else 0
CPU_P, CPU_N = 16, min(N, 3000)

ms, its = [], 0
for rep in range(3):
    g = scenarios.make_batch(sc, P, N, False, 42, 0, 0, KERNEL, capi.PLANNER_RRT_STAR, R)
    t0 = time.perf_counter()
    g.solve(10 ** 9)
    wall = time.perf_counter() - t0
    c = g.counts()
    assert (c["nodes"] == N).all()
    if rep:
        ms.append(g.last_timing()["kernel_ms"])
        its = int(c["iterations"].sum())
        goal_cost = float(np.mean([g.costs(p)[int(c["goal_node"][p])] for p in range(0, P, max(1, P // 32)) if c["goal_node"][p] >= 0]))
    if rep < 2:
        g.close()
k = float(np.mean(ms))
# CPU oracle: CPU_P of the same problems to CPU_N nodes on 16 threads; the first CPU_N nodes must be identical
planners = []
for p in range(CPU_P):
    o = orc.OracleRRTStar(3, sc["bounds"], 0.5, 0.05, R, 0.05, CPU_N, False, 42, p)
    o.set_spheres(*sc["spheres"])
    o.setup(sc["start"], sc["goal_centre"], 0.5)
    planners.append(o)
t0 = time.perf_counter()
with cf.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda o: o.solve(10 ** 9), planners))
cpu_dt = time.perf_counter() - t0
cpu_its = sum(o.iterations for o in planners)
same = True
for p in (0, CPU_P - 1):
    gs, _ = g.tree(p)
    os_, _ = planners[p].tree()
    same = same and bool(np.array_equal(gs[:CPU_N].view(np.uint64), os_.view(np.uint64)))
# the wiring: run the oracle's problem 0 to the full size and compare parents (after rewiring) and costs bit for bit
o_full = orc.OracleRRTStar(3, sc["bounds"], 0.5, 0.05, R, 0.05, N, False, 42, 0)
o_full.set_spheres(*sc["spheres"])
o_full.setup(sc["start"], sc["goal_centre"], 0.5)
o_full.solve(10 ** 9)
_, gp0 = g.tree(0)
_, op0 = o_full.tree()
wiring_same = bool(np.array_equal(gp0, op0) and np.array_equal(g.costs(0).view(np.uint64), o_full.costs().view(np.uint64))
                   and int(c["checksum"][0]) == o_full.checksum)
# algorithmic bytes: two scans of the tree per accepted iteration (nearest + find_neighbours: 2 * 24 * n at n nodes),
# one per rejected iteration (rejections are spread over the growth: charged at the mean tree size)
acc = P * (N - 1)
alg_bytes = 2 * 24 * P * (N - 1) * N / 2 + (its - acc) * 24 * N / 2
print(json.dumps({"roofline": {"bound": "hbm", "achieved": alg_bytes / (k * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": alg_bytes / (k * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": alg_bytes},
                  "planner": "RRTStar", "design": {capi.KERNEL_LANES: "decoupled: rrt_lanes.hip + rrt_star_wire.hip", capi.KERNEL_CELLS: "decoupled: rrt_cells.hip + rrt_star_wire.hip"}.get(g.last_timing()["kernel"], "one kernel: rrt_star.hip"),
                  "workload": "R^3, 64 spheres, %d problems, 1 -> %d nodes, search radius %g" % (P, N, R),
                  "kernel_ms": k, "iterations": its, "iterations_per_s": its / (k * 1e-3),
                  "nodes_per_s": P * (N - 1) / (k * 1e-3), "mean_goal_cost_sample": goal_cost,
                  "cpu_oracle": {"kind": "port", "problems": CPU_P, "max_nodes": CPU_N, "threads": 16,
                                 "iterations_per_s": cpu_its / cpu_dt, "states_identical_on_sample": same,
                                 "problem0_full_size_parents_costs_checksum_identical": wiring_same}}))
