#!/usr/bin/env python3
"""Measurement of the RRTConnect row: BASELINE.json configs[1] scene (R^3, 64 spheres),
1024 independent problems solved to completion on one MI355X, next to the CPU oracle on a bounded sample."""
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
ms, its = [], 0
for rep in range(4):
    gpu = scenarios.make_batch(sc, P, 10000, True, 42 + rep, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
    t0 = time.perf_counter()
    st = gpu.solve(10 ** 6)
    dt = time.perf_counter() - t0
    c, gc = gpu.counts(), gpu.goal_counts()
    assert (st == capi.OK).all()
    if rep:  # first repetition warms up
        ms.append(gpu.last_timing()["kernel_ms"])
        its = int(c["iterations"].sum())
        nodes = float((c["nodes"] + gc["nodes"]).mean())
    gpu.close()
k = float(np.mean(ms))
# CPU oracle, 64 of the same problems on 16 threads
planners = []
for p in range(64):
    o = orc.OracleRRTConnect(3, sc["bounds"], 0.5, 0.05, 0.05, 10000, 44, p)
    o.set_spheres(*sc["spheres"])
    o.setup(sc["start"], sc["goal_centre"], 0.5)
    planners.append(o)
t0 = time.perf_counter()
import concurrent.futures as cf  # noqa: E402
with cf.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda o: o.solve(10 ** 6), planners))
cpu_dt = time.perf_counter() - t0
cpu_its = sum(o.iterations for o in planners)
print(json.dumps({"planner": "RRTConnect", "problems": P, "kernel_ms": k, "problems_per_s": P / (k * 1e-3),
                  "iterations": its, "iterations_per_s": its / (k * 1e-3), "mean_nodes_both_trees": nodes,
                  "cpu_oracle": {"problems": 64, "threads": 16, "problems_per_s": 64 / cpu_dt, "iterations_per_s": cpu_its / cpu_dt}}))
