#!/usr/bin/env python3
"""Measurement of BASELINE.json configs[3]: SE(2) RRTConnect among 256 segments, P independent problems solved to
completion on one MI355X, next to the CPU oracle on a bounded sample.  Usage: bench_connect_se2.py [problems]"""
import concurrent.futures as cf
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time  # noqa: E402
import numpy as np  # noqa: E402
from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

sc = scenarios.config4()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ms = []
for rep in range(4):
    g = scenarios.make_se2_batch(sc, P, 10000, 42 + rep)
    st = g.solve(10 ** 7)
    c, gc = g.counts(), g.goal_counts()
    assert (st == capi.OK).all()
    if rep:
        ms.append(g.last_timing()["kernel_ms"])
        its = int(c["iterations"].sum())
        nodes = float((c["nodes"] + gc["nodes"]).mean())
    if rep < 3:
        g.close()
k = float(np.mean(ms))
CPU_P = 64
planners = []
for p in range(CPU_P):
    o = orc.OracleSE2Connect(sc["bounds"][:2], sc["bounds"][2], 0.5, 0.05, 0.05, 10000, 45, p)
    o.set_segments(sc["segments"], sc["clearance"])
    o.setup(sc["start"], sc["goal_centre"], 0.5)
    planners.append(o)
t0 = time.perf_counter()
with cf.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda o: o.solve(10 ** 7), planners))
cpu_dt = time.perf_counter() - t0
cpu_its = sum(o.iterations for o in planners)
same = all(np.array_equal(g.path(p).view(np.uint64), planners[p].path().view(np.uint64)) for p in (0, 1, CPU_P - 1))
print(json.dumps({"planner": "RRTConnect/SE(2)", "workload": "SE(2), 256 segments, %d problems to completion" % P,
                  "kernel_ms": k, "problems_per_s": P / (k * 1e-3), "iterations": its, "iterations_per_s": its / (k * 1e-3),
                  "mean_nodes_both_trees": nodes,
                  "cpu_oracle": {"kind": "port", "problems": CPU_P, "threads": 16, "problems_per_s": CPU_P / cpu_dt,
                                 "iterations_per_s": cpu_its / cpu_dt, "paths_identical_on_sample": bool(same)}}))
