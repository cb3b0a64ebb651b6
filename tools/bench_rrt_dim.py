#!/usr/bin/env python3
"""RRT in a higher-dimensional space through the general (stream) kernel: the config-5 sphere field of R^6 (or its first
`dim` coordinates), P problems grown to `nodes` nodes, then frozen iterations at that size.
    python tools/bench_rrt_dim.py [dim=6] [P=1024] [nodes=10000] [iters=1024]
One JSON line: grow and steady iterations/s, the kernel's algorithmic bytes (n * dim * 8 per iteration)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import scenarios  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 6
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
s5 = scenarios.config5()
c, r = s5["spheres"]
sc = dict(dim=dim, bounds=[(0.0, 10.0)] * dim, max_distance=1.0, goal_bias=0.05, lvs_fraction=0.05,
          start=[1.0] * dim, goal_centre=[9.0] * dim, goal_radius=1.0,
          spheres=(np.ascontiguousarray(np.hstack([c, np.full((len(c), max(0, dim - 6)), 5.0)])[:, :dim]), r * (0.45 if dim < 6 else 1.0)),
          boxes=None)
# a small launch first: the first launch of a process pays milliseconds for loading the code object, inside any HIP-event bracket
w = scenarios.make_batch(sc, 4, nodes, False, 42, 0, 0, int(os.environ.get("OXHIP_KERNEL", "0")))
w.solve(256)
w.close()
g = scenarios.make_batch(sc, P, nodes, False, 42, 0, 0, int(os.environ.get("OXHIP_KERNEL", "0")))
g.solve(10 ** 8)
t = g.last_timing()
cts = g.counts()
assert (cts["nodes"] == nodes).all(), "the scene does not let every tree reach the node count"
grow_its = float(cts["iterations"].sum())
grow_ms = t["kernel_ms"]
g.solve(iters, freeze=True)
ms = 0.0
for _ in range(3):
    g.solve(iters, freeze=True)
    ms += g.last_timing()["kernel_ms"]
ms /= 3
its = P * iters / (ms * 1e-3)
print(json.dumps({"planner": "RRT", "dim": dim, "problems": P, "nodes": nodes, "kernel": {1: "stream", 2: "resident", 5: "lanes", 6: "cells"}[g.last_timing()["kernel"]],
                  "grow_iterations_per_s": grow_its / (grow_ms * 1e-3), "grow_kernel_ms": grow_ms,
                  "steady_iterations_per_s": its, "steady_kernel_ms": ms,
                  "roofline": {"bound": "hbm", "achieved": its * nodes * dim * 8 / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": its * nodes * dim * 8 / 1e9 / 8000.0}}))
