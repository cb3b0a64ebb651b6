#!/usr/bin/env python3
"""Where an R^3 RRTConnect iteration's cycles go: needs a library built with the kernel's compile-time stamps,
    bash tools/build_variant.sh conn_stamps rrt_connect.hip "-DOXHIP_CONN_STAMPS"
    OXMPL_HIP_LIB=build_variants/conn_stamps/liboxmpl_hip.so python tools/stamps_connect.py [P]
(the product build carries no stamp code).  Problem 0 of a batch of P config-2 problems."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenarios.config2()
scenarios.make_batch(sc, P, 10000, True, 43, 0, 0, 0, capi.PLANNER_RRT_CONNECT).solve(10 ** 7)   # warm-up
g = scenarios.make_batch(sc, P, 10000, True, 43, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
g.enable_stamps(True)
g.solve(10 ** 7)
s = g.stamps()
names = ["sample", "nearest", "steer", "motion check", "-", "checksum + insert + goal test", "whole loop"]
it, ex = int(s[7]), int(s[8])
print("problem 0 of %d: %d iterations, %d extends, kernel %.3f ms" % (P, it, ex, g.last_timing()["kernel_ms"]))
for k in (0, 1, 2, 3, 5, 6):
    print("  %-30s %10d cycles  %8.0f per iteration  %8.0f per extend" % (names[k], int(s[k]), int(s[k]) / max(it, 1), int(s[k]) / max(ex, 1)))
