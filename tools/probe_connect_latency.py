#!/usr/bin/env python3
"""Per-iteration latency of the RRTConnect kernels (DESIGN.md sections 8 and 11): kernel time of the slowest
problem divided by its iteration count, for one problem alone on the chip and for a full batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402


def probe(name, make):
    for P in (1, 1024):
        make(P).solve(10 ** 7)                  # warm-up (code object load, first touch)
        g = make(P)
        g.solve(10 ** 7)
        it = g.counts()["iterations"]
        ms = g.last_timing()["kernel_ms"]
        print("%-18s P=%4d  kernel %.3f ms  slowest problem %d iterations (mean %.0f)  -> %.2f us per iteration"
              % (name, P, ms, int(it.max()), float(it.mean()), ms * 1e3 / float(it.max())))


sc2, sc4 = scenarios.config2(), scenarios.config4()
probe("RRTConnect R^3", lambda P: scenarios.make_batch(sc2, P, 10000, True, 43, 0, 0, 0, capi.PLANNER_RRT_CONNECT))
probe("RRTConnect SE(2)", lambda P: scenarios.make_se2_batch(sc4, P, 10000, 43))
