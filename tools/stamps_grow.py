#!/usr/bin/env python3
"""Diagnostic: resident-kernel phase shares while GROWING (commit every accepted iteration)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

sc = scenarios.config2()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gpu = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, int(sys.argv[3]) if len(sys.argv) > 3 else capi.KERNEL_RESIDENT)
gpu.enable_stamps(True)
for target in (2000, 4000, 6000):
    gpu.solve(target)
    s = gpu.stamps()
    it = int(s[7])
    n = int(gpu.counts()["nodes"][0])
    print("after %5d iterations (n=%5d): kernel %.3f ms; resolver sample %.0f wait %.0f combine %.0f rest %.0f cyc/iter; "
          "scanner0 wait %.0f scan %.0f; scanner5 wait %.0f scan %.0f; batches through the sequential path %d"
          % (it, n, gpu.last_timing()["kernel_ms"], int(s[0]) / target, int(s[1]) / target, int(s[3]) / target,
             int(s[2]) / target, int(s[16]) / target, int(s[24]) / target, int(s[21]) / target, int(s[29]) / target, int(s[5])))
