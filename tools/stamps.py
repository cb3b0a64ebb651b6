#!/usr/bin/env python3
"""Diagnostic: phase shares of the resident kernel from in-kernel cycle stamps (workgroup 0).
Never quote this build's run time; read the shares."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

sc = scenarios.config2()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
gpu = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_RESIDENT)
gpu.solve(10 ** 7)
gpu.enable_stamps(True)
gpu.solve(iters, freeze=True)
s = gpu.stamps()
names = ["scan+publish", "barrier1", "resolve|sample", "barrier2", "-", "-", "verdict+insert"]
tot = float(sum(int(v) for v in s[:7]))
print("steady@10k, %d iterations, kernel %.3f ms" % (iters, gpu.last_timing()["kernel_ms"]))
for nme, v in zip(names, s[:7]):
    print("  %-16s %12d cyc  %6.1f cyc/iter  %5.1f %%" % (nme, int(v), int(v) / iters, 100.0 * int(v) / tot))
print("  total %.1f cyc/iter" % (tot / iters))
print("  per-wave arrival at barrier 1 after barrier-3 release (cyc/iter):")
print("   ", " ".join("%5.0f" % (int(v) / iters) for v in s[16:32]))
