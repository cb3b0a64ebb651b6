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
KERN = int(sys.argv[3]) if len(sys.argv) > 3 else capi.KERNEL_RESIDENT
gpu = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, KERN)
gpu.solve(10 ** 7)
gpu.enable_stamps(True)
gpu.solve(iters, freeze=True)
s = gpu.stamps()
print("steady@10k, %d iterations, kernel %.3f ms (diagnostic build)" % (iters, gpu.last_timing()["kernel_ms"]))
print("resolver wave: sample %.0f  wait-for-scanners %.0f  resolve+commit %.0f  cyc/iter" % tuple(int(v) / iters for v in s[:3]))
print("   of which combine %.0f cyc/iter; exact-path events %d of %d" % (int(s[3]) / iters, int(s[4]), iters))
print("scanner waves: wait   ", " ".join("%6.0f" % (int(v) / iters) for v in s[16:24]))
print("               scan   ", " ".join("%6.0f" % (int(v) / iters) for v in s[24:32]))
print("scanner wave 5 per query: pre (absorb, q loads) %.0f  scan %.0f  reduce+publish %.0f cyc" % (int(s[8]) / iters, int(s[9]) / iters, int(s[10]) / iters))
