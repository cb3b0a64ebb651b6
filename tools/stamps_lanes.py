#!/usr/bin/env python3
"""Diagnostic counters of the lane-per-query kernel (workgroup 0): rounds, lanes per round, conflict cuts, exact-path events.
usage: stamps_lanes.py [P] [iters]   (grow phase, then `iters` frozen iterations)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

sc = scenarios.config2()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
gpu = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_LANES)
gpu.enable_stamps(True)
gpu.solve(10 ** 7)
s = gpu.stamps()
print("grow: kernel %.3f ms; problem 0: %d iterations in %d rounds (%.1f lanes offered, %.1f committed per round), %d conflict cuts, %d exact-path"
      % (gpu.last_timing()["kernel_ms"], int(s[7]), int(s[5]), int(s[6]) / max(1, int(s[5])), int(s[7]) / max(1, int(s[5])), int(s[12]), int(s[4])))
itg = int(s[7])
print("  resolver: wait %.0f  work %.0f cycles per iteration (%.0f per round); exact path %.0f per iteration (%.0f per event); lifetime %.3f ms at %.2f GHz"
      % (int(s[1]) / itg, int(s[2]) / itg, int(s[2]) / max(1, int(s[5])), int(s[3]) / itg, int(s[3]) / max(1, int(s[4])), int(s[14]) / 1e5, int(s[13]) / max(1, int(s[14])) / 10.0))
print("  scanner waves: wait", " ".join("%6.0f" % (int(v) / itg) for v in s[16:24]))
print("                 work", " ".join("%6.0f" % (int(v) / itg) for v in s[24:32]))
PH = ("combine+candidates", "ring fold", "coords+steer", "sphere filter", "motion check", "prefix (cap, goal, conflicts)", "commit")
print("  resolver phases, cycles per round:", ", ".join("%s %.0f" % (nm, int(s[32 + i]) / max(1, int(s[5]))) for i, nm in enumerate(PH)))
print("  per round: ring-fold trips of eight %.1f, nodes that needed binary64 %.1f; conflict trips %.1f, exact tests %.1f"
      % tuple(int(s[40 + i]) / max(1, int(s[5])) for i in range(4)))
print("  over all problems: exact-path events max %d (problem %d), mean %.1f; resolver lifetime max %.0f k cycles (problem %d), mean %.0f k"
      % (int(s[44]) >> 32, int(s[44]) & 0xFFFFFFFF, int(s[45]) / P, (int(s[46]) >> 16) / 1e3, int(s[46]) & 0xFFFF, int(s[47]) / P / 1e3))
print("  rounds with a two-lane candidate pass: mean %.1f per problem" % (int(s[48]) / P))
gpu.enable_stamps(True)   # (resets the cross-problem accumulators)
gpu.solve(iters, freeze=True)
s2 = gpu.stamps()
it = int(s2[7]) - int(s[7])
print("steady@10k: kernel %.3f ms (diagnostic build); problem 0: %d iterations in %d rounds (%.1f per round), %d exact-path"
      % (gpu.last_timing()["kernel_ms"], it, int(s2[5]), it / max(1, int(s2[5])), int(s2[4])))
print("resolver: wait %.0f  work %.0f cycles per iteration (%.0f per round)" % (int(s2[1]) / it, int(s2[2]) / it, int(s2[2]) / max(1, int(s2[5]))))
print("scanner waves: wait   ", " ".join("%6.0f" % (int(v) / it) for v in s2[16:24]))
print("               work   ", " ".join("%6.0f" % (int(v) / it) for v in s2[24:32]))
print("resolver lifetime: %d cycles in %.3f ms (100 MHz clock) = %.2f GHz; exact path %.0f cycles per iteration (%.0f per event)"
      % (int(s2[13]), int(s2[14]) / 1e5, int(s2[13]) / max(1, int(s2[14])) / 10.0, int(s2[3]) / it, int(s2[3]) / max(1, int(s2[4]))))
print("literal-loop (true near-tie) events: %d; whole-tree answers reused: %d" % (int(s2[15]), int(s2[11])))
print("resolver phases, cycles per round:", ", ".join("%s %.0f" % (nm, int(s2[32 + i]) / max(1, int(s2[5]))) for i, nm in enumerate(PH)))
print("over all problems: exact-path events max %d (problem %d), mean %.1f; resolver lifetime max %.0f k cycles (problem %d), mean %.0f k"
      % (int(s2[44]) >> 32, int(s2[44]) & 0xFFFFFFFF, int(s2[45]) / P, (int(s2[46]) >> 16) / 1e3, int(s2[46]) & 0xFFFF, int(s2[47]) / P / 1e3))
print("rounds with a two-lane candidate pass: mean %.1f per problem" % (int(s2[48]) / P))
