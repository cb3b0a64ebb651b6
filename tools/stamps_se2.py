#!/usr/bin/env python3
"""Where an SE(2) RRTConnect iteration's cycles go: the stamped instantiation of rrt_connect_se2.hip, problem 0
of a batch of P (default 1024) problems of BASELINE.json configs[3]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import scenarios  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenarios.config4()
scenarios.make_se2_batch(sc, P, 10000, 43).solve(10 ** 7)   # warm-up
g = scenarios.make_se2_batch(sc, P, 10000, 43)
g.enable_stamps(True)
g.solve(10 ** 7)
s = g.stamps()
names = ["sample", "rounds: nearest", "rounds: steer", "rounds: motion check", "whole-wave extends", "commit (incl. whole-wave extends)", "whole loop"]
it, ex, rounds, failed = int(s[7]), int(s[8]), int(s[9]), int(s[10])
print("problem 0 of %d: %d iterations in %d rounds (%d committed as failures by a round), %d whole-wave extends, kernel %.3f ms"
      % (P, it, rounds, failed, ex, g.last_timing()["kernel_ms"]))
for k in range(7):
    print("  %-34s %12d cycles  %8.0f per iteration  %8.0f per round" % (names[k], int(s[k]), int(s[k]) / max(it, 1), int(s[k]) / max(rounds, 1)))
