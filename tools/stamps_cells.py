#!/usr/bin/env python3
"""In-kernel phase stamps of rrt_cells.hip (diagnostic instantiation; never quote this build's run time): problem 0's wave.
usage: stamps_cells.py [P] [split]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
split = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = scenarios.config2()
names = ["own cell + needed neighbour cells", "candidate f64 + steer", "sphere filter", "motion check", "prefix (cap, goal, conflicts)", "commit",
         "shell searches (cooperative)", "flat scan (trees <= 1024 nodes)"]


def show(tag, g, iters):
    s = g.stamps()
    rounds = max(1, int(s[5]))
    print("%s: kernel %.2f ms; problem 0: rounds %d, lanes offered %d, iterations %d -> %.1f committed per round; wave lifetime %d cycles = %d per round"
          % (tag, g.last_timing()["kernel_ms"], rounds, int(s[6]), iters, iters / rounds, int(s[13]), int(s[13]) // rounds))
    for i, nm in enumerate(names):
        print("   %-32s %8d cycles per round" % (nm, int(s[32 + i]) // rounds))
    print("   whole-tree events %d (ties %d), memo hits %d, shell searches %d, chain steps %d (%.1f per round), regrids %d, conflict cuts %d"
          % (int(s[4]), int(s[15]), int(s[11]), int(s[8]), int(s[9]), int(s[9]) / rounds, int(s[10]), int(s[12])))
    print("   batch-wide: whole-tree events %d, shell searches %d" % (int(s[54]), int(s[60])))
    ci = (40, 41, 42, 43, 44, 46, 47, 48)
    print("   neighbour trips by number (trips: lanes asking / cells asked for, per trip): " +
          ", ".join("%d: %.1f / %.1f" % (int(s[24 + i]), int(s[16 + i]) / max(1, int(s[24 + i])), int(s[ci[i]]) / max(1, int(s[24 + i]))) for i in range(8)))
    print("   tail passes %d, pairs per pass %.1f" % (int(s[1]), int(s[2]) / max(1, int(s[1]))))
    print("   commit split: before insert %d, insert (atomic) %d, reductions %d, checksum+rest %d per round" % tuple(int(v) // rounds for v in (s[62], s[63], s[49], s[0])))


g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_CELLS, frozen_split=split)
g.enable_stamps(True)
g.solve(10 ** 7)
show("grow 1 -> 10,000 nodes", g, int(g.counts()["iterations"][0]))
g.enable_stamps(True)   # (re-zeroes the counters)
g.solve(4096, freeze=True)
show("steady@10k, 4096 iterations, split %d (part 0 of problem 0)" % split, g, 4096 // split)
