#!/usr/bin/env python3
"""How the two RRT* designs behave in the regime that is worst for the decoupled one: a search radius of
half the world (nearly every node is every node's neighbour) and a fine motion resolution, so that checking EVERY neighbour
pair's motions -- instead of only those whose cost test passes -- is expensive.  Prints wall seconds per design and checks
that both give the same parents, costs and checksums."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oxmpl_amd import capi, scenarios

sc = scenarios.config2()
P, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
out = {}
for name, kernel, frac, radius in (("decoupled", capi.KERNEL_AUTO, 0.01, 5.0), ("one_kernel", capi.KERNEL_STREAM, 0.01, 5.0),
                                   ("decoupled_r1", capi.KERNEL_AUTO, 0.05, 1.0), ("one_kernel_r1", capi.KERNEL_STREAM, 0.05, 1.0)):
    s2 = dict(sc, lvs_fraction=frac)
    g = scenarios.make_batch(s2, P, N, False, 7, 0, 0, kernel, capi.PLANNER_RRT_STAR, radius)
    t0 = time.perf_counter()
    g.solve(10 ** 9)
    dt = time.perf_counter() - t0
    c = g.counts()
    out[name] = (dt, c["checksum"].copy(), g.tree(0)[1].copy(), g.costs(0).copy())
    print("%-14s radius %.1f lvs_fraction %.2f: %d problems to %d nodes in %.2f s" % (name, radius, frac, P, N, dt), flush=True)
    g.close()
for a, b in (("decoupled", "one_kernel"), ("decoupled_r1", "one_kernel_r1")):
    same = bool((out[a][1] == out[b][1]).all() and np.array_equal(out[a][2], out[b][2]) and
                np.array_equal(out[a][3].view(np.uint64), out[b][3].view(np.uint64)))
    print(a, "==", b, ":", same)
    assert same
