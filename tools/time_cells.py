#!/usr/bin/env python3
"""configs[1] through the cell-grid kernel: grow 1 -> 10,000 nodes and steady@10k timings (HIP events), checksums against
rrt_lanes.hip.  usage: time_cells.py [split ...]   (OXMPL_HIP_LIB selects a build variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

sc = scenarios.config2()
P = 1024
splits = [int(a) for a in sys.argv[1:]] or [0]
ref = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_LANES)
ref.solve(10 ** 7)
rc = ref.counts()
for _ in range(3):
    ref.solve(4096, freeze=True)
rc2 = ref.counts()
w = scenarios.make_batch(sc, 4, 10000, False, 42, 0, 0, capi.KERNEL_CELLS)
w.solve(100)
w.close()
for split in splits:
    g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_CELLS, frozen_split=split)
    g.solve(10 ** 7)
    gms = g.last_timing()["kernel_ms"]
    c = g.counts()
    same = bool((c["checksum"] == rc["checksum"]).all() and (c["nodes"] == rc["nodes"]).all())
    ms = []
    for _ in range(3):
        g.solve(4096, freeze=True)
        ms.append(g.last_timing()["kernel_ms"])
    c = g.counts()
    same2 = bool((c["checksum"] == rc2["checksum"]).all() and (c["accepted"] == rc2["accepted"]).all())
    print("%s split %d: grow %.2f ms = %.0f M it/s (%s); steady %s ms = %.0f M it/s (%s)" %
          (os.environ.get("OXMPL_HIP_LIB", "product"), split, gms, int(c["iterations"].sum() - 3 * P * 4096) / gms / 1e3, same,
           ["%.3f" % m for m in ms], P * 4096 / min(ms) / 1e3, same2), flush=True)
    g.close()
