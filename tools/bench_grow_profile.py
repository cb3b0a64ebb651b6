#!/usr/bin/env python3
"""Where does a growing tree spend its time?  configs[1] grown to 10,000 nodes in stages (resume), kernel time per stage,
for the lane-group resolver (kernel 4) and the lane-per-query resolver (kernel 5).  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxmpl_amd import capi, scenarios  # noqa: E402

sc = scenarios.config2()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
stages = [250, 500, 1000, 2000, 4000, 7000, 10000]
out = {}
for kernel in (capi.KERNEL_RESIDENT_F32, capi.KERNEL_LANES):
    rows = []
    prev_it = 0
    # a stage = every problem runs until ITS tree reaches the stage's node count: emulate with per-stage batches grown from scratch
    for n in stages:
        g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, kernel)
        # grow to n nodes: iterations needed differ per problem; run in chunks until all trees hold >= n nodes is not expressible,
        # so use the iteration count of the slowest problem at that size from a reference run
        g.close()
    g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, kernel)
    done = 0
    for n in stages:
        # iterations ~ nodes * 1.05 in this scene: step the whole batch by iteration budget
        target_it = int(n * 1.05)
        g.solve(target_it - done)
        ms = g.last_timing()["kernel_ms"]
        c = g.counts()
        rows.append(dict(iterations_to=target_it, mean_nodes=float(c["nodes"].mean()), kernel_ms=ms,
                         it_per_s=P * (target_it - done) / (ms * 1e-3)))
        done = target_it
    g.close()
    out["resident_f32" if kernel == capi.KERNEL_RESIDENT_F32 else "lanes"] = rows
print(json.dumps(out))
