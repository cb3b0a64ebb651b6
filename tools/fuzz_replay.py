#!/usr/bin/env python3
"""Run one failing case of tools/fuzz_parity.py again (RRT / RRT* planners) from the .npz it wrote under $FUZZ_DUMP_DIR and say
where the GPU and the oracle part.  usage: fuzz_replay.py case.npz [kernel kind]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

z = np.load(sys.argv[1])
d = json.loads(str(z["desc"]))
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else d["kernel"]
dim, planner = d["dim"], d["planner"]
bounds = [(d["lo"], d["hi"])] * dim
g = capi.RRTBatch(dim, bounds, d["md"], d["gb"], d["nprob"], d["max_nodes"], d["frac"], d["stop"], d["seed"], d["pid0"], 0, kernel,
                  planner, d["radius"], debug_flags=d.get("flags", 0))
if len(z["sr"]):
    g.set_spheres(z["sc"], z["sr"])
if len(z["blo"]):
    g.set_boxes(z["blo"], z["bhi"])
g.setup(z["start"], z["goal"], float(z["gr"]))
if os.environ.get("FUZZ_STAMPS"):
    g.enable_stamps(True)
for a in z["schedule"]:
    g.solve(int(a))
frozen = int(z["frozen"])
if frozen:
    g.solve(frozen, freeze=True)
c = g.counts()
bad = 0
for p in range(d["nprob"]):
    if planner == capi.PLANNER_RRT:
        o = orc.OracleRRT(dim, bounds, d["md"], d["gb"], d["frac"], d["max_nodes"], d["stop"], d["seed"], d["pid0"] + p)
    else:
        o = orc.OracleRRTStar(dim, bounds, d["md"], d["gb"], d["radius"], d["frac"], d["max_nodes"], d["stop"], d["seed"], d["pid0"] + p)
    if len(z["sr"]):
        o.set_spheres(z["sc"], z["sr"])
    if len(z["blo"]):
        o.set_boxes(z["blo"], z["bhi"])
    o.setup(z["start"], z["goal"], float(z["gr"]))
    o.solve(int(sum(z["schedule"])))
    if frozen:
        o.solve(frozen, freeze=True)
    gs, gp = g.tree(p)
    os_, op = o.tree()
    n = min(len(gp), len(op))
    same = (gs[:n].view(np.uint64) == os_[:n].view(np.uint64)).all(axis=1) & (gp[:n] == op[:n])
    first = int(np.argmin(same)) if not same.all() else -1
    ok = int(c["checksum"][p]) == o.checksum and int(c["nodes"][p]) == o.num_nodes and first < 0
    print("problem %d: %s  gpu nodes %d iterations %d checksum %016x stop %d | oracle nodes %d iterations %d checksum %016x | first differing node %d"
          % (p, "ok" if ok else "MISMATCH", int(c["nodes"][p]), int(c["iterations"][p]), int(c["checksum"][p]), int(c["stop_reason"][p]) if "stop_reason" in c else -1,
             o.num_nodes, o.iterations, o.checksum, first))
    if first >= 0:
        print("   gpu   ", gs[first], gp[first])
        print("   oracle", os_[first], op[first])
    bad += not ok
print("kernel kind run:", g.last_timing())
if os.environ.get("FUZZ_STAMPS"):
    st = g.stamps()
    print("audit: accepted-but-invalid end states %d; last: problem %d m %d lane %d iteration %d, sphere mask %016x, path %d"
          % (int(st[50]), int(st[51]) >> 48, (int(st[51]) >> 40) & 0xFF, (int(st[51]) >> 32) & 0xFF, int(st[51]) & 0xFFFFFFFF, int(st[52]), int(st[53])))
sys.exit(1 if bad else 0)
