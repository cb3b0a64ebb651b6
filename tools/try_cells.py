#!/usr/bin/env python3
"""Development check of the cell-grid kernel: parity against the oracle on small cases, against rrt_lanes.hip on full-size
ones, and timings.  usage: try_cells.py [stage]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def oracle(sc, seed, pid, max_nodes, stop):
    o = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"], max_nodes, stop, seed, pid)
    if sc["spheres"] is not None:
        o.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        o.set_boxes(*sc["boxes"])
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return o


def compare(g, planners, tag):
    c = g.counts()
    bad = 0
    for p, o in enumerate(planners):
        ok = int(c["nodes"][p]) == o.num_nodes and int(c["iterations"][p]) == o.iterations and int(c["checksum"][p]) == o.checksum \
            and int(c["accepted"][p]) == o.accepted and int(c["goal_node"][p]) == o.goal_node
        if ok:
            gs, gp = g.tree(p)
            os_, op = o.tree()
            ok = np.array_equal(gp, op) and np.array_equal(bits(gs), bits(os_))
        if not ok:
            bad += 1
            if bad <= 3:
                print("  MISMATCH", tag, "problem", p, "gpu", int(c["nodes"][p]), int(c["iterations"][p]), hex(int(c["checksum"][p])),
                      "oracle", o.num_nodes, o.iterations, hex(o.checksum), "stop", int(c["stop_reason"][p]))
    print(tag, "OK" if bad == 0 else "FAILED (%d of %d)" % (bad, len(planners)), flush=True)
    return bad == 0


stage = sys.argv[1] if len(sys.argv) > 1 else "all"
K = capi.KERNEL_CELLS
allok = True
if stage in ("all", "small"):
    for name, sc in (("config1", scenarios.config1()), ("wall", scenarios.wall()), ("config2", scenarios.config2())):
        for iters, stop in ((300, True), (3000, False)):
            P = 8
            g = scenarios.make_batch(sc, P, 10000, stop, 7, 50, 0, K)
            g.enable_stamps(True)
            g.solve(iters // 3)
            g.solve(iters - iters // 3)
            pl = [oracle(sc, 7, 50 + p, 10000, stop) for p in range(P)]
            orc.solve_many(pl, iters, threads=8)
            allok &= compare(g, pl, "%s grow %d stop=%s" % (name, iters, stop))
            if not stop:
                for split in (0,):
                    g.solve(500, freeze=True)
                    orc.solve_many(pl, 500, freeze=True, threads=8)
                    allok &= compare(g, pl, "%s frozen 500" % name)
            s = g.stamps()
            print("   stamps: rounds %d lanes %d amb %d expand %d steps %d regrid %d memo %d cuts %d" %
                  (s[5], s[6], s[4], s[8], s[9], s[10], s[11], s[12]))
            g.close()
if stage in ("all", "split"):
    sc = scenarios.config2()
    P = 8
    pl = [oracle(sc, 42, p, 10000, False) for p in range(P)]
    orc.solve_many(pl, 4000, threads=8)
    orc.solve_many(pl, 1000, freeze=True, threads=8)
    for split in (1, 2, 3, 8):
        g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, K, frozen_split=split)
        g.solve(4000)
        g.solve(600, freeze=True)
        g.solve(400, freeze=True)
        allok &= compare(g, pl, "config2 4000 + frozen 1000, split %d" % split)
        g.close()
if stage in ("all", "full"):
    sc = scenarios.config2()
    P = 1024
    ref = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, capi.KERNEL_LANES)
    ref.solve(10 ** 7)
    print("lanes grow: %.2f ms" % ref.last_timing()["kernel_ms"])
    rc = ref.counts()
    for split in (1, 2, 4):
        g = scenarios.make_batch(sc, P, 10000, False, 42, 0, 0, K, frozen_split=split)
        w = scenarios.make_batch(sc, 4, 10000, False, 42, 0, 0, K)
        w.solve(100)
        w.close()
        t0 = time.perf_counter()
        g.solve(10 ** 7)
        dt = time.perf_counter() - t0
        c = g.counts()
        same = bool((c["checksum"] == rc["checksum"]).all() and (c["iterations"] == rc["iterations"]).all() and (c["nodes"] == rc["nodes"]).all())
        its = int(c["iterations"].sum())
        print("cells grow: kernel %.2f ms wall %.2f ms = %.1f M it/s, same as lanes: %s" % (g.last_timing()["kernel_ms"], dt * 1e3, its / dt / 1e6, same), flush=True)
        allok &= same
        if split == 1:
            ref.solve(4096, freeze=True)
            ref.solve(4096, freeze=True)
            print("lanes frozen: %.2f ms" % ref.last_timing()["kernel_ms"])
            rc2 = ref.counts()
        g.solve(4096, freeze=True)
        g.solve(4096, freeze=True)
        ms = g.last_timing()["kernel_ms"]
        c = g.counts()
        same = bool((c["checksum"] == rc2["checksum"]).all() and (c["iterations"] == rc2["iterations"]).all() and (c["accepted"] == rc2["accepted"]).all())
        print("cells frozen split %d: %.2f ms = %.1f M it/s, same as lanes: %s" % (split, ms, P * 4096 / ms / 1e3, same), flush=True)
        allok &= same
        g.close()
print("ALL OK" if allok else "SOME FAILED")
sys.exit(0 if allok else 1)
