#!/usr/bin/env python3
"""Measurement of the PRM row: BASELINE.json configs[4] -- R^6, 50,000 milestones, radius
connection + edge validity -- on one MI355X, next to the CPU oracle on a bounded sample of the same
stream.  Usage: bench_prm.py [milestones] [connection_radius] [repeats]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

sc = scenarios.config5()
N = int(sys.argv[1]) if len(sys.argv) > 1 else sc["max_milestones"]
R = float(sys.argv[2]) if len(sys.argv) > 2 else sc["connection_radius"]
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 5
CPU_N = int(os.environ.get("PRM_CPU_N", "50000"))   # oracle: the same stream on one core (O(n^2); a few seconds at 50,000)

g = scenarios.make_prm(sc, N, connection_radius=R)
phases, walls = [], []
for rep in range(REP + 1):
    g.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])   # clears the roadmap
    t0 = time.perf_counter()
    g.construct_roadmap()
    dt = time.perf_counter() - t0
    if rep:
        t = g.last_timing()
        phases.append(t["phase_ms"][:4])
        walls.append(dt * 1e3)
n, entries, samples = g.sizes()
t = g.last_timing()
st, path = g.solve()
tq = g.last_timing()["phase_ms"]
ph = np.mean(np.array(phases), axis=0)
pairs = n * (n - 1) // 2
dim = sc["dim"]
# the pair search screens in packed binary32: per pair and lane dim subtractions + dim squares / fmas, two pairs per
# instruction = dim VALU instructions per pair.  Peak = the MEASURED issue rate of packed binary32 instructions on this
# chip (tools/valu_mix_bench.hip -> profiles/r2_valu_peak.json: ns per wave64 instruction per SIMD at 4 waves per SIMD,
# the kernel's occupancy), x 64 lanes x SIMDs -- round 1 assumed 16 lanes x 2.4 GHz = 39.3 T/s
def packed_f32_peak():
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2_valu_peak.json")) as f:
            d = json.load(f)
        ns = [d["op_issue"][k]["4w"]["ns"] for k in ("v_pk_fma_f32", "v_pk_add_f32")]
        return d["cus"] * 4 * 64 / (sum(ns) / len(ns) * 1e-9) / 1e12, "profiles/r2_valu_peak.json: v_pk_fma_f32 / v_pk_add_f32, 4 waves per SIMD"
    except (OSError, KeyError, ValueError):
        return 39.3, "assumed: 1024 SIMDs x 16 lanes x 2.4 GHz"


PEAK_T, PEAK_SRC = packed_f32_peak()
instr_pair = dim
pair_s = ph[1] * 1e-3
out = {
    "planner": "PRM", "workload": "R^6, 32 hyperspheres, %d milestones, connection radius %g" % (n, R),
    "milestones": n, "samples": samples, "undirected_edges": entries // 2, "mean_degree": entries / n,
    "in_radius_pairs": t["candidates"], "construct_wall_ms": float(np.mean(walls)),
    "phase_ms": {"sample": ph[0], "pairs": ph[1], "edges": ph[2], "sort_csr": ph[3]},
    "milestones_per_s": n / (float(np.mean(walls)) * 1e-3),
    "pairs_per_s": pairs / pair_s,
    "roofline": {"kernel": "prm_pairs_kernel", "bound": "valu", "achieved": pairs * instr_pair / pair_s / 1e12,
                 "peak": PEAK_T, "unit": "T lane-instructions/s (packed-f32 screen: dim per pair)", "peak_source": PEAK_SRC,
                 "frac": pairs * instr_pair / pair_s / 1e12 / PEAK_T,
                 "hbm_bytes_algorithmic": n * dim * 8},
    "query": {"status": int(st), "path_states": int(len(path)), "kernel_ms": tq[4], "bfs_ms": tq[5]},
}
# ---- CPU oracle on the first CPU_N milestones of the same stream, one core (the reference is single-threaded)
o = orc.OraclePRM(dim, sc["bounds"], R, lvs_fraction=sc["lvs_fraction"], seed=42, stream=0)
o.set_spheres(*sc["spheres"])
o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
t0 = time.perf_counter()
o.construct_roadmap(min(CPU_N, n))
cpu_dt = time.perf_counter() - t0
cn = o.num_milestones
gs, goff, gn = g.roadmap()
os_, ooff, on = o.roadmap()
same = bool(np.array_equal(gs[:cn].view(np.uint64), os_.view(np.uint64)))
# the oracle's roadmap is the sub-roadmap of the first cn milestones: every edge list restricted to < cn
sub_ok = True
if cn == n:
    sub_ok = bool(np.array_equal(goff, ooff) and np.array_equal(gn, on))   # the whole roadmap, bit for bit
else:
    for i in (0, 1, cn // 2, cn - 1):
        seg = gn[int(goff[i]):int(goff[i + 1])]
        sub_ok = sub_ok and np.array_equal(seg[seg < cn], on[int(ooff[i]):int(ooff[i + 1])])
out["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": ("the whole workload" if cn == n else "first %d milestones of the same stream" % cn),
                       "construct_s": cpu_dt, "milestones_per_s": cn / cpu_dt, "pairs_per_s": cn * (cn - 1) / 2 / cpu_dt,
                       "states_identical": same, "edges_identical": bool(sub_ok)}
# the CPU cost is quadratic in the milestone count: extrapolate a partial sample by its pair count
out["speedup_vs_cpu_1core"] = cpu_dt * (n * (n - 1.0)) / (cn * (cn - 1.0)) / (float(np.mean(walls)) * 1e-3)
print(json.dumps(out))
