#!/usr/bin/env python3
"""Condense what tools/profile_cells.sh left under gpurun_out/<src> into profiles/<dst>/ (the tracked summaries) and
profiles/r3_traffic.json (the per-launch constants bench.py quotes for the cell-grid kernel):
    python tools/collect_profile_cells.py p1 r3_cells"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1])
dst = os.path.join(ROOT, "profiles", sys.argv[2])
pat = "rrt_cells_kernel"
os.makedirs(dst, exist_ok=True)


def one(globpat):
    return glob.glob(os.path.join(src, globpat))[0]


shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
rows = list(csv.reader(open(one("trace/*/*_kernel_trace.csv"))))
hdr = rows[0]
ki, s_i, e_i = hdr.index("Kernel_Name"), hdr.index("Start_Timestamp"), hdr.index("End_Timestamp")
keep = [r for r in rows[1:] if "cells" in r[ki]]
with open(os.path.join(dst, "kernel_trace_rrt.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr + ["Duration_ms"])
    for r in keep:
        w.writerow(r + ["%.4f" % ((int(r[e_i]) - int(r[s_i])) / 1e6)])
main_ms = [(int(r[e_i]) - int(r[s_i])) / 1e6 for r in keep if pat in r[ki]]
prep_ms = [(int(r[e_i]) - int(r[s_i])) / 1e6 for r in keep if "cells_prepare" in r[ki]]
print("rrt_cells_kernel launches (ms):", ["%.3f" % v for v in main_ms])
print("cells_prepare_kernel launches (ms):", ["%.3f" % v for v in prep_ms])
summary = {}
for name, out in (("pmc_fetch", "pmc_FETCH_SIZE.csv"), ("pmc_write", "pmc_WRITE_SIZE.csv"), ("pmc_sq", "pmc_SQ_valu.csv"), ("pmc_sq2", "pmc_SQ_mem.csv")):
    try:
        rr = list(csv.DictReader(open(one(name + "/*/*_counter_collection.csv"))))
    except IndexError:
        continue
    agg = collections.OrderedDict()
    for r in rr:
        if pat in r["Kernel_Name"]:
            k = (int(r["Dispatch_Id"]), r["Counter_Name"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel", "Counter_Name", "Counter_Value_summed_over_XCDs"])
        for (d, c), v in agg.items():
            w.writerow([d, pat, c, v])
            summary.setdefault(c, []).append(v)
# dispatches in launch order: bench.py's small warm-up launch, the grow launch, then the steady launches
st = {}
for c, v in summary.items():
    print(c, "warm-up %.6g; grow launch %.6g; steady launches avg %.6g" % (v[0], v[1], sum(v[2:]) / (len(v) - 2)))
    st[c] = sum(v[2:]) / (len(v) - 2)
shutil.copy(os.path.join(src, "inkernel_stamps.txt"), os.path.join(dst, "inkernel_stamps.txt"))
line = open(os.path.join(src, "bench_line_profiled.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, "bench_line_profiled.json"), "w").write(line + "\n")
# the steady launch's per-launch constants.  SQ_BUSY_CYCLES sums one counter per shader engine (32 on this chip): / 32 = the
# launch's length in shader clocks; SQ_ACTIVE_INST_VALU counts quad-cycles (x 4) over all SIMDs (1024).
steady_ms = sum(main_ms[2:]) / max(1, len(main_ms) - 2)
cycles = st["SQ_BUSY_CYCLES"] / 32.0
t = {
    "_note": "per-launch figures of the steady@10k launch (1024 problems x 4096 iterations) of rrt_cells_kernel<3,false>: rocprofv3 --pmc, "
             "separate runs per counter set (tools/profile_cells.sh), summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB and x 2 on gfx950 "
             "(guide); Infinity-Cache hits included.  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32). Source rows: "
             "profiles/%s/pmc_*.csv" % sys.argv[2],
    "cells": {
        "fetch_size_kib_per_launch": st["FETCH_SIZE"], "write_size_kib_per_launch": st["WRITE_SIZE"],
        "hbm_bytes_per_launch": int((st["FETCH_SIZE"] + st["WRITE_SIZE"]) * 1024 * 2),
        "algorithmic_bytes_per_launch": 1024 * 4096 * 10000 * 3 * 8,
        "insts_valu_per_launch": st["SQ_INSTS_VALU"], "insts_salu_per_launch": st["SQ_INSTS_SALU"],
        "active_inst_valu_quadcycles": st["SQ_ACTIVE_INST_VALU"], "busy_cycles_sum_over_32_shader_engines": st["SQ_BUSY_CYCLES"],
        "shader_cycles_per_launch": cycles, "valu_busy": st["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
        "valu_insts_per_iteration": st["SQ_INSTS_VALU"] / (1024.0 * 4096.0),
        "kernel_ms_under_profiler": steady_ms, "shader_clock_GHz": cycles / (steady_ms * 1e6),
    },
}
json.dump(t, open(os.path.join(ROOT, "profiles", "r3_traffic.json"), "w"), indent=1)
print(json.dumps(t["cells"], indent=1))
