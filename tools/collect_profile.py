#!/usr/bin/env python3
"""Condense what tools/profile_bench.sh left under gpurun_out/<src> into profiles/<dst>/ (the tracked summaries):
kernel_stats.csv, the bench kernel's rows of the kernel trace, its PMC rows, the in-kernel stamps and the profiled
bench line; prints the per-launch figures the README / r1_traffic.json quote.
    python tools/collect_profile.py prof32 r1_resident32 resident32"""
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
pat = sys.argv[3]
os.makedirs(dst, exist_ok=True)


def one(globpat):
    return glob.glob(os.path.join(src, globpat))[0]


shutil.copy(one("trace/*/*_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
rows = list(csv.reader(open(one("trace/*/*_kernel_trace.csv"))))
hdr = rows[0]
ki, s_i, e_i = hdr.index("Kernel_Name"), hdr.index("Start_Timestamp"), hdr.index("End_Timestamp")
keep = [r for r in rows[1:] if pat in r[ki]]
with open(os.path.join(dst, "kernel_trace_rrt.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr + ["Duration_ms"])
    for r in keep:
        w.writerow(r + ["%.4f" % ((int(r[e_i]) - int(r[s_i])) / 1e6)])
print("launch durations (ms):", ["%.3f" % ((int(r[e_i]) - int(r[s_i])) / 1e6) for r in keep])
summary = {}
for name, out in (("pmc_fetch", "pmc_FETCH_SIZE.csv"), ("pmc_write", "pmc_WRITE_SIZE.csv"), ("pmc_sq", "pmc_SQ_valu.csv")):
    rr = list(csv.DictReader(open(one(name + "/*/*_counter_collection.csv"))))
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
for c, v in summary.items():
    if len(v) >= 3:
        print(c, "warm-up %.6g; grow launch %.6g; steady launches avg %.6g" % (v[0], v[1], sum(v[2:]) / (len(v) - 2)))
    else:
        print(c, v)
shutil.copy(os.path.join(src, "inkernel_stamps.txt"), os.path.join(dst, "inkernel_stamps.txt"))
line = open(os.path.join(src, "bench_line_profiled.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, "bench_line_profiled.json"), "w").write(line + "\n")
