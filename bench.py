#!/usr/bin/env python3
"""bench.py -- RRT iterations/sec (batched problems), R^3, 10k-node trees, 64-sphere field.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], one GPU; configs[2] = the same, 1024 problems per rank):
1024 independent planning problems per GPU, trees pre-grown (untimed) to 10,000 nodes; one
"step" = every problem runs ITERS further RRT iterations with inserts suppressed (freeze), so
every nearest-neighbour scan covers exactly 10,000 nodes: sample -> NN -> steer -> motion
check (6 states x 64 spheres) per iteration, nothing skipped but the push.  Inputs are resident
in HBM before the timed region; problems are sharded across ranks with no data-path collective
(weak scaling); torch.distributed (RCCL) is used for the barriers and the throughput gather.

One JSON line on rank 0.  `roofline` prices the grow kernel against the HBM roofline with
ALGORITHMIC bytes (24 B x tree size per iteration); `cpu_baseline` times the CPU
oracle (our C restatement of oxmpl's loop, kind "port") on a bounded sample of the same
workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES = 10000
BYTES_PER_ITER = N_NODES * 3 * 8          # B(n) = n * d * 8
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
# measured on this chip with tools/scan_bench.hip: the bare register scan (10,240 nodes, tie detector
# included, no reduce / resolve) sustains 1.0 us per CU whatever the wave geometry -> 256 CUs
VALU_SCAN_CEILING_ITS = 258.0e6
# the binary32-screen kernel (rrt_resident32.hip): 6 VALU instructions per (register row of 64 nodes, query) -- three packed
# f32 ops per query pair halved, and_or + med3 + min -- over 160 rows, 4 cycles per wave64 instruction, 4 SIMDs per CU,
# 2.4 GHz, 256 CUs: an instruction-count bound for the scan alone (no reduce, no resolver)
VALU_SCREEN_CEILING_ITS = 256 * 4 * 2.4e9 / (160 * 6 * 4)


def measured_traffic(kernel_name, iters_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC run (profiles/r1_traffic.json), valid
    for the profiled shape only (1024 problems x 4096 iterations); None otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
            t = json.load(f)[kernel_name]
        if iters_per_launch * BYTES_PER_ITER == t["algorithmic_bytes_per_launch"]:
            return t["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(sc, seed, threads, iters):
    """Oracle (kind 'port') on `threads` host cores: `threads` problems grown to 10k nodes
    (untimed), then `iters` frozen iterations each (timed) -- same per-iteration work as the GPU step."""
    from oracle import oracle_py as orc
    planners = []
    for p in range(threads):
        o = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                          N_NODES, False, seed, p)
        o.set_spheres(*sc["spheres"])
        o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
        planners.append(o)
    orc.solve_many(planners, 10 ** 7, threads=threads)
    assert all(p.num_nodes == N_NODES for p in planners)
    t0 = time.perf_counter()
    orc.solve_many(planners, iters, freeze=True, threads=threads)
    dt = time.perf_counter() - t0
    # the reference is single-threaded per planner: one problem on one core
    t1 = time.perf_counter()
    orc.solve_many(planners[:1], iters, freeze=True, threads=1)
    dt1 = time.perf_counter() - t1
    return planners, dict(value=threads * iters / dt, unit="iterations/s", cores=threads, kind="port",
                          value_1core=iters / dt1,
                          sample="%d problems x %d frozen iterations at n=10000 on %d threads "
                                 "(oracle/rrt_oracle.c, C restatement of rrt.rs:170-225)" % (threads, iters, threads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--problems", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--strong-total", type=int, default=0,
                    help="strong scaling: this many problems in total, divided over the ranks (SURVEY.md 8(d): 8192)")
    ap.add_argument("--iters", type=int, default=4096, help="RRT iterations per problem per step")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 stream, 2 resident, 3 resident + pruned scan, 4 resident + binary32 screen")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary stream-kernel measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    import torch  # plumbing only: device sync + torch.distributed (RCCL) barriers / gather
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import numpy as np
    from oxmpl_amd import capi, scenarios, sharding

    sc = scenarios.config2()
    if args.strong_total:
        if args.strong_total % world:
            raise SystemExit("--strong-total must be a multiple of the rank count")
        args.problems = args.strong_total // world
    P, seed = args.problems, 42
    first_id, _ = sharding.problem_range(rank, P)
    gpu = scenarios.make_batch(sc, P, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id,
                               device=local_rank, kernel=args.kernel)
    # untimed: grow every tree to 10,000 nodes (also the "grow" figure reported below)
    barrier()
    t0 = time.perf_counter()
    gpu.solve(10 ** 7)
    torch.cuda.synchronize()
    grow_s = time.perf_counter() - t0
    grow_t = gpu.last_timing()
    c = gpu.counts()
    assert (c["nodes"] == N_NODES).all()
    grow_iters = int(c["iterations"].sum())

    for _ in range(args.warmup):
        gpu.solve(args.iters, freeze=True)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        gpu.solve(args.iters, freeze=True)
        t = gpu.last_timing()           # HIP events on the library's own stream
        kernel_ms += t["kernel_ms"]
        launches += t["launches"]
    barrier()
    dt = time.perf_counter() - t0
    c2 = gpu.counts()
    done = int((c2["iterations"] - c["iterations"]).sum())
    assert done == P * args.iters * (args.steps + args.warmup)
    iters_timed = P * args.iters * args.steps

    kname = {1: "stream", 2: "resident", 3: "pruned", 4: "resident_f32"}[gpu.last_timing()["kernel"]]
    # secondary (rank 0, N=1 only): the HBM-streaming kernel on the same workload, 2 steps
    secondary = None
    if world == 1 and not args.no_secondary and kname != "stream":
        g2 = scenarios.make_batch(sc, P, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id,
                                  device=local_rank, kernel=capi.KERNEL_STREAM)
        g2.solve(10 ** 7)
        g2.solve(args.iters, freeze=True)
        ms2 = 0.0
        for _ in range(2):
            g2.solve(args.iters, freeze=True)
            ms2 += g2.last_timing()["kernel_ms"]
        ach2 = P * args.iters * BYTES_PER_ITER / (ms2 / 2 * 1e-3) / 1e9
        secondary = {"kernel": "stream", "iterations_per_s": P * args.iters / (ms2 / 2 * 1e-3), "kernel_avg_ms": ms2 / 2,
                     "roofline": {"bound": "hbm", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach2 / HBM_PEAK_GBS, "traffic": measured_traffic("stream", P * args.iters)}}
        g2.close()

    # RCCL all-gather over xGMI (nccl backend): throughput report only, no data-path collective
    allst = sharding.gather_stats([dt, float(iters_timed), kernel_ms, float(launches), float(grow_iters), grow_s,
                                   grow_t["kernel_ms"]], device="cuda")

    if rank == 0:
        agg = sharding.aggregate(allst)
        t_max, value = agg["t_max"], agg["value"]
        avg_launch_ms = float(allst[0, 2] / allst[0, 3])
        iters_per_launch = P * args.iters
        achieved = iters_per_launch * BYTES_PER_ITER / (avg_launch_ms * 1e-3) / 1e9
        out = {
            "metric": "RRT iterations/sec (batched problems), R^3 10k-node tree, 64-sphere field",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong_total else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: R^3 RRT, 64 random spheres, %d problem instances per MI355X, "
                                   "steady@10k (trees pre-grown to 10000 nodes, inserts suppressed)" % P,
                       "problems_per_gpu": P, "iterations_per_problem_per_step": args.iters, "tree_nodes": N_NODES,
                       "spheres": 64, "max_distance": 0.5, "goal_bias": 0.05, "parallelism": "problem-parallel x%d" % world,
                       "kernel": kname,
                       # what `dtype` means here: every value that enters a result (distances, steer, motion check, tree,
                       # checksum) is computed in f64 in the reference's evaluation order; the resident_f32 / stream kernels
                       # additionally SCREEN nearest-neighbour candidates in packed binary32 with a proven error bound and
                       # fall back to the f64 scan when the screen cannot decide -- bit-identical results
                       "arithmetic": ("f64 results; packed-f32 candidate screen + f64 decision" if kname in ("resident_f32", "stream")
                                      else "f64 throughout")},
            # bound "hbm" = the roofline of any design that re-reads the tree per iteration (33.3 M it/s);
            # the resident kernels keep the tree in VGPRs, so frac > 1 and their own bound is VALU issue
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kname, iters_per_launch),
                         "kernel_avg_ms": avg_launch_ms, "algorithmic_bytes_per_launch": iters_per_launch * BYTES_PER_ITER,
                         "valu": ({"bound": "packed-f32 screen, VALU instruction count (bench.py)", "peak_iterations_per_s": VALU_SCREEN_CEILING_ITS,
                                   "frac": (iters_per_launch / (avg_launch_ms * 1e-3)) / VALU_SCREEN_CEILING_ITS}
                                  if kname == "resident_f32" else
                                  {"bound": "f64-valu scan (tools/scan_bench.hip, measured)", "peak_iterations_per_s": VALU_SCAN_CEILING_ITS,
                                   "frac": (iters_per_launch / (avg_launch_ms * 1e-3)) / VALU_SCAN_CEILING_ITS})},
            "grow": {"iterations": float(allst[:, 4].sum()), "wall_s": float(allst[:, 5].max()),
                     "iterations_per_s": float(allst[:, 4].sum() / allst[:, 5].max()),
                     "kernel_ms_rank0": float(allst[0, 6])},
        }
        if secondary is not None:
            out["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:  # the contract: rank 0, N = 1 only
            threads = min(os.cpu_count() or 1, 16)
            planners, base = cpu_baseline(sc, seed, threads, 60000)   # ~25 s of CPU work on 16 threads
            out["cpu_baseline"] = base
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gpu.close()


if __name__ == "__main__":
    main()
