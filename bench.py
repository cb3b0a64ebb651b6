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

One JSON line on rank 0 (flat objects only: no nested dictionaries inside `roofline` etc.).

`roofline`: the bound that binds the kernel that ran.  rrt_cells.hip (what KERNEL_AUTO runs here) finds the nearest node
through an exact cell grid -- ~6 cell blocks of 64 B per query instead of the 240 KB SURVEY.md 8(d) charges a scan -- so it is
bound by vector-ALU issue under memory latency: `frac` = the VALU busy fraction of the steady launch (performance counters of a
committed profile of this shape, profiles/r3_traffic.json), `achieved` = iterations/s from HIP events, `peak` = achieved / frac,
`traffic` / `hbm_moved_GBps` = the bytes it really moves.  rrt_lanes.hip (round 2's default, reported as `secondary_lanes`) keeps
the tree in the register file: its `peak` is the measured rate of its own screen loop (profiles/r2_valu_peak.json).  The byte
figure SURVEY.md 8(d) defines (24 B x tree size per iteration) is kept as `hbm_equivalent_GBps`; for the stream kernel it is the
bound itself.

`cpu_baseline` times the CPU oracle (our C restatement of oxmpl's loop, kind "port") on a bounded sample of
the same workload on this host's cores -- and the run is only accepted if the GPU's per-problem node
counts, iteration counts and checksums (every iteration folds nearest index, q_new bits and verdict) equal
the oracle's on that sample, after the grow phase and after the frozen iterations.

`secondary` / `secondary_lanes` / `secondary_f64`: the same steps through the stream kernel, the lane-per-query kernel and the
all-binary64 resident kernel (checksums of all 1024 problems must equal the main run's).  `secondary_rrt_star`: the RRT* row on the same scene and batch size, 1024 trees grown
to 10,000 nodes; the checker grows problem 0 with the CPU oracle as well and refuses the line unless parents after rewiring,
costs and checksum are identical.  `secondary_rrt_connect`: the RRTConnect rows -- BASELINE.json configs[3], SE(2) among
256 segments, and the R^3 scene of this benchmark -- 1024 problems each, solved to completion by one launch; two problems per row are
solved by the CPU oracle as well (both trees, checksum, iterations, merged path identical, or no line).
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
KNAME = {1: "stream", 2: "resident", 5: "lanes", 6: "cells"}
ROWS_PER_ITERATION = 10240 // 64          # register rows (64 nodes each) one query is screened against
CUS = 256


def valu_peak(kname):
    """Measured VALU ceiling of the kernel's scan on this chip, iterations/s, and where it comes from.
    lanes: the bare screen loop of the kernel in (row, query) pairs per second, divided by the 160 rows of a 10,240-slot tree.
    resident (binary64 scanners): tools/scan_bench.hip, 1.0 us per 10k-node scan per CU (profiles/r1_resident)."""
    if kname == "lanes":
        # the dot-product screen: D packed fused multiply-adds per PAIR of 64-node register rows and query + one v_min3_f32
        # (2.0 instructions per row and query in R^3), 24 rows, two waves per SIMD as in the kernel; a 10,000-node tree occupies
        # 160 rows across the eight scanner waves (blocks of four rows)
        try:
            with open(os.path.join(ROOT, "profiles", "r2_valu_peak.json")) as f:
                d = json.load(f)
            for e in d["dot_screen_min3"]:
                if e["dim"] == 3 and e["rows"] == 24 and e["waves_per_simd"] == 2 and e["queries_per_pass"] == 8:
                    return e["row_queries_per_s_chip"] / ROWS_PER_ITERATION, \
                        "profiles/r2_valu_peak.json: dot_screen_min3, R^3, 24 rows, 2 waves/SIMD, / 160 rows per iteration"
        except (OSError, KeyError, ValueError):
            pass
        return None, "profiles/r2_valu_peak.json has no dot_screen_min3 entry"
    if kname == "resident":
        return 258.0e6, "tools/scan_bench.hip: 1.0 us per 10k-node binary64 scan per CU x 256 CUs"
    return None, None


def measured_traffic(kernel_name, iters_per_launch):
    """HBM bytes per launch from a committed rocprofv3 PMC run of this shape (1024 problems x 4096 iterations);
    (None, None) for any other shape.  A constant from a file, not a measurement of this run: `traffic_source` says so."""
    for name in ("r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)[kernel_name]
            if iters_per_launch * BYTES_PER_ITER == t["algorithmic_bytes_per_launch"]:
                return t["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of this shape; not this run)" % name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def cells_pmc():
    """The cell-grid kernel's steady launch as the performance counters saw it (profiles/r3_traffic.json, produced by
    tools/profile_cells.sh + tools/collect_profile_cells.py): VALU busy fraction, VALU instructions per iteration."""
    try:
        with open(os.path.join(ROOT, "profiles", "r3_traffic.json")) as f:
            return json.load(f)["cells"]
    except (OSError, KeyError, ValueError):
        return None


def copy_peak_gbs(torch, device):
    """device-to-device copy bandwidth of this GPU (read + write bytes per second), the measured counterpart of the
    8 TB/s vendor figure (SURVEY.md 8(d) asks for both)."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(5):
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del a, b
    return 2.0 * n / (best * 1e-3) / 1e9


def oracle_planners(sc, seed, first_id, count):
    from oracle import oracle_py as orc
    planners = []
    for p in range(count):
        o = orc.OracleRRT(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"],
                          N_NODES, False, seed, first_id + p)
        o.set_spheres(*sc["spheres"])
        o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
        planners.append(o)
    return orc, planners


def host_cores():
    """(cores the machine has, cores this process may use): os.cpu_count(), then the scheduler affinity mask and the cgroup v2 /
    v1 CPU quota, whichever is smaller -- a container that is given 16 of 256 cores reports (256, 16)."""
    total = os.cpu_count() or 1
    usable = total
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as f:
                txt = f.read().strip()
            if parse is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    quota, period = txt, f.read().strip()
            else:
                quota, period = parse(txt)
            if quota not in ("max", "-1") and int(quota) > 0:
                usable = min(usable, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return total, max(1, usable)


def oracle_check(sc, seed, first_id, count, threads, frozen_iters, gpu_after_grow, gpu_after_frozen, offset=0):
    """`count` problems of this rank's shard (local indices offset .. offset + count - 1) through the oracle: grown to 10k nodes,
    then `frozen_iters` frozen iterations -- node counts, iteration counts and per-iteration checksums must equal the GPU's at
    both points, or the run is refused.  Returns (planners, seconds of the frozen leg)."""
    orc, planners = oracle_planners(sc, seed, first_id + offset, count)
    orc.solve_many(planners, 10 ** 7, threads=threads)
    for i, o in enumerate(planners):
        p = offset + i
        if not (o.num_nodes == N_NODES == int(gpu_after_grow["nodes"][p]) and o.iterations == int(gpu_after_grow["iterations"][p])
                and o.checksum == int(gpu_after_grow["checksum"][p])):
            raise SystemExit("bench.py: GPU != oracle after the grow phase, problem id %d: refusing to report a number" % (first_id + p))
    t0 = time.perf_counter()
    orc.solve_many(planners, frozen_iters, freeze=True, threads=threads)
    dt = time.perf_counter() - t0
    for i, o in enumerate(planners):
        p = offset + i
        if not (o.iterations == int(gpu_after_frozen["iterations"][p]) and o.checksum == int(gpu_after_frozen["checksum"][p])
                and o.num_nodes == int(gpu_after_frozen["nodes"][p])):
            raise SystemExit("bench.py: GPU != oracle after %d frozen iterations, problem id %d: refusing to report a number"
                             % (frozen_iters, first_id + p))
    return orc, planners, dt


def cpu_baseline_and_check(sc, seed, first_id, P, frozen_iters, gpu_after_grow, gpu_after_frozen):
    """Oracle (kind 'port') on ALL the cores this process may use: one problem per thread -- the reference is
    single-threaded per planner -- grown to 10k nodes (untimed), then `frozen_iters` frozen iterations each (timed): the same
    per-iteration work as the GPU step, and the SAME iterations the GPU ran for those problems (checked, see oracle_check)."""
    total, usable = host_cores()
    threads = min(usable, P)
    orc, planners, dt = oracle_check(sc, seed, first_id, threads, threads, frozen_iters, gpu_after_grow, gpu_after_frozen)
    # one problem on one core
    n1 = min(frozen_iters, 60000)
    t1 = time.perf_counter()
    orc.solve_many(planners[:1], n1, freeze=True, threads=1)
    dt1 = time.perf_counter() - t1
    return dict(value=threads * frozen_iters / dt, unit="iterations/s", cores=threads, kind="port", value_1core=n1 / dt1,
                host_cores=total, usable_cores=usable, threads=threads,
                cores_note="host_cores = os.cpu_count(); usable_cores = min(that, scheduler affinity, cgroup CPU quota); one oracle "
                           "planner per usable core",
                sample="%d problems x %d frozen iterations at n=10000 on %d threads (oracle/rrt_oracle.c, C restatement "
                       "of rrt.rs:170-225); the GPU ran the same iterations of the same problems" % (threads, frozen_iters, threads),
                verified="GPU node counts, iteration counts and per-iteration checksums == oracle for problems %d..%d after the "
                         "grow phase and after the frozen iterations" % (first_id, first_id + threads - 1))


def _se2_oracle(sc4, seed, pid):
    from oracle import oracle_py as orc
    o = orc.OracleSE2Connect(sc4["bounds"][:2], sc4["bounds"][2], sc4["max_distance"], sc4["goal_bias"], sc4["lvs_fraction"], 10000, seed, pid)
    o.set_segments(sc4["segments"], sc4["clearance"])
    o.setup(sc4["start"], sc4["goal_centre"], sc4["goal_radius"])
    return o


def _connect_oracle(sc, seed, pid):
    from oracle import oracle_py as orc
    o = orc.OracleRRTConnect(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"], 10000, seed, pid)
    o.set_spheres(*sc["spheres"])
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return o


def _connect_rows(args, scenarios, capi, sc, P, seed, first_id, device):
    """the RRTConnect rows (DESIGN.md sections 8 and 11; BASELINE.json configs[3] = SE(2) among 256 segments): 1024 problems solved to
    completion by one launch each; two problems per row are solved by the CPU oracle too -- both trees, checksum, iteration count and
    the merged path must be identical, or nothing is printed"""
    import numpy as np
    rows = {}
    sc4 = scenarios.config4()
    for name, make, make_oracle in (
            ("se2_256_segments", lambda: scenarios.make_se2_batch(sc4, P, 10000, seed, first_id, device),
             lambda pid: _se2_oracle(sc4, seed, pid)),
            ("r3_64_spheres", lambda: scenarios.make_batch(sc, P, 10000, True, seed, first_id, device, 0, capi.PLANNER_RRT_CONNECT),
             lambda pid: _connect_oracle(sc, seed, pid))):
        make().solve(10 ** 7)                       # warm-up (code object load)
        g = make()
        stc = g.solve(10 ** 7)
        assert (stc == capi.OK).all()
        c, gc, ms = g.counts(), g.goal_counts(), g.last_timing()["kernel_ms"]
        verified = None
        if not args.no_cpu_baseline:
            verified = True
            for pr in (0, P - 1):
                o = make_oracle(first_id + pr)
                o.solve(10 ** 7)
                same = (int(c["iterations"][pr]) == o.iterations and int(c["checksum"][pr]) == o.checksum
                        and int(c["nodes"][pr]) == o.num_nodes(0) and int(gc["nodes"][pr]) == o.num_nodes(1))
                for w, (gs, gp) in enumerate((g.tree(pr), g.goal_tree(pr))):
                    os_, op = o.tree(w)
                    same = same and np.array_equal(gs.view(np.uint64), os_.view(np.uint64)) and np.array_equal(gp, op)
                gpath, opath = g.path(pr), o.path()
                same = same and gpath.shape == opath.shape and np.array_equal(gpath.view(np.uint64), opath.view(np.uint64))
                if not same:
                    raise SystemExit("bench.py: RRTConnect (%s) != oracle on problem %d: refusing to report" % (name, first_id + pr))
        its = int(c["iterations"].sum())
        rows[name] = {"planner": "RRTConnect (rrt_connect.rs), %d problems to completion" % P, "kernel_ms": ms,
                      "problems_per_s": P / (ms * 1e-3), "iterations": its, "iterations_per_s": its / (ms * 1e-3),
                      "slowest_problem_iterations": int(c["iterations"].max()),
                      "us_per_iteration_of_the_slowest_problem": ms * 1e3 / float(c["iterations"].max()),
                      "mean_nodes_both_trees": float((c["nodes"] + gc["nodes"]).mean()),
                      "two_problems_equal_oracle": verified}
        g.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--problems", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--strong-total", type=int, default=0,
                    help="strong scaling: this many problems in total, divided over the ranks (SURVEY.md 8(d): 8192)")
    ap.add_argument("--iters", type=int, default=4096, help="RRT iterations per problem per step")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 stream, 2 resident (binary64 scanners), 4 resident + binary32 screen (lane groups), "
                                                          "5 resident + binary32 dot-product screen, lane-per-query resolver")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for the barriers and the one all-gather (nccl = RCCL over xGMI; gloo: rehearsals "
                         "of the N > 1 branch on a box with fewer GPUs than ranks)")
    ap.add_argument("--device-map", default="", help="comma-separated HIP device per local rank (default: local rank r -> device r); "
                                                     "'0,0' runs two ranks on GPU 0")
    ap.add_argument("--rank-check", type=int, default=2, help="N > 1: problems of its own shard every rank checks against the oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (stream kernel, all-binary64 resident kernel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    device = local_rank
    if args.device_map:
        dm = [int(v) for v in args.device_map.split(",")]
        if local_rank >= len(dm):
            raise SystemExit("--device-map names %d devices but local rank %d exists" % (len(dm), local_rank))
        device = dm[local_rank]

    import torch  # plumbing only: device sync + torch.distributed (RCCL) barriers / gather
    import torch.distributed as dist
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from oxmpl_amd import capi, scenarios, sharding

    sc = scenarios.config2()
    if args.strong_total:
        if args.strong_total % world:
            raise SystemExit("--strong-total must be a multiple of the rank count")
        args.problems = args.strong_total // world
    P, seed = args.problems, 42
    first_id, _ = sharding.problem_range(rank, P)
    gpu = scenarios.make_batch(sc, P, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id,
                               device=device, kernel=args.kernel)
    # one tiny launch first: the first launch of a process pays for loading the code object (milliseconds, inside the
    # HIP-event bracket of whatever runs first); the "grow" figure below is the second launch of the process
    warm = scenarios.make_batch(sc, 4, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id, device=device,
                                kernel=args.kernel)
    warm.solve(256)
    warm.close()
    # untimed: grow every tree to 10,000 nodes (also the "grow" figure reported below)
    barrier()
    t0 = time.perf_counter()
    gpu.solve(10 ** 7)
    torch.cuda.synchronize()
    grow_s = time.perf_counter() - t0
    grow_t = gpu.last_timing()
    grow_kname = KNAME[grow_t["kernel"]]
    c = gpu.counts()
    assert (c["nodes"] == N_NODES).all()
    grow_iters = int(c["iterations"].sum())

    snap = None   # (frozen iterations per problem, counters) at a point the CPU check can afford to reach
    for w in range(args.warmup):
        gpu.solve(args.iters, freeze=True)
        if not args.no_cpu_baseline and (w + 1) * args.iters <= 400000:
            snap = ((w + 1) * args.iters, gpu.counts())   # untimed: warm-up
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
    kname = KNAME[gpu.last_timing()["kernel"]]

    def side_run(kernel, steps=2):
        """the same workload through another kernel kind (rank 0, N = 1 only)"""
        g2 = scenarios.make_batch(sc, P, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id,
                                  device=device, kernel=kernel)
        g2.solve(10 ** 7)
        total = args.steps + args.warmup          # as many frozen steps as the main run, the last `steps` of them timed
        steps = min(steps, total)
        ms2 = 0.0
        for i in range(total):
            g2.solve(args.iters, freeze=True)
            if i >= total - steps:
                ms2 += g2.last_timing()["kernel_ms"]
        cc = g2.counts()
        g2.close()
        # same problems, same streams, same number of iterations: every checksum must equal the main run's
        same = bool((cc["checksum"] == c2["checksum"]).all() and (cc["iterations"] == c2["iterations"]).all())
        if not same:
            raise SystemExit("bench.py: kernel kind %d disagrees with the main run's checksums: refusing to report" % kernel)
        return ms2 / steps, same

    secondary_stream = secondary_f64 = secondary_lanes = None
    copy_gbs = None
    if world == 1 and not args.no_secondary:
        copy_gbs = copy_peak_gbs(torch, torch.device("cuda", device))
        if kname != "stream":
            ms, same = side_run(capi.KERNEL_STREAM)
            ach = P * args.iters * BYTES_PER_ITER / (ms * 1e-3) / 1e9
            tr, src = measured_traffic("stream", P * args.iters)
            secondary_stream = {"kernel": "stream", "iterations_per_s": P * args.iters / (ms * 1e-3), "kernel_avg_ms": ms,
                                "bound": "hbm", "algorithmic_GBps": ach, "frac_of_vendor_peak_algorithmic": ach / HBM_PEAK_GBS,
                                "traffic": tr, "traffic_source": src,
                                # what the kernel really moves: the fl32 shadow (0.5 x algorithmic) + winners; at 1024 problems
                                # the 126 MB shadow fits the 256 MB Infinity Cache, so this rate is MALL-served, not HBM
                                "moved_GBps": (tr / (ms * 1e-3) / 1e9) if tr else None,
                                "served_from": "Infinity Cache (126 MB shadow < 256 MB MALL) -- not an HBM figure",
                                "checksums_equal_main_run": same}
        if kname != "lanes":
            try:
                ms, same = side_run(capi.KERNEL_LANES)
                pk, pk_src = valu_peak("lanes")
                secondary_lanes = {"kernel": "lanes (round 2's default: register-resident brute-force screen, one CU per problem)",
                                   "iterations_per_s": P * args.iters / (ms * 1e-3), "kernel_avg_ms": ms, "bound": "valu",
                                   "peak_iterations_per_s": pk, "frac": (P * args.iters / (ms * 1e-3) / pk) if pk else None,
                                   "peak_source": pk_src, "checksums_equal_main_run": same}
            except capi.OxhipError:
                secondary_lanes = None
        if kname != "resident":
            try:
                ms, same = side_run(capi.KERNEL_RESIDENT)
                pk, pk_src = valu_peak("resident")
                secondary_f64 = {"kernel": "resident (binary64 scanners: no binary32 anywhere)", "dtype": "f64",
                                 "iterations_per_s": P * args.iters / (ms * 1e-3), "kernel_avg_ms": ms, "bound": "valu",
                                 "peak_iterations_per_s": pk, "frac": P * args.iters / (ms * 1e-3) / pk, "peak_source": pk_src,
                                 "checksums_equal_main_run": same}
            except capi.OxhipError:
                secondary_f64 = None

    # the RRT* row (geometry by the same kernel as above, wiring by rrt_star_wire.hip), same scene, same batch
    # size, search radius 1: 1024 trees grown to 10,000 nodes; problem 0 is grown by the CPU oracle too and its parents after
    # rewiring, costs and checksum must be identical, or nothing is printed
    secondary_star = None
    if world == 1 and not args.no_secondary:
        import numpy as np
        star = scenarios.make_batch(sc, P, N_NODES, stop_at_goal=False, seed=seed, first_problem_id=first_id, device=device,
                                    kernel=capi.KERNEL_AUTO, planner=capi.PLANNER_RRT_STAR, search_radius=1.0)
        star.solve(10 ** 9)
        st_t = star.last_timing()
        st_c = star.counts()
        assert (st_c["nodes"] == N_NODES).all()
        verified = None
        if not args.no_cpu_baseline:
            from oracle import oracle_py as orc
            o = orc.OracleRRTStar(sc["dim"], sc["bounds"], sc["max_distance"], sc["goal_bias"], 1.0, sc["lvs_fraction"], N_NODES, False,
                                  seed, first_id)
            o.set_spheres(*sc["spheres"])
            o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
            o.solve(10 ** 9)
            _, gp = star.tree(0)
            _, op = o.tree()
            verified = bool(np.array_equal(gp, op) and np.array_equal(star.costs(0).view(np.uint64), o.costs().view(np.uint64))
                            and int(st_c["checksum"][0]) == o.checksum and int(st_c["iterations"][0]) == o.iterations)
            if not verified:
                raise SystemExit("bench.py: RRT* (decoupled design) != oracle on problem %d: refusing to report" % first_id)
        secondary_star = {"planner": "RRT* (rrt_star.rs), search radius 1.0, %d trees grown 1 -> %d nodes" % (P, N_NODES),
                          "design": {capi.KERNEL_CELLS: "decoupled: rrt_cells.hip + rrt_star_wire.hip", capi.KERNEL_LANES: "decoupled: rrt_lanes.hip + rrt_star_wire.hip"}.get(st_t["kernel"], "one kernel: rrt_star.hip"),
                          "iterations": int(st_c["iterations"].sum()), "kernel_ms": st_t["kernel_ms"],
                          "iterations_per_s": float(st_c["iterations"].sum()) / (st_t["kernel_ms"] * 1e-3),
                          "problem0_parents_costs_checksum_equal_oracle": verified}
        star.close()

    secondary_connect = None
    if world == 1 and not args.no_secondary:
        secondary_connect = _connect_rows(args, scenarios, capi, sc, P, seed, first_id, device)

    def check_point():
        """(frozen iterations per problem, GPU counters there): the whole frozen run when the CPU can afford it, else the last
        warm-up boundary"""
        frozen = args.iters * (args.steps + args.warmup)
        if frozen <= 400000:
            return frozen, c2
        if snap is not None:
            return snap
        raise SystemExit("bench.py: %d frozen iterations per problem are too many for the CPU check and there is no "
                         "warm-up boundary to compare at; use --warmup >= 1 or --no-cpu-baseline" % frozen)

    # N > 1: every rank checks the LAST `rank_check` problems of its own shard against the oracle (same rule as the N = 1 run:
    # no agreement, no line) -- after the timed region, outside every bracket
    checked = 0
    if world > 1 and not args.no_cpu_baseline and args.rank_check > 0:
        k = min(args.rank_check, P)
        point = check_point()
        oracle_check(sc, seed, first_id, k, min(k, host_cores()[1]), point[0], c, point[1], offset=P - k)
        checked = k

    # one all-gather (RCCL over xGMI with the nccl backend): throughput report only, no data-path collective
    allst = sharding.gather_stats([dt, float(iters_timed), kernel_ms, float(launches), float(grow_iters), grow_s,
                                   grow_t["kernel_ms"], float(checked), float(first_id), float(device)],
                                  device="cuda" if args.backend == "nccl" else None)

    if rank == 0:
        agg = sharding.aggregate(allst)
        t_max, value = agg["t_max"], agg["value"]
        avg_launch_ms = float(allst[0, 2] / allst[0, 3])
        iters_per_launch = P * args.iters
        its = iters_per_launch / (avg_launch_ms * 1e-3)
        hbm_equiv = its * BYTES_PER_ITER / 1e9
        traffic, traffic_src = measured_traffic(kname, iters_per_launch)
        if kname == "stream":
            roofline = {"bound": "hbm", "achieved": hbm_equiv, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_equiv / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_src, "kernel_avg_ms": avg_launch_ms,
                        "algorithmic_bytes_per_launch": iters_per_launch * BYTES_PER_ITER}
        elif kname == "cells":
            # The cell-grid kernel neither re-reads the tree (SURVEY.md 8(d)'s byte model) nor scans it: per query it reads ~6 cell
            # blocks of 64 B and the winner's coordinates.  What binds it is vector-ALU issue under memory latency: `frac` = the VALU
            # busy fraction of the steady launch (performance counters of a committed profile of this shape), `peak` = the rate at
            # which this instruction stream would run with the vector ALUs never idle; the bytes it really moves are below.
            pm = cells_pmc()
            busy = pm["valu_busy"] if pm and traffic else None
            moved = (traffic / (avg_launch_ms * 1e-3) / 1e9) if traffic else None
            roofline = {"bound": "valu", "achieved": its, "peak": (its / busy) if busy else None, "unit": "iterations/s", "frac": busy,
                        "peak_source": "this run's rate / VALU busy fraction: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x shader cycles) of the steady "
                                       "launch, profiles/r3_traffic.json (rocprofv3 --pmc; not this run)",
                        "valu_insts_per_iteration": pm["valu_insts_per_iteration"] if pm else None,
                        "traffic": traffic, "traffic_source": traffic_src, "kernel_avg_ms": avg_launch_ms,
                        "hbm_moved_GBps": moved, "hbm_moved_over_vendor_peak": (moved / HBM_PEAK_GBS) if moved else None,
                        "hbm_moved_over_copy_peak": (moved / copy_gbs) if (moved and copy_gbs) else None,
                        "algorithmic_bytes_per_launch": iters_per_launch * BYTES_PER_ITER,
                        # SURVEY.md 8(d)'s byte model, for reference: a design that re-read the tree would need this HBM rate
                        "hbm_equivalent_GBps": hbm_equiv, "hbm_equivalent_over_vendor_peak": hbm_equiv / HBM_PEAK_GBS,
                        "hbm_vendor_peak_GBps": HBM_PEAK_GBS, "hbm_copy_peak_GBps": copy_gbs}
        else:
            peak, peak_src = valu_peak(kname)
            roofline = {"bound": "valu", "achieved": its, "peak": peak, "unit": "iterations/s",
                        "frac": (its / peak) if peak else None, "peak_source": peak_src,
                        "traffic": traffic, "traffic_source": traffic_src, "kernel_avg_ms": avg_launch_ms,
                        "algorithmic_bytes_per_launch": iters_per_launch * BYTES_PER_ITER,
                        # SURVEY.md 8(d)'s byte model, for reference: a design that re-read the tree would need this HBM rate
                        "hbm_equivalent_GBps": hbm_equiv, "hbm_equivalent_over_vendor_peak": hbm_equiv / HBM_PEAK_GBS,
                        "hbm_vendor_peak_GBps": HBM_PEAK_GBS, "hbm_copy_peak_GBps": copy_gbs}
        per_rank_its = (allst[:, 1] / allst[:, 0]).tolist()
        rounds = -(-P // CUS)
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
                       # (rrt_lanes.hip: one workgroup = one problem per CU at a time, P problems in ceil(P / 256) rounds; rrt_cells.hip:
                       #  one wave per problem -- per part of a problem when frozen -- so the chip fills by waves, not by CUs)
                       "workgroup_rounds": rounds if kname != "cells" else None,
                       "last_round_fill": ((P - (rounds - 1) * CUS) / CUS) if kname != "cells" else None,
                       # what `dtype` means here: every value that enters a result (distances, steer, motion check, tree,
                       # checksum) is computed in f64 in the reference's evaluation order; the lanes / stream kernels
                       # additionally SCREEN nearest-neighbour candidates in packed binary32 with a proven error bound and
                       # fall back to the f64 scan when the screen cannot decide -- bit-identical results
                       "arithmetic": ("f64 results; binary32 candidate screen (exact cell grid) + f64 decision" if kname == "cells"
                                      else "f64 results; packed-f32 candidate screen + f64 decision" if kname in ("stream", "lanes")
                                      else "f64 throughout")},
            "roofline": roofline,
            "grow": {"iterations": float(allst[:, 4].sum()), "wall_s": float(allst[:, 5].max()),
                     "iterations_per_s": float(allst[:, 4].sum() / allst[:, 5].max()),
                     "kernel_ms_rank0": float(allst[0, 6]), "kernel": grow_kname},
            "per_rank": {"iterations_per_s_min": min(per_rank_its), "iterations_per_s_max": max(per_rank_its),
                         "step_time_skew": float(allst[:, 0].max() / allst[:, 0].min()),
                         "iterations_per_s": per_rank_its, "first_problem_id": [int(v) for v in allst[:, 8]],
                         "device": [int(v) for v in allst[:, 9]], "kernel_ms_per_step": (allst[:, 2] / args.steps).tolist(),
                         # N > 1: problems of its own shard each rank compared with the oracle (grow phase + frozen iterations)
                         "problems_checked_against_oracle": [int(v) for v in allst[:, 7]]},
        }
        if world > 1:
            out["config"]["backend"] = args.backend + (" (RCCL)" if args.backend == "nccl" else " (CPU collective: a rehearsal)")
            if len(set(out["per_rank"]["device"])) < world:
                out["config"]["note"] = "ranks share a GPU (--device-map %s): a functional run of the N > 1 branch, not a scaling point" % args.device_map
        if copy_gbs is not None:
            out["hbm_copy_peak_GBps"] = copy_gbs
        if secondary_stream is not None:
            if copy_gbs and secondary_stream["moved_GBps"]:
                secondary_stream["moved_over_copy_peak"] = secondary_stream["moved_GBps"] / copy_gbs
                secondary_stream["moved_over_vendor_peak"] = secondary_stream["moved_GBps"] / HBM_PEAK_GBS
            out["secondary"] = secondary_stream
        if secondary_lanes is not None:
            out["secondary_lanes"] = secondary_lanes
        if secondary_f64 is not None:
            out["secondary_f64"] = secondary_f64
        if secondary_star is not None:
            out["secondary_rrt_star"] = secondary_star
        if secondary_connect:
            out["secondary_rrt_connect"] = secondary_connect
        if not args.no_cpu_baseline and world == 1:  # the contract: rank 0, N = 1 only
            point = check_point()
            out["cpu_baseline"] = cpu_baseline_and_check(sc, seed, first_id, P, point[0], c, point[1])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    gpu.close()


if __name__ == "__main__":
    main()
