#!/usr/bin/env python3
"""Single-problem latency of the drop-in: the trait surface of the reference is ONE problem per
`Planner::solve` call (oxmpl/src/base/planner.rs:48-60), and one problem is one workgroup = one CU of 256.

  leg A  configs[0] (README quick-start, R^2, disc obstacle): oxmpl_amd.geometric.RRT(...).setup(checker).solve(5.0) -- wall time
         of create + setup + solve + path read-back, per seed, next to the CPU oracle solving the same stream on one core.
  leg B  configs[1]'s scene (R^3, 64 spheres), ONE problem grown to 10,000 nodes (stop_at_goal off): create + setup + solve wall,
         kernel time, and the oracle on one core.
  leg C  the same for P = 1, 16, 64, 256 problems per call (what batching buys while the chip is still filling).

Every GPU result is compared with the oracle's (path bits / checksum); a mismatch aborts.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oxmpl_amd import capi, scenarios  # noqa: E402
from oxmpl_amd.base import ProblemDefinition, RealVectorState, RealVectorStateSpace, SphereBoxValidityChecker  # noqa: E402
from oxmpl_amd.geometric import RRT  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402


class BallGoal:
    def __init__(self, centre, radius):
        self.target, self.radius = RealVectorState(centre), radius


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def leg_a(seeds):
    sc = scenarios.config1()
    rows = []
    for seed in seeds:
        t0 = time.perf_counter()
        space = RealVectorStateSpace(2, sc["bounds"])
        pd = ProblemDefinition.from_real_vector(space, RealVectorState(sc["start"]), BallGoal(sc["goal_centre"], sc["goal_radius"]))
        planner = RRT(sc["max_distance"], sc["goal_bias"], pd, seed=seed)
        t1 = time.perf_counter()
        planner.setup(SphereBoxValidityChecker(spheres=[(c, r) for c, r in zip(*sc["spheres"])]))
        t2 = time.perf_counter()
        path = planner.solve(5.0)
        t3 = time.perf_counter()
        kernel_ms = planner._batch.last_timing()["kernel_ms"]
        c = planner._batch.counts()
        o = orc.OracleRRT(2, sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"], 10000, True, seed, 0)
        o.set_spheres(*sc["spheres"])
        o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
        t4 = time.perf_counter()
        o.solve(1 << 40)
        t5 = time.perf_counter()
        gp = np.array([s.values for s in path.states])
        assert np.array_equal(bits(gp), bits(o.path())) and int(c["checksum"][0]) == o.checksum, "GPU != oracle (config 1, seed %d)" % seed
        rows.append(dict(seed=seed, iterations=int(c["iterations"][0]), nodes=int(c["nodes"][0]), path_states=len(path),
                         gpu_wall_ms=(t3 - t0) * 1e3, gpu_create_ms=(t1 - t0) * 1e3, gpu_setup_ms=(t2 - t1) * 1e3,
                         gpu_solve_ms=(t3 - t2) * 1e3, gpu_kernel_ms=kernel_ms, cpu_1core_solve_ms=(t5 - t4) * 1e3))
    return rows


KERNEL = int(sys.argv[sys.argv.index("--kernel") + 1]) if "--kernel" in sys.argv else capi.KERNEL_AUTO   # (leg B / C only)


def leg_b(P, nodes=10000):
    sc = scenarios.config2()
    t0 = time.perf_counter()
    g = scenarios.make_batch(sc, P, nodes, False, 42, 0, 0, KERNEL)
    t1 = time.perf_counter()
    g.solve(10 ** 7)
    t2 = time.perf_counter()
    tm = g.last_timing()
    c = g.counts()
    assert (c["nodes"] == nodes).all()
    o = orc.OracleRRT(3, sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"], nodes, False, 42, P - 1)
    o.set_spheres(*sc["spheres"])
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    t3 = time.perf_counter()
    o.solve(10 ** 7)
    t4 = time.perf_counter()
    assert int(c["checksum"][P - 1]) == o.checksum and int(c["iterations"][P - 1]) == o.iterations, "GPU != oracle (config 2, P = %d)" % P
    its = int(c["iterations"].sum())
    g.close()
    return dict(problems=P, nodes=nodes, iterations=its, gpu_create_setup_ms=(t1 - t0) * 1e3, gpu_solve_wall_ms=(t2 - t1) * 1e3,
                gpu_kernel_ms=tm["kernel_ms"], launches=tm["launches"], kernel={1: "stream", 2: "resident", 5: "lanes", 6: "cells"}[tm["kernel"]],
                gpu_iterations_per_s=its / (t2 - t1), cpu_1core_one_problem_ms=(t4 - t3) * 1e3,
                cpu_1core_iterations_per_s=o.iterations / (t4 - t3),
                gpu_over_one_core=(its / (t2 - t1)) / (o.iterations / (t4 - t3)))


def main():
    # one throw-away planner first: the first launch of a process loads the code object (tens of milliseconds)
    w = scenarios.make_batch(scenarios.config2(), 1, 2000, False, 1, 0, 0, capi.KERNEL_AUTO)
    w.solve(64)
    w.close()
    out = {"what": "single-problem latency of the drop-in: one Planner::solve call = one workgroup = one CU of 256",
           "config1_readme_scene_mirror_api": leg_a(range(6)),
           "config2_scene_to_10000_nodes": [leg_b(P) for P in (1, 16, 64, 256)]}
    a = out["config1_readme_scene_mirror_api"]
    out["summary"] = {
        "config1_gpu_wall_ms_median": float(np.median([r["gpu_wall_ms"] for r in a])),
        "config1_gpu_kernel_ms_median": float(np.median([r["gpu_kernel_ms"] for r in a])),
        "config1_cpu_1core_ms_median": float(np.median([r["cpu_1core_solve_ms"] for r in a])),
        "config2_P1_gpu_wall_ms": out["config2_scene_to_10000_nodes"][0]["gpu_solve_wall_ms"],
        "config2_P1_cpu_1core_ms": out["config2_scene_to_10000_nodes"][0]["cpu_1core_one_problem_ms"],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
