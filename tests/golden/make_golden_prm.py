#!/usr/bin/env python3
"""Generates tests/golden/prm_golden.json: PRM roadmaps and queries computed by an independent
numpy / pure-Python restatement of oxmpl's PRM (oxmpl/src/geometric/planners/prm.rs:96-154 construct,
:161-187 check_motion, :227-307 solve with its breadth-first search, :189-208 path), on top of the
primitives of make_golden.py (ChaCha12 / rand transforms, distance, interpolate, num_steps).

Written separately from oracle/prm_oracle.c (different language, different data layout: numpy row
operations instead of per-node heap vectors) so that the two restatements pin each other.  The
reference itself cannot be run here (no cargo / rustc) and its PRM tests hold no vectors
(oxmpl/tests/prm_rvss_tests.rs asserts properties only): PARITY UNPINNED against a rustc-built oxmpl.

Run:  python tests/golden/make_golden_prm.py      (a few seconds)
"""
import json
import os
import sys
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (ChaCha12Rng, Field, check_motion, distance, hexf, random_range,  # noqa: E402
                         sphere_field)

FNV_P = 0x100000001B3
MASK = (1 << 64) - 1


def prm_construct(dim, bounds, radius, fraction, field, seed, stream, max_milestones, max_samples):
    """prm.rs:96-154 with the build-defined caps where the reference reads its wall clock."""
    rng = ChaCha12Rng(seed, stream)
    states = np.zeros((0, dim))
    edges = []
    n_samples = 0
    while len(edges) < max_milestones and n_samples < max_samples:
        q = [random_range(rng, lo, hi) for lo, hi in bounds]      # sample_uniform, rvss.rs:233-249
        n_samples += 1
        if not field.is_valid(q):                                  # prm.rs:123
            continue
        mine = []
        if len(edges):
            acc = np.zeros(len(edges))
            for k in range(dim):                                   # distance(q_rand, other), rvss.rs:137-155
                d = q[k] - states[:, k]
                acc = acc + d * d
            near = np.nonzero(np.sqrt(acc) < radius)[0]            # prm.rs:134 (strict)
            for i in near:
                if check_motion(field, bounds, fraction, q, [float(v) for v in states[i]]):  # from = new sample
                    mine.append(int(i))
        new_idx = len(edges)
        edges.append(mine)
        for i in mine:
            edges[i].append(new_idx)                               # prm.rs:143-145
        states = np.vstack([states, np.array(q)[None, :]])
    return dict(states=states, edges=edges, n_samples=n_samples, draws=rng.draws if hasattr(rng, "draws") else None)


def prm_solve(dim, bounds, radius, fraction, field, rm, start, goal_c, goal_r):
    """prm.rs:227-307; returns (status, start_connections, goal_indices, path)"""
    states, edges = rm["states"], rm["edges"]
    n = len(edges)
    if n == 0:
        return "unsampled", [], [], []
    if not field.is_valid(start):
        return "invalid_start", [], [], []
    sc = [i for i in range(n)
          if distance(start, [float(v) for v in states[i]]) < radius
          and check_motion(field, bounds, fraction, start, [float(v) for v in states[i]])]
    gi = [i for i in range(n) if distance([float(v) for v in states[i]], goal_c) <= goal_r]
    if not sc or not gi:
        return "no_solution", sc, gi, []
    queue = deque(sc)                     # prm.rs:271
    parent = {}
    visited = [False] * n
    for i in sc:                          # prm.rs:275-279 (second enqueue)
        queue.append(i)
        parent[i] = None
        visited[i] = True
    reached = None
    while queue:
        cur = queue.popleft()
        if cur in gi:
            reached = cur
            break
        for nb in edges[cur]:
            if not visited[nb]:
                visited[nb] = True
                parent[nb] = cur
                queue.append(nb)
    if reached is None:
        return "no_solution", sc, gi, []
    chain = []
    cur = reached
    while parent[cur] is not None:        # prm.rs:199-203
        chain.append(cur)
        cur = parent[cur]
    chain.append(cur)
    chain.reverse()
    path = [list(start)] + [[float(v) for v in states[i]] for i in chain]
    return "solved", sc, gi, path


def csr_checksum(edges):
    h = 0xCBF29CE484222325
    for i, lst in enumerate(edges):
        h = ((h ^ (len(lst) & MASK)) * FNV_P) & MASK
        for v in lst:
            h = ((h ^ v) * FNV_P) & MASK
    return h


def states_checksum(states):
    h = 0xCBF29CE484222325
    for v in np.ascontiguousarray(states, dtype=np.float64).view(np.uint64).ravel():
        h = ((h ^ int(v)) * FNV_P) & MASK
    return h


def record(params, rm, queries, field):
    rec = dict(n=len(rm["edges"]), n_samples=rm["n_samples"],
               edge_entries=sum(len(e) for e in rm["edges"]),
               csr_checksum="%016x" % csr_checksum(rm["edges"]),
               states_checksum="%016x" % states_checksum(rm["states"]),
               states_head=[[hexf(v) for v in row] for row in rm["states"][:48]],
               edges_head=[list(e) for e in rm["edges"][:48]],
               queries=[])
    for start, goal_c, goal_r in queries:
        status, sc, gi, path = prm_solve(params["dim"], params["bounds"], params["radius"], params["fraction"], field,
                                         rm, start, goal_c, goal_r)
        rec["queries"].append(dict(start=start, goal_c=goal_c, goal_r=goal_r, status=status, start_connections=sc,
                                   goal_indices=gi, path=[[hexf(v) for v in row] for row in path]))
    return rec


def main():
    out = {}
    # ---- the reference's PRM test scene (oxmpl/tests/prm_rvss_tests.rs:111-160): [0,10]^2, wall at x=5,
    #      start (1,5), goal ball (9,5) r=0.5, PRM::new(5.0, 0.5)
    pw = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], radius=0.5, fraction=0.05, spheres=[],
              boxes=[([4.75, 2.0], [5.25, 8.0])], seed=3, stream=0, max_milestones=1500, max_samples=10 ** 9)
    fw = Field(2, [], pw["boxes"])
    rm = prm_construct(2, pw["bounds"], 0.5, 0.05, fw, 3, 0, 1500, 10 ** 9)
    out["wall"] = dict(params=pw, run=record(pw, rm, [([1.0, 5.0], [9.0, 5.0], 0.5),
                                                      ([9.0, 9.0], [1.0, 1.0], 0.4),
                                                      ([5.0, 5.0], [9.0, 5.0], 0.5),      # start inside the wall
                                                      ([1.0, 5.0], [20.0, 20.0], 0.5)],   # goal outside the bounds
                                             fw))
    # ---- R^3 with the 64-sphere field of BASELINE.json configs[1]
    start3, goal3 = [0.5, 0.5, 0.5], [9.5, 9.5, 9.5]
    spheres = sphere_field(0x5EED0001, 64, 3, 1.0, 9.0, 0.3, 0.8, [start3, goal3])
    p3 = dict(dim=3, bounds=[(0.0, 10.0)] * 3, radius=1.5, fraction=0.05, boxes=[], seed=42, stream=1,
              max_milestones=700, max_samples=10 ** 9,
              spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in spheres])
    f3 = Field(3, spheres)
    rm = prm_construct(3, p3["bounds"], 1.5, 0.05, f3, 42, 1, 700, 10 ** 9)
    out["r3"] = dict(params=p3, run=record(p3, rm, [(start3, goal3, 1.0), (goal3, start3, 1.0)], f3))
    # ---- R^6 (BASELINE.json configs[4] shape, small): 16 hyperspheres, connection radius 4
    s6, g6 = [3.0] * 6, [7.0] * 6
    sph6 = sphere_field(0x5EED0006, 16, 6, 1.0, 9.0, 3.0, 4.5, [s6, g6])
    p6 = dict(dim=6, bounds=[(0.0, 10.0)] * 6, radius=4.0, fraction=0.05, boxes=[], seed=7, stream=2,
              max_milestones=600, max_samples=10 ** 9,
              spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in sph6])
    f6 = Field(6, sph6)
    rm = prm_construct(6, p6["bounds"], 4.0, 0.05, f6, 7, 2, 600, 10 ** 9)
    out["r6"] = dict(params=p6, run=record(p6, rm, [(s6, g6, 3.0)], f6))
    # ---- sample cap: stops on max_samples before max_milestones
    pc = dict(pw)
    pc.update(max_milestones=10 ** 6, max_samples=300, seed=9, stream=4)
    rm = prm_construct(2, pc["bounds"], 0.5, 0.05, fw, 9, 4, 10 ** 6, 300)
    out["sample_cap"] = dict(params=pc, run=record(pc, rm, [([1.0, 5.0], [9.0, 5.0], 0.5)], fw))

    path = os.path.join(HERE, "prm_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)
    for k, v in out.items():
        r = v["run"]
        print(k, "n", r["n"], "samples", r["n_samples"], "edge entries", r["edge_entries"], "csr", r["csr_checksum"],
              [(q["status"], len(q["start_connections"]), len(q["goal_indices"]), len(q["path"])) for q in r["queries"]])


if __name__ == "__main__":
    main()
