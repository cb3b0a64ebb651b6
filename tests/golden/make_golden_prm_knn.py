#!/usr/bin/env python3
"""Generates tests/golden/prm_knn_golden.json: roadmaps of the k-NEAREST variant of PRM (BASELINE.json configs[4] words the
roadmap as "all-pairs k-NN"; the reference itself connects by radius, oxmpl/src/geometric/planners/prm.rs:131-138), by an
independent numpy restatement on top of make_golden_prm.py's pieces.

The variant, defined here and in oracle/prm_oracle.c alike: when milestone j is added (prm.rs:122-145) its candidates are the k
EARLIER milestones nearest to it, ordered by (distance, index) -- `distance` = the space's sqrt form, the lower index first among
equal distances; fewer than k earlier milestones: all of them --, visited in ascending index order; an edge is made, both ways
(prm.rs:143-145), iff check_motion(new -> old) holds.  Everything else (sampling, validity, the query with its radius rule for the
start connections, the BFS) is prm.rs unchanged.  PARITY UNPINNED: an extension has no upstream to pin against.

Run:  python tests/golden/make_golden_prm_knn.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ChaCha12Rng, Field, check_motion, hexf, random_range, sphere_field  # noqa: E402
import make_golden_prm as mp  # noqa: E402


def prm_construct_knn(dim, bounds, k, fraction, field, seed, stream, max_milestones, max_samples):
    rng = ChaCha12Rng(seed, stream)
    states = np.zeros((0, dim))
    edges = []
    n_samples = 0
    while len(edges) < max_milestones and n_samples < max_samples:
        q = [random_range(rng, lo, hi) for lo, hi in bounds]
        n_samples += 1
        if not field.is_valid(q):
            continue
        mine = []
        n = len(edges)
        if n:
            acc = np.zeros(n)
            for kk in range(dim):
                d = q[kk] - states[:, kk]
                acc = acc + d * d
            dist = np.sqrt(acc)
            order = np.lexsort((np.arange(n), dist))[:min(k, n)]     # by distance, then index
            for i in sorted(int(v) for v in order):                  # visited in ascending index order
                if check_motion(field, bounds, fraction, q, [float(v) for v in states[i]]):
                    mine.append(i)
        new_idx = n
        edges.append(mine)
        for i in mine:
            edges[i].append(new_idx)
        states = np.vstack([states, np.array(q)[None, :]])
    return dict(states=states, edges=edges, n_samples=n_samples)


def main():
    out = {}
    pw = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], radius=0.5, knn_k=6, fraction=0.05, spheres=[],
              boxes=[([4.75, 2.0], [5.25, 8.0])], seed=3, stream=0, max_milestones=1200, max_samples=10 ** 9)
    fw = Field(2, [], pw["boxes"])
    rm = prm_construct_knn(2, pw["bounds"], 6, 0.05, fw, 3, 0, 1200, 10 ** 9)
    out["wall_k6"] = dict(params=pw, run=mp.record(pw, rm, [([1.0, 5.0], [9.0, 5.0], 0.5), ([9.0, 9.0], [1.0, 1.0], 0.4)], fw))
    start3, goal3 = [0.5, 0.5, 0.5], [9.5, 9.5, 9.5]
    spheres = sphere_field(0x5EED0001, 64, 3, 1.0, 9.0, 0.3, 0.8, [start3, goal3])
    p3 = dict(dim=3, bounds=[(0.0, 10.0)] * 3, radius=1.5, knn_k=10, fraction=0.05, boxes=[], seed=42, stream=1, max_milestones=700,
              max_samples=10 ** 9, spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in spheres])
    f3 = Field(3, spheres)
    rm = prm_construct_knn(3, p3["bounds"], 10, 0.05, f3, 42, 1, 700, 10 ** 9)
    out["r3_k10"] = dict(params=p3, run=mp.record(p3, rm, [(start3, goal3, 1.0)], f3))
    s6, g6 = [3.0] * 6, [7.0] * 6
    sph6 = sphere_field(0x5EED0006, 16, 6, 1.0, 9.0, 3.0, 4.5, [s6, g6])
    p6 = dict(dim=6, bounds=[(0.0, 10.0)] * 6, radius=4.0, knn_k=8, fraction=0.05, boxes=[], seed=7, stream=2, max_milestones=600,
              max_samples=10 ** 9, spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in sph6])
    f6 = Field(6, sph6)
    rm = prm_construct_knn(6, p6["bounds"], 8, 0.05, f6, 7, 2, 600, 10 ** 9)
    out["r6_k8"] = dict(params=p6, run=mp.record(p6, rm, [(s6, g6, 3.0)], f6))
    pk1 = dict(pw)
    pk1.update(knn_k=1, max_milestones=300, seed=11, stream=6)       # k = 1: a forest of nearest-earlier links
    rm = prm_construct_knn(2, pk1["bounds"], 1, 0.05, fw, 11, 6, 300, 10 ** 9)
    out["wall_k1"] = dict(params=pk1, run=mp.record(pk1, rm, [([1.0, 5.0], [9.0, 5.0], 0.5)], fw))
    path = os.path.join(HERE, "prm_knn_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)
    for kname, v in out.items():
        r = v["run"]
        print(kname, "n", r["n"], "samples", r["n_samples"], "edge entries", r["edge_entries"], [(q["status"], len(q["path"])) for q in r["queries"]])


if __name__ == "__main__":
    main()
