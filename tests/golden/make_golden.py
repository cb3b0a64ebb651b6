#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

This is an INDEPENDENT restatement (pure Python ints for the RNG, numpy f64 for the
geometry -- IEEE binary64, correctly rounded sqrt/div, no FMA) of the same reference
lines the C oracle follows:

  oxmpl/src/geometric/planners/rrt.rs:90-128,140-227
  oxmpl/src/base/spaces/real_vector_state_space.rs:103-129,137-186,233-253
  rand 0.9.1 random_bool / random_range(f64); rand_chacha 0.9.0 ChaCha12Rng

The reference itself cannot be built or imported here (no Rust toolchain, oxmpl_py not
installed), and holds no golden vectors for this path, so these
fixtures pin the C oracle against a second implementation, not against oxmpl:
PARITY UNPINNED.  The only externally published vectors are the ChaCha block outputs
(RFC 7539 2.3.2; the all-zero-key ChaCha20/12/8 keystreams).

Usage: python tests/golden/make_golden.py   (writes *.json next to this file)
"""
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------- RNG
def _rotl(v, c):
    return ((v << c) & M32) | (v >> (32 - c))


def chacha_block(key, counter, stream, rounds):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key)
    s += [counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + s[i]) & M32 for i in range(16)]


class ChaCha12Rng:
    """ChaCha12Rng::from_seed(seed LE || 24 zero bytes) + set_stream(problem_id)."""

    def __init__(self, seed, stream):
        self.key = [seed & M32, (seed >> 32) & M32, 0, 0, 0, 0, 0, 0]
        self.stream = stream
        self.counter = 0
        self.buf = []
        self.index = 64
        self.draws = 0

    def next_u64(self):
        if self.index >= 64:
            self.buf = []
            for b in range(4):
                self.buf += chacha_block(self.key, self.counter + b, self.stream, 12)
            self.counter += 4
            self.index = 0
        lo, hi = self.buf[self.index], self.buf[self.index + 1]
        self.index += 2
        self.draws += 1
        return (hi << 32) | lo


def bernoulli_p_int(p):
    if p == 1.0:
        return M64
    v = p * 18446744073709551616.0
    if not (v > 0.0):
        return 0
    if v >= 18446744073709551616.0:
        return M64
    return int(v)


def random_bool(rng, p):
    p_int = bernoulli_p_int(p)
    if p_int == M64:
        return True
    return rng.next_u64() < p_int


def random_range(rng, lo, hi):
    scale = hi - lo
    while True:
        bits = (rng.next_u64() >> 12) | 0x3FF0000000000000
        v12 = struct.unpack("<d", struct.pack("<Q", bits))[0]
        v01 = v12 - 1.0
        res = v01 * scale
        res = res + lo
        if res < hi:
            return res


# ---------------------------------------------------------------------- geometry
def distance(a, b):
    acc = 0.0
    for x, y in zip(a, b):
        d = x - y
        acc = acc + d * d
    return math.sqrt(acc)


def interpolate(frm, to, t):
    return [f + (g - f) * t for f, g in zip(frm, to)]


def maximum_extent(bounds):
    if any((not math.isfinite(lo)) or (not math.isfinite(hi)) for lo, hi in bounds):
        return 1.0
    acc = 0.0
    for lo, hi in bounds:
        d = hi - lo
        acc = acc + d * d
    return math.sqrt(acc)


def num_steps(dist, lvsl):
    q = dist / (lvsl * 0.1)
    if math.isnan(q):
        return 0
    c = math.ceil(q) if math.isfinite(q) else q
    if not (c > 0.0):
        return 0
    if c >= 18446744073709551616.0:
        return M64
    return int(c)


def f64_bits(v):
    return struct.unpack("<Q", struct.pack("<d", float(v)))[0]


class Field:
    def __init__(self, dim, spheres=(), boxes=()):
        self.dim = dim
        self.sc = np.array([s[0] for s in spheres], dtype=np.float64).reshape(-1, dim)
        self.sr = np.array([s[1] for s in spheres], dtype=np.float64)
        self.boxes = [(list(lo), list(hi)) for lo, hi in boxes]

    def is_valid(self, p):
        if len(self.sr):
            acc = np.zeros(len(self.sr))
            for k in range(self.dim):
                d = self.sc[:, k] - p[k]
                acc = acc + d * d
            if not bool(np.all(np.sqrt(acc) > self.sr)):
                return False
        for lo, hi in self.boxes:
            if all(lo[k] <= p[k] <= hi[k] for k in range(self.dim)):
                return False
        return True


def check_motion(field, bounds, fraction, frm, to):
    dist = distance(frm, to)
    lvsl = maximum_extent(bounds) * fraction
    n = num_steps(dist, lvsl)
    if n <= 1:
        return field.is_valid(to)
    for i in range(1, n + 1):
        t = float(i) / float(n)
        if not field.is_valid(interpolate(frm, to, t)):
            return False
    return True


FNV_P = 0x100000001B3
FNV_BASIS = 0xCBF29CE484222325


def rrt_solve(dim, bounds, max_distance, goal_bias, fraction, field, start, goal_c, goal_r,
              seed, pid, max_iterations, max_nodes, stop_at_goal=True, freeze=False, goal_sampler=None):
    rng = ChaCha12Rng(seed, pid)
    cap = max_nodes + 1
    tree = np.zeros((cap, dim), dtype=np.float64)
    parents = [-1]
    tree[0] = start
    n = 1
    chk = 0xCBF29CE484222325
    iterations = accepted = 0
    goal_node = -1
    for _ in range(max_iterations):
        if (not freeze) and n >= max_nodes:
            break
        if random_bool(rng, goal_bias):
            q_rand = goal_sampler(rng) if goal_sampler is not None else list(goal_c)   # (make_golden_disc.py passes the disc sampler)
        else:
            q_rand = [random_range(rng, lo, hi) for lo, hi in bounds]
        acc = np.zeros(n)
        for k in range(dim):
            d = tree[:n, k] - q_rand[k]
            acc = acc + d * d
        dists = np.sqrt(acc)
        nearest = int(np.argmin(dists))  # first occurrence of the minimum == strict '<' scan
        min_dist = float(dists[nearest])
        q_near = [float(v) for v in tree[nearest]]
        if min_dist > max_distance:
            t = max_distance / min_dist
            q_new = interpolate(q_near, q_rand, t)
        else:
            q_new = list(q_rand)
        ok = check_motion(field, bounds, fraction, q_near, q_new)
        # checksum (build-defined): digest of the iteration folded from the FNV basis, then H <- H * P + g (mod 2^64)
        g = ((FNV_BASIS ^ nearest) * FNV_P) & M64
        for v in q_new:
            g = ((g ^ f64_bits(v)) * FNV_P) & M64
        g = ((g ^ int(ok)) * FNV_P) & M64
        chk = (chk * FNV_P + g) & M64
        iterations += 1
        hit = False
        if ok:
            accepted += 1
            if not freeze:
                tree[n] = q_new
                parents.append(nearest)
                n += 1
                if distance(q_new, goal_c) <= goal_r:
                    if goal_node < 0:
                        goal_node = n - 1
                    hit = True
        if hit and stop_at_goal:
            break
    path = []
    if goal_node >= 0:
        i = goal_node
        while i >= 0:
            path.append([float(v) for v in tree[i]])
            i = parents[i]
        path.reverse()
    return dict(n=n, iterations=iterations, accepted=accepted, checksum=chk, goal_node=goal_node,
                states=tree[:n].copy(), parents=parents, path=path, rng_draws=rng.draws)


def _nearest(tree, n, q, dim):
    acc = np.zeros(n)
    for k in range(dim):
        d = tree[:n, k] - q[k]
        acc = acc + d * d
    dists = np.sqrt(acc)
    i = int(np.argmin(dists))
    return i, float(dists[i])


def rrt_connect_solve(dim, bounds, max_distance, goal_bias, fraction, field, start, goal_c, goal_r,
                      seed, pid, max_iterations, max_nodes):
    """oxmpl/src/geometric/planners/rrt_connect.rs:121-159,199-309 (goal tree root = goal centre)"""
    rng = ChaCha12Rng(seed, pid)
    trees = [np.zeros((max_nodes + 1, dim)), np.zeros((max_nodes + 1, dim))]
    parents = [[-1], [-1]]
    trees[0][0] = start
    trees[1][0] = goal_c
    n = [1, 1]
    chk = 0xCBF29CE484222325
    iterations = 0
    end = [-1, -1]

    def extend(w, target):
        i, md = _nearest(trees[w], n[w], target, dim)
        q_near = [float(v) for v in trees[w][i]]
        if md > max_distance:
            q_new, res = interpolate(q_near, target, max_distance / md), 1
        else:
            q_new, res = list(target), 2
        if not check_motion(field, bounds, fraction, q_near, q_new):
            return 0, i, q_new
        trees[w][n[w]] = q_new
        parents[w].append(i)
        n[w] += 1
        return res, i, q_new

    for _ in range(max_iterations):
        if n[0] >= max_nodes or n[1] >= max_nodes:
            break
        grow_start = n[0] <= n[1]
        if random_bool(rng, goal_bias):
            q_rand = list(goal_c)
        else:
            q_rand = [random_range(rng, lo, hi) for lo, hi in bounds]
        wa, wb = (0, 1) if grow_start else (1, 0)
        ra, near_a, q_new_a = extend(wa, q_rand)
        chk = ((chk ^ int(grow_start)) * FNV_P) & M64
        chk = ((chk ^ near_a) * FNV_P) & M64
        for v in q_new_a:
            chk = ((chk ^ f64_bits(v)) * FNV_P) & M64
        chk = ((chk ^ ra) * FNV_P) & M64
        iterations += 1
        done = False
        if ra:
            idx_a = n[wa] - 1
            if grow_start and distance(q_new_a, goal_c) <= goal_r:
                end = [idx_a, -1]
                done = True
            else:
                rb, near_b, q_new_b = extend(wb, q_new_a)
                chk = ((chk ^ near_b) * FNV_P) & M64
                for v in q_new_b:
                    chk = ((chk ^ f64_bits(v)) * FNV_P) & M64
                chk = ((chk ^ rb) * FNV_P) & M64
                if rb == 2:
                    idx_b = n[wb] - 1
                    end = [idx_a, idx_b] if grow_start else [idx_b, idx_a]
                    done = True
        if done:
            break
    path = []
    if end[0] >= 0:
        i = end[0]
        while i >= 0:
            path.append([float(v) for v in trees[0][i]])
            i = parents[0][i]
        path.reverse()
        if end[1] >= 0:
            i = parents[1][end[1]]
            while i >= 0:
                path.append([float(v) for v in trees[1][i]])
                i = parents[1][i]
    return dict(n=n, iterations=iterations, checksum=chk, end=end, path=path, rng_draws=rng.draws,
                states=[trees[0][:n[0]].copy(), trees[1][:n[1]].copy()], parents=parents)


# --------------------------------------------------------------------- scenarios
def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return state, z ^ (z >> 31)


def sphere_field(seed, n, dim, lo, hi, rmin, rmax, keep_clear):
    """n spheres, centres U[lo,hi)^dim, radii U[rmin,rmax) via SplitMix64 + the [1,2)-1 transform;
    a sphere is redrawn when it comes within 0.5 of any keep_clear point."""
    st = seed

    def u(a, b):
        nonlocal st
        st, z = splitmix64(st)
        bits = (z >> 12) | 0x3FF0000000000000
        v = struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0
        return v * (b - a) + a

    out = []
    while len(out) < n:
        c = [u(lo, hi) for _ in range(dim)]
        r = u(rmin, rmax)
        if all(distance(c, p) > r + 0.5 for p in keep_clear):
            out.append((c, r))
    return out


def hexf(v):
    return "%016x" % f64_bits(v)


def tree_record(res, first):
    m = min(first, res["n"])
    return dict(
        n=res["n"], iterations=res["iterations"], accepted=res["accepted"],
        checksum="%016x" % res["checksum"], goal_node=res["goal_node"], rng_draws=res["rng_draws"],
        first_states=[[hexf(v) for v in row] for row in res["states"][:m]],
        first_parents=[int(p) for p in res["parents"][:m]],
        path=[[hexf(v) for v in row] for row in res["path"]],
    )


def main():
    out = {}

    # ---- RNG KATs
    rng = ChaCha12Rng(42, 7)
    out["rng"] = dict(
        seed=42, stream=7,
        u64=["%016x" % rng.next_u64() for _ in range(40)],
        chacha12_zero_block=[("%08x" % w) for w in chacha_block([0] * 8, 0, 0, 12)],
        p_int={repr(p): "%016x" % bernoulli_p_int(p) for p in [0.0, 0.05, 0.5, 0.999, 1.0, 1e-300]},
    )
    rng = ChaCha12Rng(1234, 0)
    out["rng"]["range"] = [
        dict(lo=lo, hi=hi, v=[hexf(random_range(rng, lo, hi)) for _ in range(6)])
        for lo, hi in [(0.0, 10.0), (-10.0, 10.0), (1.0, 1.0000000000000004), (-1e300, 1e300), (3.0, 3.5)]
    ]
    rng = ChaCha12Rng(99, 3)
    out["rng"]["bools"] = dict(seed=99, stream=3, p=0.05, v=[int(random_bool(rng, 0.05)) for _ in range(200)])

    # ---- arithmetic KATs (values chosen to exercise rounding: non-representable decimals)
    r2 = ChaCha12Rng(7, 7)
    kat = []
    for dim in (1, 2, 3, 6):
        for _ in range(8):
            a = [random_range(r2, -10.0, 10.0) for _ in range(dim)]
            b = [random_range(r2, -10.0, 10.0) for _ in range(dim)]
            t = random_range(r2, 0.0, 1.0)
            kat.append(dict(a=[hexf(v) for v in a], b=[hexf(v) for v in b], t=hexf(t),
                            distance=hexf(distance(a, b)),
                            interpolate=[hexf(v) for v in interpolate(a, b, t)]))
    out["space"] = dict(
        kat=kat,
        extent=[dict(bounds=bd, extent=hexf(maximum_extent(bd))) for bd in
                [[(-10.0, 10.0)] * 2, [(0.0, 10.0)] * 3, [(0.0, 10.0)] * 2, [(-1.0, 1.0), (-2.0, 2.0)],
                 [(0.0, 1.0)] * 6]],
        num_steps=[dict(dist=hexf(d), lvsl=hexf(l), n=num_steps(d, l)) for d, l in
                   [(0.5, math.sqrt(300.0) * 0.05), (0.5, math.sqrt(800.0) * 0.05),
                    (0.5, math.sqrt(200.0) * 0.05), (0.0, 1.0), (1e-9, 1.0), (0.1, 1.0), (0.1000001, 1.0),
                    (3.0, 0.25), (0.49999999999999994, math.sqrt(300.0) * 0.05)]],
    )

    # ---- config 1: README quick-start (README.md:147-171): 2-D, disc obstacle r=2 at origin
    c1 = dict(dim=2, bounds=[(-10.0, 10.0), (-10.0, 10.0)], max_distance=0.5, goal_bias=0.05,
              fraction=0.05, start=[-5.0, -5.0], goal_c=[5.0, 5.0], goal_r=0.5,
              spheres=[([0.0, 0.0], 2.0)], boxes=[], max_nodes=20000, max_iterations=200000)
    f1 = Field(2, c1["spheres"])
    runs = []
    for seed in range(6):
        res = rrt_solve(2, c1["bounds"], 0.5, 0.05, 0.05, f1, c1["start"], c1["goal_c"], 0.5,
                        seed, 0, c1["max_iterations"], c1["max_nodes"])
        rec = tree_record(res, 64)
        rec["seed"] = seed
        rec["pid"] = 0
        runs.append(rec)
    out["config1"] = dict(params=c1, runs=runs)

    # ---- reference test scenario (oxmpl/tests/rrt_rvss_tests.rs:109-165): wall, RRT::new(0.5, 0.0)
    cw = dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], max_distance=0.5, goal_bias=0.0,
              fraction=0.05, start=[1.0, 5.0], goal_c=[9.0, 5.0], goal_r=0.5, spheres=[],
              boxes=[([4.75, 2.0], [5.25, 8.0])], max_nodes=20000, max_iterations=200000)
    fw = Field(2, [], cw["boxes"])
    runs = []
    for seed in range(3):
        res = rrt_solve(2, cw["bounds"], 0.5, 0.0, 0.05, fw, cw["start"], cw["goal_c"], 0.5,
                        seed, 5, cw["max_iterations"], cw["max_nodes"])
        rec = tree_record(res, 64)
        rec["seed"] = seed
        rec["pid"] = 5
        runs.append(rec)
    out["wall"] = dict(params=cw, runs=runs)

    # ---- config 2: R^3, 64 spheres (BASELINE.json configs[1]); truncated growth
    start3, goal3 = [0.5, 0.5, 0.5], [9.5, 9.5, 9.5]
    spheres = sphere_field(0x5EED0001, 64, 3, 1.0, 9.0, 0.3, 0.8, [start3, goal3])
    c2 = dict(dim=3, bounds=[(0.0, 10.0)] * 3, max_distance=0.5, goal_bias=0.05, fraction=0.05,
              start=start3, goal_c=goal3, goal_r=0.5, boxes=[],
              spheres=[[[hexf(v) for v in c], hexf(r)] for c, r in spheres])
    f2 = Field(3, spheres)
    runs = []
    for pid in (0, 1, 1023):
        res = rrt_solve(3, c2["bounds"], 0.5, 0.05, 0.05, f2, start3, goal3, 0.5, 42, pid,
                        max_iterations=1500, max_nodes=10000, stop_at_goal=False)
        rec = tree_record(res, 256)
        rec["seed"] = 42
        rec["pid"] = pid
        rec["max_iterations"] = 1500
        runs.append(rec)
    # a frozen ("steady") leg on the tree grown above: 200 iterations, inserts suppressed
    out["config2"] = dict(params=c2, runs=runs)

    # ---- RRTConnect (rrt_connect.rs) on the README scene and on the reference's wall test scene
    #      (oxmpl/tests/rrt_connect_rvss_tests.rs: RRTConnect::new(0.5, 0.0))
    for key, prm, fld in (("connect_config1", c1, f1), ("connect_wall", cw, fw)):
        runs = []
        for seed in range(4):
            res = rrt_connect_solve(2, prm["bounds"], 0.5, prm["goal_bias"], 0.05, fld, prm["start"], prm["goal_c"], 0.5,
                                    seed, 11, 200000, 20000)
            runs.append(dict(seed=seed, pid=11, n=res["n"], iterations=res["iterations"], checksum="%016x" % res["checksum"],
                             end=res["end"], rng_draws=res["rng_draws"],
                             path=[[hexf(v) for v in row] for row in res["path"]],
                             states=[[[hexf(v) for v in row] for row in t[:48]] for t in res["states"]],
                             parents=[[int(x) for x in pp[:48]] for pp in res["parents"]]))
        out[key] = dict(params=prm, runs=runs)

    with open(os.path.join(HERE, "rrt_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "rrt_golden.json"))
    for k in ("connect_config1", "connect_wall"):
        for r in out[k]["runs"]:
            print(k, "seed", r["seed"], "n", r["n"], "iters", r["iterations"], "end", r["end"], "path", len(r["path"]), "chk", r["checksum"])
    for k in ("config1", "wall", "config2"):
        for r in out[k]["runs"]:
            print(k, "seed", r["seed"], "pid", r["pid"], "n", r["n"], "iters", r["iterations"],
                  "goal", r["goal_node"], "path", len(r["path"]), "chk", r["checksum"])


if __name__ == "__main__":
    main()
