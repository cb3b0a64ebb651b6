#!/usr/bin/env python3
"""Golden vectors for the disc goal sampler (OXHIP_GOAL_SAMPLE_UNIFORM_DISC): an independent pure-Python restatement, run only
to (re)generate tests/golden/rrt_disc_golden.json.

What it restates: GoalSampleableRegion::sample_goal of the reference's test fixture, /root/reference/oxmpl/tests/rrt_rvss_tests.rs:55-66
(angle = rng.random_range(0.0..2.0 * PI); radius = self.radius * rng.random::<f64>().sqrt(); centre + radius (cos, sin)), with rand
0.9's transforms (random_range: the 52-bit [1, 2) - 1 form; random::<f64>(): (u64 >> 11) * 2^-53 -- restated from memory, the crate
is not in the image: PARITY UNPINNED as everywhere) and the build's portable sin / cos (ox_sincos: Cody-Waite reduction by pi/2 in
three pieces + the msun / fdlibm kernels, one unfused binary64 operation per step -- Python floats are exactly that).  Scenes: the
reference's wall scene and the README scene, goal_bias 0.05 / 0.5 / 1.0.

    python tests/golden/make_golden_disc.py      (writes tests/golden/rrt_disc_golden.json)
"""
import json
import math
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (ChaCha12Rng, random_range, rrt_solve, Field, tree_record)


def hi_word(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0] >> 32


def from_hi(hi):
    return struct.unpack("<d", struct.pack("<Q", hi << 32))[0]


S1, S2, S3 = -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04
S4, S5, S6 = 2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10
C1, C2, C3 = 4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05
C4, C5, C6 = -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11


def k_sin(x, y, iy):
    z = x * x
    v = z * x
    r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))
    if iy == 0:
        return x + v * (S1 + z * r)
    return x - ((z * (0.5 * y - v * r) - y) - v * S1)


def k_cos(x, y):
    ix = hi_word(x) & 0x7fffffff
    z = x * x
    r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))
    if ix < 0x3FD33333:
        return 1.0 - (0.5 * z - (z * r - x * y))
    qx = 0.28125 if ix > 0x3fe90000 else from_hi(ix - 0x00200000)
    hz = 0.5 * z - qx
    a = 1.0 - qx
    return a - (hz - (z * r - x * y))


def ox_sincos(x):
    invpio2, pio2_1, pio2_1t = 6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11
    pio2_2, pio2_2t = 6.07710050630396597660e-11, 2.02226624879595063154e-21
    pio2_3, pio2_3t = 2.02226624871116645580e-21, 8.47842766036889956997e-32
    ix = hi_word(x) & 0x7fffffff
    if ix <= 0x3fe921fb:
        if ix < 0x3e400000:
            return x, 1.0
        return k_sin(x, 0.0, 0), k_cos(x, 0.0)
    n = int(x * invpio2 + 0.5)
    fn = float(n)
    r = x - fn * pio2_1
    w = fn * pio2_1t
    j = ix >> 20
    y0 = r - w
    i = j - ((hi_word(y0) >> 20) & 0x7ff)
    if i > 16:
        t = r
        w = fn * pio2_2
        r = t - w
        w = fn * pio2_2t - ((t - r) - w)
        y0 = r - w
        i = j - ((hi_word(y0) >> 20) & 0x7ff)
        if i > 49:
            t = r
            w = fn * pio2_3
            r = t - w
            w = fn * pio2_3t - ((t - r) - w)
            y0 = r - w
    y1 = (r - y0) - w
    ks, kc = k_sin(y0, y1, 1), k_cos(y0, y1)
    return [(ks, kc), (kc, -ks), (-ks, -kc), (-kc, ks)][n & 3]


def disc_sampler(goal_c, goal_r):
    two_pi = 2.0 * 3.14159265358979323846

    def sample(rng):
        angle = mg.random_range(rng, 0.0, two_pi)
        u01 = float(rng.next_u64() >> 11) * 2.0 ** -53
        radius = goal_r * math.sqrt(u01)
        sn, cs = ox_sincos(angle)
        rx, ry = radius * cs, radius * sn
        return [goal_c[0] + rx, goal_c[1] + ry]
    return sample


def main():
    out = {"_generator": "tests/golden/make_golden_disc.py",
           "_parity": "UNPINNED: the reference cannot be built or imported here and holds no vectors for this path; rand's transforms are "
                      "restated from memory; sin / cos are the build's ox_sincos, within one ulp of any libm"}
    xs = [0.0, 1e-9, 0.5, 0.7853981633974483, 0.7853981633974484, 1.0, 1.5707963267948966, 2.0, 3.0, 3.141592653589793, 4.0, 4.71238898038469,
          5.5, 6.0, 6.283185307179585]
    out["sincos"] = [dict(x=mg.hexf(x), sin=mg.hexf(ox_sincos(x)[0]), cos=mg.hexf(ox_sincos(x)[1])) for x in xs]
    scenes = {
        "wall": dict(dim=2, bounds=[(0.0, 10.0), (0.0, 10.0)], max_distance=0.5, fraction=0.05, start=[1.0, 5.0], goal_c=[9.0, 5.0], goal_r=0.5,
                     spheres=[], boxes=[([4.75, 2.0], [5.25, 8.0])], max_nodes=20000, max_iterations=200000),
        "config1": dict(dim=2, bounds=[(-10.0, 10.0), (-10.0, 10.0)], max_distance=0.5, fraction=0.05, start=[-5.0, -5.0], goal_c=[5.0, 5.0],
                        goal_r=0.5, spheres=[([0.0, 0.0], 2.0)], boxes=[], max_nodes=20000, max_iterations=200000),
    }
    for name, sc in scenes.items():
        fld = mg.Field(2, sc["spheres"], sc["boxes"])
        runs = []
        for goal_bias in (0.05, 0.5, 1.0):
            for seed in range(2):
                res = mg.rrt_solve(2, sc["bounds"], sc["max_distance"], goal_bias, sc["fraction"], fld, sc["start"], sc["goal_c"], sc["goal_r"],
                                   seed, 9, sc["max_iterations"] if goal_bias < 1.0 else 300, sc["max_nodes"],
                                   goal_sampler=disc_sampler(sc["goal_c"], sc["goal_r"]))
                rec = mg.tree_record(res, 48)
                rec.update(seed=seed, pid=9, goal_bias=goal_bias, max_iterations=sc["max_iterations"] if goal_bias < 1.0 else 300)
                runs.append(rec)
        out[name] = dict(params=sc, runs=runs)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rrt_disc_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, [(k, len(v["runs"])) for k, v in out.items() if isinstance(v, dict)])


if __name__ == "__main__":
    main()
