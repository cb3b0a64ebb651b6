// ox_sincos.hpp -- sin / cos of the disc goal sampler (OXHIP_GOAL_SAMPLE_UNIFORM_DISC: oxmpl/tests/rrt_rvss_tests.rs:55-66 calls
// angle.cos() / angle.sin(), i.e. whatever libm the host has).  One portable routine instead: argument reduction by pi/2 in
// three Cody-Waite pieces and the sin / cos kernels of FreeBSD msun (e_rem_pio2.c medium path, k_sin.c, k_cos.c), every
// operation a single unfused binary64 operation (the translation unit is built with -ffp-contract=off), so the device and the
// CPU checker of the test suite, which restates it operation for operation, agree bit for bit.  Valid for
// 0 <= x < 2^19 pi/2; the sampler passes [0, 2 pi).  Below one ulp; against a given libm the last bit differs now and then.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oxhip {

__device__ __forceinline__ uint32_t ox_hi_word(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ double ox_from_hi(uint32_t hi) { return __hiloint2double((int)hi, 0); }

__device__ __forceinline__ double ox_k_sin(double x, double y, int iy) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

__device__ __forceinline__ double ox_k_cos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const uint32_t ix = ox_hi_word(x) & 0x7fffffffu;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    const double qx = ix > 0x3fe90000u ? 0.28125 : ox_from_hi(ix - 0x00200000u);
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

// x in [0, 2^19 pi/2): s = sin x, c = cos x
__device__ __forceinline__ void ox_sincos(double x, double& s, double& c) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    const uint32_t ix = ox_hi_word(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) {   // |x| <= pi/4
        if (ix < 0x3e400000u) { s = x; c = 1.0; return; }   // |x| < 2^-27
        s = ox_k_sin(x, 0.0, 0);
        c = ox_k_cos(x, 0.0);
        return;
    }
    const int n = (int)(x * invpio2 + 0.5);
    const double fn = (double)n;
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    const int j = (int)(ix >> 20);
    double y0 = r - w;
    int i = j - (int)((ox_hi_word(y0) >> 20) & 0x7ffu);
    if (i > 16) {   // second piece
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y0 = r - w;
        i = j - (int)((ox_hi_word(y0) >> 20) & 0x7ffu);
        if (i > 49) {   // third piece
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            y0 = r - w;
        }
    }
    const double y1 = (r - y0) - w;
    const double ks = ox_k_sin(y0, y1, 1), kc = ox_k_cos(y0, y1);
    switch (n & 3) {
        case 0: s = ks; c = kc; break;
        case 1: s = kc; c = -ks; break;
        case 2: s = -ks; c = -kc; break;
        default: s = -kc; c = ks; break;
    }
}

}  // namespace oxhip
