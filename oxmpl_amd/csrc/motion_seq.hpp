// motion_seq.hpp -- validity and motion check evaluated by ONE thread (the RRT kernels stripe theirs over a workgroup or
// a wave): used where every lane owns a motion of its own -- PRM's edge kernel (prm_kernels.hip) and the edge kernel of the
// decoupled RRT* (rrt_star_wire.hip).  check_motion as in rrt.rs:90-116 / rrt_star.rs:73-104 / prm.rs:161-187 (the same
// loop in all three): num_steps = ceil(dist / res); <= 1: is_valid(to); else every interpolated state from + (to - from) s/n.
#pragma once
#include "rrt_device.hpp"

namespace oxhip {

template <int D>
__device__ __forceinline__ bool state_valid_seq(const DevParams& p, const double s[D]) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    for (uint32_t j = 0; j < nobs; ++j)  // j is wave-uniform: the obstacle table is read through the scalar cache
        if (obstacle_hit<D>(p, D, s, j)) return false;
    return true;
}

// prm.rs:161-187, one thread per motion.  Every state the check interpolates lies within dist/2 (+ rounding)
// of the segment midpoint, so a sphere with d2(centre, mid) > (r + h)^2 cannot be hit by any of them and
// is dropped before the step loop (h = dist/2 with a relative 1e-6 and an absolute 1e-9 * |coordinates|
// margin, orders of magnitude above the few-ulp rounding of interpolate; the filter never decides a
// motion invalid).  Typically 2-3 of 32 spheres survive, which is what makes this phase cheap.
template <int D>
__device__ __forceinline__ bool motion_valid_seq(const DevParams& p, const double from[D], const double to[D]) {
    if (p.n_spheres + p.n_boxes == 0) return true;
    const double dist = sqrt(dist2<D>(from, to, D));
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    if (nsteps <= 1) return state_valid_seq<D>(p, to);
    const double dn = (double)nsteps;
    double mid[D];
    lerp<D>(from, to, 0.5, mid, D);
    const double h = 0.5 * dist * (1.0 + 1e-6) + p.filt_abs;
    for (uint32_t w0 = 0; w0 < p.n_spheres; w0 += 64) {
        const uint32_t cnt = p.n_spheres - w0 < 64u ? p.n_spheres - w0 : 64u;
        uint64_t mask = 0;
        for (uint32_t jj = 0; jj < cnt; ++jj) {   // wave-uniform index: scalar loads
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) c[k] = p.sph_c[(size_t)k * p.n_spheres + w0 + jj];
            const double rr = p.sph_r[w0 + jj] + h;
            const double lim = rr * rr * (1.0 + 1e-9);
            if (!(dist2<D>(c, mid, D) > lim)) mask |= 1ull << jj;   // NaN / inf radii stay in
        }
        if (mask == 0) continue;
        for (uint32_t step = 1; step <= nsteps; ++step) {
            const double t = (double)step / dn;
            double s[D];
            lerp<D>(from, to, t, s, D);
            for (uint64_t m = mask; m != 0; m &= m - 1)
                if (obstacle_hit<D>(p, D, s, w0 + (uint32_t)(__ffsll((unsigned long long)m) - 1))) return false;
        }
    }
    if (p.n_boxes != 0) {
        for (uint32_t step = 1; step <= nsteps; ++step) {
            const double t = (double)step / dn;
            double s[D];
            lerp<D>(from, to, t, s, D);
            for (uint32_t b = 0; b < p.n_boxes; ++b)
                if (obstacle_hit<D>(p, D, s, p.n_spheres + b)) return false;
        }
    }
    return true;
}

}  // namespace oxhip
