// oxhip_host.hpp -- host-side helpers shared by the C-ABI translation units (oxhip_api.hip: RRT /
// RRTConnect batches and primitives; oxhip_prm_api.hip: PRM).  Not installed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/oxmpl_hip.h"

namespace oxhip {

inline thread_local std::string g_last_error;

inline int32_t fail(int32_t code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::oxhip::fail(OXHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define OX_TRY(expr) do { int32_t s_ = (expr); if (s_ != OXHIP_OK) return s_; } while (0)

inline int32_t select_device(int32_t device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(OXHIP_ERR_NO_DEVICE, "no HIP device visible (liboxmpl_hip has no CPU fallback)");
    if (device < 0 || device >= count) return fail(OXHIP_ERR_BAD_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return OXHIP_OK;
}

// largest x with sqrt(x) <= r under correctly rounded binary64 sqrt, so that
//   sqrt(d2) >  r  <=>  d2 >  T      (sphere validity, strict)
//   sqrt(d2) <= r  <=>  d2 <= T      (ball goal)
// hold exactly and the kernels need no sqrt per obstacle.  r < 0 -> -1, NaN -> NaN.
inline double sqrt_le_threshold(double r) {
    if (std::isnan(r)) return r;
    if (r < 0.0) return -1.0;
    if (std::isinf(r)) return r;
    double x = r * r;
    if (std::isinf(x)) x = std::numeric_limits<double>::max();
    while (std::sqrt(x) > r) x = std::nextafter(x, -1.0);
    for (;;) {
        double y = std::nextafter(x, std::numeric_limits<double>::infinity());
        if (std::isinf(y) || std::sqrt(y) > r) break;
        x = y;
    }
    return x;
}

// largest x with sqrt(x) < r:  sqrt(d2) < r  <=>  d2 <= T  (PRM's strict connection radius, prm.rs:134).
// r <= 0 or NaN -> -1 (no d2 >= 0 qualifies); +inf -> DBL_MAX (every finite d2 qualifies).
inline double sqrt_lt_threshold(double r) {
    if (std::isnan(r) || r <= 0.0) return -1.0;
    if (std::isinf(r)) return std::numeric_limits<double>::max();
    return sqrt_le_threshold(std::nextafter(r, 0.0));
}

// rand 0.9 Bernoulli::new
inline uint64_t bernoulli_p_int(double p) {
    if (p == 1.0) return ~0ull;
    double v = p * 18446744073709551616.0;
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)v;
}

constexpr double kMaxMagnitude = 1e150;  // keeps every squared difference finite

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
};

// Streams are recycled.  On this stack hipStreamCreate takes 1.5 ms and hipStreamDestroy 1.7 ms (rocprofv3 --hip-trace of
// tools/bench_single.py, profiles/r3_single/hip_api_stats.csv) -- more than every other call of a one-problem Planner::solve together,
// kernel included -- so a destroyed planner's stream (drained) goes to a small per-device pool and the next create takes it from
// there.  Pooled streams live until the process ends.
hipError_t oxhip_stream_acquire(int device, hipStream_t* out);   // non-blocking streams; `device` is the current device
void oxhip_stream_release(int device, hipStream_t s);            // synchronises s first

struct TmpStream {
    hipStream_t s = nullptr;
    int device = 0;
    ~TmpStream() { if (s) oxhip_stream_release(device, s); }
};
template <typename T>
int32_t to_device(DevBuf<T>& buf, const T* host, size_t n, hipStream_t s) {
    HIP_TRY(buf.alloc(n));
    if (n) HIP_TRY(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    return OXHIP_OK;
}
template <typename T>
int32_t to_host(T* host, const DevBuf<T>& buf, size_t n, hipStream_t s) {
    if (n) HIP_TRY(hipMemcpyAsync(host, buf.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return OXHIP_OK;
}
inline int32_t upload(DevBuf<double>& buf, const std::vector<double>& host, hipStream_t s) {
    HIP_TRY(buf.alloc(host.size()));
    if (!host.empty()) {
        HIP_TRY(hipMemcpyAsync(buf.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return OXHIP_OK;
}

// RealVectorStateSpace::new + get_maximum_extent + get_longest_valid_segment_length for a config's bounds:
// validates what the reference rejects (real_vector_state_space.rs:69-93,239-244), clamps the fraction
// (rvss.rs:121-129) and returns res = lvsl * 0.1 (rrt.rs:97, prm.rs:168).
inline int32_t space_resolution(uint32_t dim, const double* bounds, double& fraction, double& res) {
    if (dim == 0 || dim > OXHIP_MAX_DIM) return fail(OXHIP_ERR_BAD_ARG, "dim must be in 1..8");
    for (uint32_t k = 0; k < dim; ++k) {
        double lo = bounds[2 * k], hi = bounds[2 * k + 1];
        if (!std::isfinite(lo) || !std::isfinite(hi))  // real_vector_state_space.rs:239-241
            return fail(OXHIP_ERR_UNBOUNDED, "dimension " + std::to_string(k) + " is unbounded");
        if (lo >= hi) return fail(OXHIP_ERR_ZERO_VOLUME, "lower bound >= upper bound");  // rvss.rs:78-83,242-244
        if (std::fabs(lo) > kMaxMagnitude || std::fabs(hi) > kMaxMagnitude)
            return fail(OXHIP_ERR_BAD_ARG, "bounds beyond 1e150 would overflow squared distances");
    }
    // set_longest_valid_segment_fraction clamp (rvss.rs:121-129)
    if (fraction > 0.0 && fraction <= 1.0) {} else if (fraction <= 0.0) fraction = 0.0; else fraction = 1.0;
    // get_maximum_extent (rvss.rs:103-118): sequential sum of squared widths, sqrt
    double acc = 0.0;
    for (uint32_t k = 0; k < dim; ++k) {
        double w = bounds[2 * k + 1] - bounds[2 * k];
        double sq = w * w;
        acc = acc + sq;
    }
    double extent = std::sqrt(acc);
    double lvsl = extent * fraction;  // rvss.rs:251-253
    res = lvsl * 0.1;                 // rrt.rs:97
    if (!(res > 0.0)) return fail(OXHIP_ERR_BAD_ARG, "longest valid segment length is 0: check_motion would never terminate");
    return OXHIP_OK;
}

}  // namespace oxhip
