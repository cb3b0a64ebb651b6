// lane_query_common.hpp -- small device helpers shared by the lane-per-query kernels (rrt_lanes.hip, rrt_cells.hip):
// bit casts, in-place minima, wave-wide 64-bit sums, the two-instruction screen verdict, lane masks.
#pragma once

#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"

namespace oxhip {

typedef float lf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lf32_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float lbits_f32(uint32_t v) { return __builtin_bit_cast(float, v); }

__device__ __forceinline__ void vmin_f32(float& acc, float x) {   // plain v_min_f32 in place: no canonicalising v_max in front,
    asm("v_min_f32 %0, %0, %1" : "+v"(acc) : "v"(x));             // and no renamed register to copy back where branches join
}
__device__ __forceinline__ void vmin3_f32(float& acc, float x, float y) {   // acc = min(acc, x, y) in one issue slot (never NaN here)
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
}

// wave-wide sum of a 64-bit value (mod 2^64); every lane of the last row holds it, lane 63 is read
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_add_step(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + (((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v = dpp_add_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_add_step<0x140, 0xf>(v);   // row_mirror: every lane holds its row's sum
    v = dpp_add_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v = dpp_add_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3: lane 63 holds the total
    return uni64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }
__device__ __forceinline__ double wave_max_f64pos(double v) {   // v >= +0 (or NaN -> treated as huge): bit patterns order like values
    const uint32_t hi = wave_max_u32((uint32_t)__double2hiint(v));
    const uint32_t lo = wave_max_u32((uint32_t)__double2hiint(v) == hi ? (uint32_t)__double2loint(v) : 0u);
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ float f32_up(double x) {   // a binary32 value >= x (two ulps of slack; NaN stays NaN, +inf stays +inf)
    const float t = (float)x;
    return t + fabsf(t) * 0x1p-22f + 1e-37f;
}

// bits <- 2 bits + [!(sp > thr)]: a screen verdict shifted into a lane's bit string in two instructions (compare into vcc,
// add-with-carry of the string to itself); NaN counts as "look".  After eight calls verdict t sits at bit 7 - t.
__device__ __forceinline__ void screen_bit(uint32_t& bits, float sp, float thr) {
    asm("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(sp), "v"(thr) : "vcc");
}
__device__ __forceinline__ uint32_t rev8(uint32_t bits) { return __brev(bits) >> 24; }   // verdict t back at bit t
__device__ __forceinline__ uint64_t below_mask(uint32_t lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint64_t first_n_mask(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }


struct LMargins {
    double e2;        // 2E: twice the bound on |s' + |b|^2 - d^2| of the dot-product screen (E = u H^2 D (3D + 9), rrt_lanes.hip)
    bool usable;      // H small enough for binary32 products
};

}  // namespace oxhip
