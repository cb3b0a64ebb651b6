// lanes_reduce.hpp -- wave-wide minima of 8 (or 4) per-lane float values at once, for the scanner waves of rrt_lanes.hip.
//
// Eight separate 6-step DPP reductions cost 48 VALU instructions.  Here every step that folds one lane bit also halves the
// number of live registers (a "transposing" butterfly): two registers A, B become one whose lanes with the bit clear hold
// min(A[l], A[partner]) and whose lanes with the bit set hold min(B[l], B[partner]).
//   bit 3 (lane ^ 8)   DPP row_shl:8 / row_shr:8 under bank masks 0x3 / 0xc      8 values -> 4 registers   8 instructions
//   bit 4 (lane ^ 16)  v_permlane16_swap_b32 + v_min_f32 (gfx950)                4 -> 2                     4
//   bit 5 (lane ^ 32)  v_permlane32_swap_b32 + v_min_f32 (gfx950)                2 -> 1                     2
//   bits 0, 1, 2       quad_perm, quad_perm, row_half_mirror on the one register                           3
// 17 instructions instead of 48; the minimum of value b ends up in the eight lanes 8b .. 8b + 7 (NQ = 8).  With four values
// (R^6 passes) bit 5 is folded by a swap against a copy: 12 instructions instead of 24, same lanes 8b .. 8b + 7 (and + 32).
// Values must not be NaN (v_min_f32 would drop them silently); +inf is fine.
// tools/reduce_check.hip compares both with a plain shuffle reduction on the device.
#pragma once
#include <hip/hip_runtime.h>

namespace oxhip {

// the explicit s_nop 1 cover "VALU writes a VGPR -> DPP / permlane-swap reads it" (two wait states): the hazard recogniser
// does not look inside inline asm
__device__ __forceinline__ float lanes_min8_transposed(const float (&v)[8]) {
    float r0, r1, r2, r3;
    asm("s_nop 1\n"
        "v_min_f32_dpp %0, %4, %4 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %1, %6, %6 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %2, %8, %8 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %3, %10, %10 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %0, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "v_min_f32_dpp %1, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "v_min_f32_dpp %2, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "v_min_f32_dpp %3, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "s_nop 1\n"
        "v_permlane16_swap_b32 %0, %1\n"
        "v_permlane16_swap_b32 %2, %3\n"
        "s_nop 1\n"
        "v_min_f32 %0, %0, %1\n"
        "v_min_f32 %2, %2, %3\n"
        "s_nop 1\n"
        "v_permlane32_swap_b32 %0, %2\n"
        "s_nop 1\n"
        "v_min_f32 %0, %0, %2\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
    return r0;   // lanes 8b .. 8b + 7: the wave's minimum of v[b]
}

__device__ __forceinline__ float lanes_min4_transposed(const float (&v)[4]) {
    float r0, r1, c;
    asm("s_nop 1\n"
        "v_min_f32_dpp %0, %3, %3 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %1, %5, %5 row_shl:8 row_mask:0xf bank_mask:0x3\n"
        "v_min_f32_dpp %0, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "v_min_f32_dpp %1, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xc\n"
        "s_nop 1\n"
        "v_permlane16_swap_b32 %0, %1\n"
        "s_nop 1\n"
        "v_min_f32 %0, %0, %1\n"
        "s_nop 1\n"
        "v_mov_b32 %2, %0\n"
        "s_nop 1\n"
        "v_permlane32_swap_b32 %0, %2\n"
        "s_nop 1\n"
        "v_min_f32 %0, %0, %2\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        : "=&v"(r0), "=&v"(r1), "=&v"(c)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return r0;   // lanes 8b .. 8b + 7 (and 32 + those): the wave's minimum of v[b]
}

template <int NQ> __device__ __forceinline__ float lanes_min_transposed(const float (&v)[NQ]);
template <> __device__ __forceinline__ float lanes_min_transposed<8>(const float (&v)[8]) { return lanes_min8_transposed(v); }
template <> __device__ __forceinline__ float lanes_min_transposed<4>(const float (&v)[4]) { return lanes_min4_transposed(v); }

}  // namespace oxhip
