// prm_kernels.hip -- PRM roadmap construction and query on gfx950 (wave64).
//
// Reference: oxmpl/src/geometric/planners/prm.rs
//   :96-154  construct_roadmap   sample -> is_valid -> for every earlier milestone i (ascending):
//                                distance(q, m_i) < connection_radius && check_motion(q, m_i) -> edge
//   :161-187 check_motion        (same discretisation as rrt.rs:90-116)
//   :249-264 solve               start connections and goal milestones (the BFS itself is host code)
//
// The reference interleaves sampling and connecting, but nothing it samples depends on the roadmap:
// the milestone sequence is a function of the RNG stream and the validity field alone, and the edge set
// is { (j, i) : i < j, distance < r, check_motion(m_j -> m_i) }, every node's `edges` list ending up in
// ascending order.  That makes construction three data-parallel phases:
//   1. prm_sample_kernel   ordered compaction of the valid samples of the ChaCha12 stream
//   2. prm_pairs_kernel    all pairs (j, i<j): 4 register-resident j per thread, the i side streamed through
//                          the scalar cache as VALU scalar operands; f64 VALU bound (3*dim-1 flops + 1 compare
//                          per pair); emits the sparse candidates
//   3. prm_edge_kernel     check_motion(m_j -> m_i) per candidate; emits both directed keys (u<<32 | v)
// followed by one radix sort of the keys (rocPRIM) and a CSR extraction, which yields each node's
// neighbours in ascending order -- exactly the reference's `edges` vectors.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>

#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "motion_seq.hpp"

namespace oxhip {

// ------------------------------------------------------------------------------------------------
// 1. sampling: one workgroup walks the stream in batches of kSampleBatch samples

constexpr int kSampleBatch = 512;

// u64 word `a` of the (seed, stream) ChaCha12 stream, computed from scratch (slow path only)
__device__ inline uint64_t stream_word(uint64_t seed, uint64_t stream, uint64_t a) {
    uint32_t o[16];
    chacha12_block(seed, a >> 3, stream, o);
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w)
        if ((uint32_t)(a & 7) == (uint32_t)w) { lo = o[2 * w]; hi = o[2 * w + 1]; }
    return (hi << 32) | lo;
}

template <int DIM>
__global__ __launch_bounds__(kSampleBatch) void prm_sample_kernel(DevParams p, PrmArgs a) {
    // words of the batch: kSampleBatch * DIM consecutive u64 = at most kSampleBatch*DIM/8 + 1 blocks
    constexpr int kBlocks = kSampleBatch * DIM / 8 + 2;
    __shared__ uint32_t wbuf[kBlocks][16];
    __shared__ double smp[kSampleBatch][DIM];   // redraw path: the batch as drawn sequentially
    __shared__ uint32_t cw[kSampleBatch];       // redraw path: words consumed up to and including sample s
    __shared__ uint32_t wave_cnt[kSampleBatch / 64];
    __shared__ uint32_t sh_last, sh_any;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PrmState st = *a.state;
    uint32_t n = st.n_milestones;
    uint64_t pos = st.draws, ns = st.n_samples;
    uint32_t redraws = st.redraw_batches;
    while (n < a.n_target && ns < a.max_samples) {
        const uint64_t left = a.max_samples - ns;
        const uint32_t m = left < (uint64_t)kSampleBatch ? (uint32_t)left : (uint32_t)kSampleBatch;
        // ---- the batch's blocks
        const uint64_t blk0 = pos >> 3;
        for (uint32_t b = tid; b < (uint32_t)kBlocks; b += kSampleBatch) {
            uint32_t o[16];
            chacha12_block(p.seed, blk0 + b, a.stream, o);
#pragma unroll
            for (int w = 0; w < 16; ++w) wbuf[b][w] = o[w];
        }
        if (tid == 0) { sh_last = 0xFFFFFFFFu; sh_any = 0; }
        __syncthreads();
        auto word = [&](uint64_t abs_word) -> uint64_t {
            const uint64_t rel = abs_word - (blk0 << 3);
            if (rel < (uint64_t)kBlocks * 8) {
                const uint32_t b = (uint32_t)(rel >> 3), w = (uint32_t)(rel & 7) * 2;
                return ((uint64_t)wbuf[b][w + 1] << 32) | wbuf[b][w];
            }
            return stream_word(p.seed, a.stream, abs_word);
        };
        // ---- every thread draws its sample as if no draw of the batch were rejected (rvss.rs:233-249)
        double q[DIM];
        bool redraw = false;
        if (tid < m) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                const uint64_t bits = (word(pos + (uint64_t)tid * DIM + k) >> 12) | 0x3FF0000000000000ull;
                const double v01 = __longlong_as_double((long long)bits) - 1.0;
                double res = v01 * p.scale[k];
                res = res + p.lo[k];
                redraw = redraw || !(res < p.hi[k]);
                q[k] = res;
            }
        }
        if (redraw) sh_any = 1;
        __syncthreads();
        const bool any_redraw = sh_any != 0;
        if (any_redraw) {
            // rand's sample_single loop rejected a draw somewhere in this batch: later samples start at
            // shifted stream positions.  Thread 0 replays the batch in order (rare: p ~ 2^-52 per draw).
            if (tid == 0) {
                uint64_t wp = pos;
                for (uint32_t s = 0; s < m; ++s) {
                    for (int k = 0; k < DIM; ++k) {
                        double res;
                        for (;;) {
                            const uint64_t bits = (word(wp++) >> 12) | 0x3FF0000000000000ull;
                            const double v01 = __longlong_as_double((long long)bits) - 1.0;
                            res = v01 * p.scale[k];
                            res = res + p.lo[k];
                            if (res < p.hi[k]) break;
                        }
                        smp[s][k] = res;
                    }
                    cw[s] = (uint32_t)(wp - pos);
                }
            }
            __syncthreads();
            if (tid < m) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) q[k] = smp[tid][k];
            }
            ++redraws;
        }
        // ---- is_valid (prm.rs:123) and ordered compaction
        const bool valid = tid < m && state_valid_seq<DIM>(p, q);
        const uint64_t bal = __ballot(valid);
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kSampleBatch / 64; ++w) {
            const uint32_t c = wave_cnt[w];
            before += (uint32_t)w < wave ? c : 0u;
            total += c;
        }
        const uint32_t rank = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const bool keep = valid && n + rank < a.n_target;
        if (keep) {
#pragma unroll
            for (int k = 0; k < DIM; ++k) a.ms[(size_t)(n + rank) * DIM + k] = q[k];
            if (n + rank == a.n_target - 1) sh_last = tid;  // the sample that completes the roadmap
        }
        __syncthreads();
        // samples after the completing one were never drawn by the reference
        const uint32_t consumed = sh_last != 0xFFFFFFFFu ? sh_last + 1 : m;
        pos += any_redraw ? (uint64_t)cw[consumed - 1] : (uint64_t)consumed * DIM;
        ns += consumed;
        n = n + total < a.n_target ? n + total : a.n_target;
        __syncthreads();  // wbuf / smp / sh_* are rewritten by the next batch
    }
    if (tid == 0) {
        st.n_milestones = n;
        st.draws = pos;
        st.n_samples = ns;
        st.redraw_batches = redraws;
        *a.state = st;
    }
}

// ---- the parallel sampler.  A rejected draw is a ~2^-52 event per coordinate, so a round of M samples is
// first drawn as if none happens: sample s sits at stream word pos0 + s * DIM.  Any rejection anywhere
// in the round raises a flag and the host replays that round with the sequential kernel above.
constexpr int kSpecThreads = 256;

template <int DIM>
__global__ __launch_bounds__(kSpecThreads) void prm_sample_spec_kernel(DevParams p, PrmArgs a, PrmSpec sp) {
    constexpr int kBlocks = kSpecThreads * DIM / 8 + 2;
    __shared__ uint32_t wbuf[kBlocks][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t s0 = blockIdx.x * kSpecThreads;           // first sample of this workgroup (round-relative)
    const uint64_t w0 = sp.pos0 + (uint64_t)s0 * DIM;        // its first stream word
    const uint64_t blk0 = w0 >> 3;
    for (uint32_t b = tid; b < (uint32_t)kBlocks; b += kSpecThreads) {
        uint32_t o[16];
        chacha12_block(p.seed, blk0 + b, a.stream, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) wbuf[b][w] = o[w];
    }
    __syncthreads();
    const uint32_t s = s0 + tid;
    const bool act = s < sp.m;
    double q[DIM];
    bool redraw = false;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const uint32_t rel = (uint32_t)(w0 - (blk0 << 3)) + tid * DIM + k;   // < kBlocks * 8
        const uint64_t word = ((uint64_t)wbuf[rel >> 3][(rel & 7) * 2 + 1] << 32) | wbuf[rel >> 3][(rel & 7) * 2];
        const uint64_t bits = (word >> 12) | 0x3FF0000000000000ull;
        const double v01 = __longlong_as_double((long long)bits) - 1.0;
        double res = v01 * p.scale[k];
        res = res + p.lo[k];
        redraw = redraw || !(res < p.hi[k]);
        q[k] = res;
    }
    if (__ballot(act && redraw) != 0 && lane == 0) atomicOr(sp.redraw_flag, 1u);
    const bool valid = act && state_valid_seq<DIM>(p, q);                      // prm.rs:123
    const uint64_t bal = __ballot(valid);
    if (valid) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) sp.tmp[(size_t)s * DIM + k] = q[k];
    }
    if (lane == 0) {
        sp.vbits[s >> 6] = bal;
        sp.wave_off[s >> 6] = (uint32_t)__popcll(bal);    // counts now, exclusive offsets after the scan kernel
    }
}

// one workgroup: exclusive scan of the per-wave counts, the sample that completes the roadmap, the new state
__global__ __launch_bounds__(1024) void prm_sample_scan_kernel(PrmArgs a, PrmSpec sp, uint32_t dim) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t sh_consumed;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nw = (sp.m + 63) >> 6;
    const uint32_t per = (nw + 1023) / 1024;
    const uint32_t b = tid * per, e = b + per < nw ? b + per : nw;
    uint32_t mine = 0;
    for (uint32_t w = b; w < e; ++w) mine += sp.wave_off[w];
    // inclusive scan across the workgroup
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if ((int)lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    if (tid == 0) sh_consumed = sp.m;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        base += (uint32_t)w < wave ? wsum[w] : 0u;
        total += wsum[w];
    }
    uint32_t run = base + inc - mine;   // exclusive prefix of this thread's first wave
    const PrmState st = *a.state;
    const uint32_t need = a.n_target - st.n_milestones;   // > 0: the host only launches an incomplete round
    for (uint32_t w = b; w < e; ++w) {
        const uint32_t c = sp.wave_off[w];
        sp.wave_off[w] = run;
        if (run < need && run + c >= need) {
            // the (need - run)-th valid sample of this wave completes the roadmap: nothing after it is drawn
            uint64_t bits = sp.vbits[w];
            for (uint32_t k = 1; k < need - run; ++k) bits &= bits - 1;
            sh_consumed = w * 64 + (uint32_t)(__ffsll((unsigned long long)bits) - 1) + 1;
        }
        run += c;
    }
    __syncthreads();
    if (tid == 0) {
        PrmState ns = st;
        const uint32_t consumed = sh_consumed;
        ns.n_milestones = total >= need ? a.n_target : st.n_milestones + total;
        ns.n_samples = st.n_samples + consumed;
        ns.draws = st.draws + (uint64_t)consumed * dim;
        sp.result[0] = ns;            // committed by the host only when no draw of the round was rejected
    }
}

// ordered compaction of the round's valid samples behind the existing milestones
template <int DIM>
__global__ __launch_bounds__(kSpecThreads) void prm_sample_compact_kernel(PrmArgs a, PrmSpec sp, uint32_t n0) {
    const uint32_t s = blockIdx.x * kSpecThreads + threadIdx.x, lane = threadIdx.x & 63;
    if (s >= sp.m) return;
    const uint64_t bal = sp.vbits[s >> 6];
    if (!((bal >> lane) & 1ull)) return;
    const uint32_t dst = n0 + sp.wave_off[s >> 6] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (dst >= a.n_target) return;
#pragma unroll
    for (int k = 0; k < DIM; ++k) a.ms[(size_t)dst * DIM + k] = sp.tmp[(size_t)s * DIM + k];
}

// ------------------------------------------------------------------------------------------------
// 2. all pairs within the connection radius

constexpr int kPairThreads = 256;
#ifndef OXHIP_PAIR_R
#define OXHIP_PAIR_R 4
#endif
#ifndef OXHIP_PAIR_IC
#define OXHIP_PAIR_IC 256
#endif
// Tile shape (j per thread, i per workgroup), measured at 50,000 milestones in R^6 on one MI355X:
// (4,2048) 1.32 ms, (4,1024) 0.87, (4,512) 0.81, (4,256) 0.79, (4,128) 0.76, (2,512) 0.77, (2,256) 0.75 --
// the triangular grid wants many small workgroups; 0.75 ms is the chip's sustained f64 VALU rate (~30 T op/s).
constexpr int kPairR = OXHIP_PAIR_R;                // milestones j held in registers per thread
constexpr int kPairJB = kPairThreads * kPairR;      // j per workgroup
constexpr int kPairIC = OXHIP_PAIR_IC;              // i per workgroup

constexpr int kStage = 512;                         // per-wave LDS staging of hits before they go to HBM

// Hits are appended to a per-wave LDS buffer with ballot / prefix-count positions (the wave is its only
// writer, so no atomic is involved) and flushed to the candidate list with ONE global atomic per flush.
// A single global counter bumped once per hit serialises at the L2: 291 k hits cost 3 ms that way.
typedef float pair_f32x2 __attribute__((ext_vector_type(2)));

// host twin of screen_margins / screen_threshold (rrt_device.hpp): the largest binary32 squared distance a pair whose
// binary64 d2 is <= thr can show, given the magnitude bound m of the coordinates; +inf switches the screen off
static float host_screen_threshold(double m_all, int dim, double thr) {
    const double m = m_all * 1.001, u = 0x1p-24;
    if (!(m < 1e15) || !(thr >= 0.0)) return std::numeric_limits<float>::infinity();
    const double r = std::sqrt(thr);
    if (!(r < 1e18)) return std::numeric_limits<float>::infinity();
    const double a2 = 2.0 * (std::sqrt((double)dim) * 4.1 * u * m + 1e-18);
    const double r_hi = 1.0 + 2.0 * (0x1p-19 + (double)(dim + 2) * u);
    const double d = (r * (1.0 + 1e-12) + a2) * r_hi * r_hi;
    return (float)(d * d * (1.0 + 0x1p-20));
}

struct PairStage {
    uint2* buf;       // this wave's kStage entries in LDS
    uint32_t cnt;     // wave-uniform
};

__device__ __forceinline__ void stage_flush(const PrmArgs& a, PairStage& st, uint32_t lane) {
    if (st.cnt == 0) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&a.state->n_cand, (unsigned long long)st.cnt);
    base = uni64(base);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's LDS writes precede its reads below
    for (uint32_t e = lane; e < st.cnt; e += 64) {
        const unsigned long long slot = base + e;
        if (slot < (unsigned long long)a.cand_cap) a.cand[slot] = st.buf[e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and those reads precede the next writes
    st.cnt = 0;
}

// A range of i against the thread's kPairR milestones.  `ci` is wave-uniform: it is loaded through the scalar
// cache (s_load) and feeds the f64 VALU as a scalar operand, so the inner loop touches neither LDS nor
// the vector memory path.  Hits are rare (a few 1e-4 of all pairs): one wave-uniform branch per i.
template <int DIM, bool DIAG>
__device__ __forceinline__ void pairs_one_i(const PrmArgs& a, PairStage& st, uint32_t lane, const double (&cj)[kPairR][DIM],
                                            const pair_f32x2 (&cj32)[kPairR / 2][DIM], const uint32_t (&jr)[kPairR], const float (&ci)[DIM],
                                            uint32_t i, const double (&thr)[kPairR], const float (&thr32)[kPairR], const double* __restrict__ ms) {
    // binary32 SCREEN of distance(q_rand, other)^2 for the thread's two milestones at once (packed: v_pk_add / v_pk_fma):
    // a pair whose screened value exceeds thr32 cannot pass the reference's test (screen_threshold, rrt_device.hpp) ...
    static_assert(kPairR % 2 == 0, "the milestones of a thread share packed registers two by two");
    pair_f32x2 s[kPairR / 2];
#pragma unroll
    for (int r2 = 0; r2 < kPairR / 2; ++r2) {
        const pair_f32x2 e = cj32[r2][0] - ci[0];
        s[r2] = e * e;
    }
#pragma unroll
    for (int k = 1; k < DIM; ++k) {
#pragma unroll
        for (int r2 = 0; r2 < kPairR / 2; ++r2) {
            const pair_f32x2 e = cj32[r2][k] - ci[k];
            s[r2] = __builtin_elementwise_fma(e, e, s[r2]);
        }
    }
    bool h[kPairR];
    bool any = false;
#pragma unroll
    for (int r = 0; r < kPairR; ++r) {
        h[r] = !(s[r / 2][r % 2] > thr32[r]);   // "cannot be excluded": a NaN (inf - inf when the screen is switched off) passes
        if (DIAG) h[r] = h[r] && i < jr[r];
        any = any || h[r];
    }
    if (__ballot(any) != 0) {
        // ... and the few that pass it get the reference's own arithmetic: distance(q_rand, other)^2 in binary64
        // (rvss.rs:137-155: sequential sum over k), sqrt(d2) < connection_radius decided exactly as d2 <= thr
        double cif[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) cif[k] = ms[(size_t)i * DIM + k];
#pragma unroll
        for (int r = 0; r < kPairR; ++r) {
            if (h[r]) {
                const double d0 = cj[r][0] - cif[0];
                double acc = d0 * d0;
#pragma unroll
                for (int k = 1; k < DIM; ++k) {
                    double d = cj[r][k] - cif[k];
                    d = d * d;
                    acc = acc + d;
                }
                h[r] = acc <= thr[r];
            }
        }
#pragma unroll
        for (int r = 0; r < kPairR; ++r) {
            const uint64_t m = __ballot(h[r]);
            if (h[r]) st.buf[st.cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(jr[r], i);
            st.cnt += (uint32_t)__popcll(m);
        }
        if (st.cnt > (uint32_t)(kStage - 64 * kPairR)) stage_flush(a, st, lane);  // room for one more full i
    }
}

// i in [lo, hi), two per trip with two scalar register sets: the coordinates of one i are in flight
// (s_load) while the other's arithmetic runs, so the scalar-cache latency never shows.
template <int DIM, bool DIAG>
__device__ __forceinline__ void pairs_range(const PrmArgs& a, PairStage& st, uint32_t lane, const double (&cj)[kPairR][DIM],
                                            const pair_f32x2 (&cj32)[kPairR / 2][DIM], const uint32_t (&jr)[kPairR],
                                            const double* __restrict__ ms, const float* __restrict__ ms32, uint32_t lo,
                                            uint32_t hi, const double (&thr)[kPairR], const float (&thr32)[kPairR]) {
    if (lo >= hi) return;
    float ca[DIM], cb[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) ca[k] = ms32[(size_t)lo * DIM + k];
    for (uint32_t i = lo; i < hi; i += 2) {
        // Scalar loads return out of order, so the only wait is "all of them" (lgkmcnt(0)): each set is
        // requested right before the other set's arithmetic and waited for right after it.
        const uint32_t ib = i + 1 < hi ? i + 1 : i;
#pragma unroll
        for (int k = 0; k < DIM; ++k) cb[k] = ms32[(size_t)ib * DIM + k];
        __builtin_amdgcn_sched_barrier(0);
        pairs_one_i<DIM, DIAG>(a, st, lane, cj, cj32, jr, ca, i, thr, thr32, ms);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t ia = i + 2 < hi ? i + 2 : i;
#pragma unroll
        for (int k = 0; k < DIM; ++k) ca[k] = ms32[(size_t)ia * DIM + k];
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < hi) pairs_one_i<DIM, DIAG>(a, st, lane, cj, cj32, jr, cb, i + 1, thr, thr32, ms);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int DIM>
__global__ __launch_bounds__(kPairThreads) void prm_pairs_kernel(PrmArgs a, const double* __restrict__ ms,
                                                                  const float* __restrict__ ms32, uint32_t j0, uint32_t j1,
                                                                  double thr_all, float thr32_all, const double* __restrict__ thr_row,
                                                                  const float* __restrict__ thr32_row) {
    __shared__ uint2 stage[kPairThreads / 64][kStage];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t jb0 = j0 + blockIdx.y * kPairJB;                 // first j of this block
    const uint32_t jb1 = jb0 + kPairJB < j1 ? jb0 + kPairJB : j1;   // one past its last j
    const uint32_t i_lo = blockIdx.x * kPairIC;
    if (jb0 >= j1 || i_lo + 1 >= jb1) return;                        // no pair (j, i) with i < j in this block
    const uint32_t i_hi = i_lo + kPairIC < jb1 - 1 ? i_lo + kPairIC : jb1 - 1;  // i <= jb1 - 2
    // my milestones; a slot beyond the range holds +inf: every d2 is +inf and never <= thr
    double cj[kPairR][DIM];
    uint32_t jr[kPairR];
#pragma unroll
    for (int r = 0; r < kPairR; ++r) {
        jr[r] = jb0 + r * kPairThreads + tid;
#pragma unroll
        for (int k = 0; k < DIM; ++k) cj[r][k] = jr[r] < jb1 ? ms[(size_t)jr[r] * DIM + k] : __builtin_inf();
    }
    pair_f32x2 cj32[kPairR / 2][DIM];   // fl32 of the milestones, two per packed register (+inf stays +inf)
#pragma unroll
    for (int r2 = 0; r2 < kPairR / 2; ++r2)
#pragma unroll
        for (int k = 0; k < DIM; ++k) cj32[r2][k] = pair_f32x2{(float)cj[2 * r2][k], (float)cj[2 * r2 + 1][k]};
    // the radius rule has one threshold for every pair; the k-nearest variant's candidate search gives every row j its own
    // (thr_row[j]: a radius expected to hold several times k earlier milestones)
    double thr[kPairR];
    float thr32[kPairR];
#pragma unroll
    for (int r = 0; r < kPairR; ++r) {
        const bool row = thr_row != nullptr && jr[r] < jb1;
        thr[r] = row ? thr_row[jr[r]] : thr_all;
        thr32[r] = row ? thr32_row[jr[r]] : thr32_all;
    }
    PairStage st{stage[tid >> 6], 0u};
    // i below every j of the block: no index test; the rest of the range (the diagonal blocks) tests i < j
    const uint32_t i_mid = i_hi < jb0 ? i_hi : (i_lo > jb0 ? i_lo : jb0);
    pairs_range<DIM, false>(a, st, lane, cj, cj32, jr, ms, ms32, i_lo, i_mid, thr, thr32);
    pairs_range<DIM, true>(a, st, lane, cj, cj32, jr, ms, ms32, i_mid, i_hi, thr, thr32);
    stage_flush(a, st, lane);
}

// ------------------------------------------------------------------------------------------------
// 3. check_motion per candidate pair (from = the newer milestone j, to = the older one i: prm.rs:134)

template <int DIM>
__global__ __launch_bounds__(256) void prm_edge_kernel(DevParams p, PrmArgs a, uint32_t n_cand, uint32_t key_shift) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    bool ok = false;
    uint2 pr = make_uint2(0u, 0u);
    if (c < n_cand) {
        pr = a.cand[c];
        double from[DIM], to[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            from[k] = a.ms[(size_t)pr.x * DIM + k];
            to[k] = a.ms[(size_t)pr.y * DIM + k];
        }
        ok = motion_valid_seq<DIM>(p, from, to);
    }
    // one atomic per wave: a counter bumped once per edge serialises at the L2
    const uint64_t bal = __ballot(ok);
    if (bal == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.state->n_keys, 2u * (uint32_t)__popcll(bal));
    base = uni(base);
    if (ok) {
        const uint32_t slot = base + 2u * (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        a.keys[slot] = ((uint64_t)pr.x << key_shift) | pr.y;       // i in j's list
        a.keys[slot + 1] = ((uint64_t)pr.y << key_shift) | pr.x;   // j in i's list (prm.rs:143-145)
    }
}

// ------------------------------------------------------------------------------------------------
// 3b. the k-NEAREST variant (BASELINE.json configs[4]: "all-pairs k-NN"; the reference connects by radius): milestone j's
// candidates are the k earlier milestones nearest to it by (distance, index) instead of all within the radius.  The pair search
// above runs with a per-row radius expected to hold several times k earlier milestones; its hits, sorted by (j, i), are cut down
// to each row's k nearest here -- by the reference's own distance (sqrt of the sequential sum, rvss.rs:137-155), the lower index
// first among equal distances -- and a row whose radius held fewer than k gets the exact search over all earlier milestones.
__global__ __launch_bounds__(256) void prm_cand_keys_kernel(const uint2* __restrict__ cand, uint32_t n, uint32_t shift, uint64_t* __restrict__ keys) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) keys[c] = ((uint64_t)cand[c].x << shift) | cand[c].y;
}

template <int DIM>
__device__ __forceinline__ double prm_pair_distance(const double* __restrict__ ms, uint32_t j, uint32_t i) {
    const double d0 = ms[(size_t)j * DIM] - ms[(size_t)i * DIM];   // distance(q_rand, other): the new milestone first
    double acc = d0 * d0;
#pragma unroll
    for (int k = 1; k < DIM; ++k) {
        double d = ms[(size_t)j * DIM + k] - ms[(size_t)i * DIM + k];
        d = d * d;
        acc = acc + d;
    }
    return sqrt(acc);
}

template <int DIM>
__global__ __launch_bounds__(256) void prm_knn_dist_kernel(PrmArgs a, const uint64_t* __restrict__ sorted, uint32_t n, uint32_t shift,
                                                            double* __restrict__ dist) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const uint64_t key = sorted[c];
    dist[c] = prm_pair_distance<DIM>(a.ms, (uint32_t)(key >> shift), (uint32_t)(key & ((1ull << shift) - 1ull)));
}

// one thread per row: its segment of the sorted candidates, k passes of "the next (distance, index) after the last one taken".
// counters[0] = pairs selected so far (one atomic per wave), counters[1] = rows whose segment holds fewer than min(k, j) pairs
__global__ __launch_bounds__(256) void prm_knn_select_kernel(const uint64_t* __restrict__ sorted, const double* __restrict__ dist, uint32_t n_sorted,
                                                              uint32_t shift, uint32_t j0, uint32_t j1, uint32_t k, uint2* __restrict__ sel,
                                                              uint32_t* counters, uint32_t* __restrict__ failed_rows) {
    const uint32_t j = j0 + blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool row = j < j1 && j > 0;
    const uint32_t need = row ? (k < j ? k : j) : 0u;
    uint32_t lo = 0, hi = 0;
    if (row) {
        const uint64_t k0 = (uint64_t)j << shift, k1 = (uint64_t)(j + 1u) << shift;
        uint32_t a0 = 0, b0 = n_sorted;
        while (a0 < b0) { const uint32_t mid = a0 + ((b0 - a0) >> 1); if (sorted[mid] < k0) a0 = mid + 1; else b0 = mid; }
        lo = a0;
        b0 = n_sorted;
        while (a0 < b0) { const uint32_t mid = a0 + ((b0 - a0) >> 1); if (sorted[mid] < k1) a0 = mid + 1; else b0 = mid; }
        hi = a0;
    }
    const bool failed = row && hi - lo < need;
    if (failed) failed_rows[atomicAdd(&counters[1], 1u)] = j;
    const uint32_t take = failed ? 0u : need;
    // wave prefix of `take`, one atomic per wave
    uint32_t pre = take;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)pre, d, 64);
        if (lane >= (uint32_t)d) pre += o;
    }
    const uint32_t total = (uint32_t)__shfl((int)pre, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(&counters[0], total);
    base = (uint32_t)__shfl((int)base, 63, 64) + pre - take;
    const uint64_t mask = (1ull << shift) - 1ull;
    double last_d = -1.0;
    uint32_t last_i = 0;
    bool first = true;
    for (uint32_t t = 0; t < take; ++t) {
        double best_d = __builtin_inf();
        uint32_t best_i = 0xFFFFFFFFu;
        for (uint32_t c = lo; c < hi; ++c) {
            const double d = dist[c];
            const uint32_t i = (uint32_t)(sorted[c] & mask);
            const bool after = first || d > last_d || (d == last_d && i > last_i);
            const bool better = d < best_d || (d == best_d && i < best_i);
            if (after && better) { best_d = d; best_i = i; }
        }
        sel[base + t] = make_uint2(j, best_i);
        last_d = best_d;
        last_i = best_i;
        first = false;
    }
}

// one wave per row the radius failed: the exact k nearest among ALL earlier milestones, k rounds of the lexicographic minimum
template <int DIM>
__global__ __launch_bounds__(64) void prm_knn_brute_kernel(PrmArgs a, const uint32_t* __restrict__ failed_rows, uint32_t k, uint2* __restrict__ sel,
                                                            uint32_t* counters) {
    const uint32_t j = failed_rows[blockIdx.x], lane = threadIdx.x;
    const uint32_t need = k < j ? k : j;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[0], need);
    base = uni(base);
    double last_d = -1.0;
    uint32_t last_i = 0;
    for (uint32_t t = 0; t < need; ++t) {
        Exact e{__builtin_inf(), 0xFFFFFFFFu};
        for (uint32_t i = lane; i < j; i += 64) {
            const double d = prm_pair_distance<DIM>(a.ms, j, i);
            const bool after = t == 0 || d > last_d || (d == last_d && i > last_i);
            if (after && (d < e.dist || (d == e.dist && i < e.idx))) { e.dist = d; e.idx = i; }
        }
        e = exact_wave_reduce(e);
        if (lane == 0) sel[base + t] = make_uint2(j, e.idx);
        last_d = unid(e.dist);
        last_i = uni(e.idx);
    }
}

// CSR from the sorted directed keys (u << key_shift | v): offsets by binary search, neighbours = low fields
__global__ __launch_bounds__(256) void prm_csr_kernel(const uint64_t* sorted, uint32_t n_keys, uint32_t n_nodes,
                                                       uint32_t key_shift, uint32_t* offsets, uint32_t* nbrs) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_keys) nbrs[t] = (uint32_t)(sorted[t] & ((1ull << key_shift) - 1ull));
    if (t <= n_nodes) {
        const uint64_t key = (uint64_t)t << key_shift;
        uint32_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (sorted[mid] < key) lo = mid + 1; else hi = mid;
        }
        offsets[t] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// query (prm.rs:243-264): start validity, start connections, goal milestones

template <int DIM>
__global__ __launch_bounds__(256) void prm_query_kernel(DevParams p, PrmArgs a, uint32_t n, PrmQuery q, double thr,
                                                         uint8_t* flags, uint32_t* start_valid) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    double s[DIM], g[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) { s[k] = q.start[k]; g[k] = q.goal_c[k]; }
    if (i == 0) *start_valid = state_valid_seq<DIM>(p, s) ? 1u : 0u;   // prm.rs:244
    if (i >= n) return;
    double m[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) m[k] = a.ms[(size_t)i * DIM + k];
    uint8_t f = 0;
    if (dist2<DIM>(s, m, DIM) <= thr && motion_valid_seq<DIM>(p, s, m)) f |= 1;   // prm.rs:251-252
    if (dist2<DIM>(m, g, DIM) <= q.goal_thr) f |= 2;                               // goal.is_satisfied, prm.rs:261
    flags[i] = f;
}

// ------------------------------------------------------------------------------------------------
// launchers

template <typename F>
static void dim_dispatch(uint32_t dim, F&& f) {
    switch (dim) {
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        case 5: f(std::integral_constant<int, 5>{}); break;
        case 6: f(std::integral_constant<int, 6>{}); break;
        case 7: f(std::integral_constant<int, 7>{}); break;
        default: f(std::integral_constant<int, 8>{}); break;
    }
}

void launch_prm_sample(const DevParams& p, const PrmArgs& a, hipStream_t s) {
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_sample_kernel<D>, dim3(1), dim3(kSampleBatch), 0, s, p, a);
    });
}

void launch_prm_sample_spec(const DevParams& p, const PrmArgs& a, const PrmSpec& sp, uint32_t n0, hipStream_t s) {
    const uint32_t blocks = (sp.m + kSpecThreads - 1) / kSpecThreads;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_sample_spec_kernel<D>, dim3(blocks), dim3(kSpecThreads), 0, s, p, a, sp);
    });
    hipLaunchKernelGGL(prm_sample_scan_kernel, dim3(1), dim3(1024), 0, s, a, sp, p.dim);
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_sample_compact_kernel<D>, dim3(blocks), dim3(kSpecThreads), 0, s, a, sp, n0);
    });
}

__global__ void prm_shadow_kernel(const double* __restrict__ ms, float* __restrict__ ms32, uint64_t first, uint64_t last) {
    const uint64_t i = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < last) ms32[i] = (float)ms[i];
}

float prm_screen_threshold(const DevParams& p, double thr) {
    double m = 0.0;
    for (uint32_t k = 0; k < p.dim; ++k) m = std::fmax(m, std::fmax(std::fabs(p.lo[k]), std::fabs(p.hi[k])));
    return host_screen_threshold(m, (int)p.dim, thr);
}

void launch_prm_pairs(const DevParams& p, const PrmArgs& a, uint32_t j0, uint32_t j1, double thr, hipStream_t s, const double* thr_row,
                      const float* thr32_row) {
    if (j1 <= j0 || j1 < 2) return;
    // fl32 shadow of the new milestones (the older ones have theirs), then the screen threshold: milestones are samples
    // inside the bounds, so the magnitude bound M of the error model (rrt_device.hpp) is the bounds'
    const uint64_t first = (uint64_t)j0 * p.dim, last = (uint64_t)j1 * p.dim;
    hipLaunchKernelGGL(prm_shadow_kernel, dim3((uint32_t)((last - first + 255) / 256)), dim3(256), 0, s, (const double*)a.ms,
                       a.ms32, first, last);
    double m = 0.0;
    for (uint32_t k = 0; k < p.dim; ++k) m = std::fmax(m, std::fmax(std::fabs(p.lo[k]), std::fabs(p.hi[k])));
    const float thr32 = host_screen_threshold(m, (int)p.dim, thr);
    const uint32_t jblocks = (j1 - j0 + kPairJB - 1) / kPairJB;
    const uint32_t ichunks = (j1 - 1 + kPairIC - 1) / kPairIC;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_pairs_kernel<D>, dim3(ichunks, jblocks), dim3(kPairThreads), 0, s, a, (const double*)a.ms,
                           (const float*)a.ms32, j0, j1, thr, thr32, thr_row, thr32_row);
    });
}

// directed keys are (u << shift) | v with shift = bits of (capacity - 1): the sort covers 2 * shift bits
uint32_t prm_key_shift(uint32_t cap) {
    uint32_t shift = 1;
    while (shift < 32 && ((cap - 1) >> shift) != 0) ++shift;
    return shift;
}

void launch_prm_edges(const DevParams& p, const PrmArgs& a, uint32_t n_cand, hipStream_t s) {
    if (n_cand == 0) return;
    const uint32_t shift = prm_key_shift(a.cap);
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_edge_kernel<D>, dim3((n_cand + 255) / 256), dim3(256), 0, s, p, a, n_cand, shift);
    });
}

void launch_prm_knn_keys(const PrmArgs& a, uint32_t n_cand, uint64_t* keys, hipStream_t s) {
    if (n_cand) hipLaunchKernelGGL(prm_cand_keys_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, s, (const uint2*)a.cand, n_cand, prm_key_shift(a.cap), keys);
}
void launch_prm_knn_select(const DevParams& p, const PrmArgs& a, const uint64_t* sorted, double* dist, uint32_t n_sorted, uint32_t j0, uint32_t j1,
                           uint32_t k, uint2* sel, uint32_t* counters, uint32_t* failed_rows, hipStream_t s) {
    const uint32_t shift = prm_key_shift(a.cap);
    if (n_sorted)
        dim_dispatch(p.dim, [&](auto d) {
            constexpr int D = decltype(d)::value;
            hipLaunchKernelGGL(prm_knn_dist_kernel<D>, dim3((n_sorted + 255) / 256), dim3(256), 0, s, a, sorted, n_sorted, shift, dist);
        });
    hipLaunchKernelGGL(prm_knn_select_kernel, dim3((j1 - j0 + 255) / 256), dim3(256), 0, s, sorted, (const double*)dist, n_sorted, shift, j0, j1, k,
                       sel, counters, failed_rows);
}
void launch_prm_knn_brute(const DevParams& p, const PrmArgs& a, const uint32_t* failed_rows, uint32_t n_failed, uint32_t k, uint2* sel,
                          uint32_t* counters, hipStream_t s) {
    if (n_failed == 0) return;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_knn_brute_kernel<D>, dim3(n_failed), dim3(64), 0, s, a, failed_rows, k, sel, counters);
    });
}

hipError_t prm_sort_keys(void* tmp, size_t& tmp_bytes, uint64_t* in, uint64_t* out, uint32_t n_keys, uint32_t cap,
                         hipStream_t s) {
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, (size_t)n_keys, 0u, 2u * prm_key_shift(cap), s);
}

void launch_prm_csr(const uint64_t* sorted, uint32_t n_keys, uint32_t n_nodes, uint32_t cap, uint32_t* offsets,
                    uint32_t* nbrs, hipStream_t s) {
    const uint32_t work = n_keys > n_nodes + 1 ? n_keys : n_nodes + 1;
    hipLaunchKernelGGL(prm_csr_kernel, dim3((work + 255) / 256), dim3(256), 0, s, sorted, n_keys, n_nodes, prm_key_shift(cap),
                       offsets, nbrs);
}

void launch_prm_query(const DevParams& p, const PrmArgs& a, uint32_t n, const PrmQuery& q, double thr, uint8_t* flags,
                      uint32_t* start_valid, hipStream_t s) {
    const uint32_t blocks = n ? (n + 255) / 256 : 1;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_query_kernel<D>, dim3(blocks), dim3(256), 0, s, p, a, n, q, thr, flags, start_valid);
    });
}

}  // namespace oxhip
