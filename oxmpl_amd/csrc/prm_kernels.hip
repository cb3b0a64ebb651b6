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

#include <cstring>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>

#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

// ------------------------------------------------------------------------------------------------
// validity and motion check evaluated by one thread (the RRT kernels stripe them over a workgroup)

template <int D>
__device__ __forceinline__ bool state_valid_seq(const DevParams& p, const double s[D]) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    for (uint32_t j = 0; j < nobs; ++j)  // j is wave-uniform: the obstacle table is read through the scalar cache
        if (obstacle_hit<D>(p, D, s, j)) return false;
    return true;
}

// prm.rs:161-187
template <int D>
__device__ __forceinline__ bool motion_valid_seq(const DevParams& p, const double from[D], const double to[D]) {
    if (p.n_spheres + p.n_boxes == 0) return true;
    const double dist = sqrt(dist2<D>(from, to, D));
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    if (nsteps <= 1) return state_valid_seq<D>(p, to);
    const double dn = (double)nsteps;
    for (uint32_t step = 1; step <= nsteps; ++step) {
        const double t = (double)step / dn;
        double s[D];
        lerp<D>(from, to, t, s, D);
        if (!state_valid_seq<D>(p, s)) return false;
    }
    return true;
}

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

// ------------------------------------------------------------------------------------------------
// 2. all pairs within the connection radius

constexpr int kPairThreads = 256;
constexpr int kPairR = 4;                           // milestones j held in registers per thread
constexpr int kPairJB = kPairThreads * kPairR;      // j per workgroup
constexpr int kPairIC = 1024;                       // i per workgroup

constexpr int kStage = 512;                         // per-wave LDS staging of hits before they go to HBM

// Hits are appended to a per-wave LDS buffer with ballot / prefix-count positions (the wave is its only
// writer, so no atomic is involved) and flushed to the candidate list with ONE global atomic per flush.
// A single global counter bumped once per hit serialises at the L2: 291 k hits cost 3 ms that way.
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

// One i against the thread's kPairR milestones.  `ci` is wave-uniform: it is loaded through the scalar
// cache (s_load) and feeds the f64 VALU as a scalar operand, so the inner loop touches neither LDS nor
// the vector memory path.  Hits are rare (a few 1e-4 of all pairs): one wave-uniform branch per i.
template <int DIM, bool DIAG>
__device__ __forceinline__ void pairs_one_i(const PrmArgs& a, PairStage& st, uint32_t lane, const double (&cj)[kPairR][DIM],
                                            const uint32_t (&jr)[kPairR], const double* __restrict__ ms, uint32_t i,
                                            double thr) {
    double ci[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) ci[k] = ms[(size_t)i * DIM + k];
    bool h[kPairR];
    bool any = false;
#pragma unroll
    for (int r = 0; r < kPairR; ++r) {
        const double d2 = dist2<DIM>(cj[r], ci, DIM);   // distance(q_rand, other)^2, rvss.rs:137-155
        h[r] = d2 <= thr;                                 // sqrt(d2) < connection_radius, exactly
        if (DIAG) h[r] = h[r] && i < jr[r];
        any = any || h[r];
    }
    if (__ballot(any) != 0) {
#pragma unroll
        for (int r = 0; r < kPairR; ++r) {
            const uint64_t m = __ballot(h[r]);
            if (h[r]) st.buf[st.cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(jr[r], i);
            st.cnt += (uint32_t)__popcll(m);
        }
        if (st.cnt > (uint32_t)(kStage - 64 * kPairR)) stage_flush(a, st, lane);  // room for one more full i
    }
}

template <int DIM>
__global__ __launch_bounds__(kPairThreads) void prm_pairs_kernel(PrmArgs a, const double* __restrict__ ms, uint32_t j0,
                                                                  uint32_t j1, double thr) {
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
    PairStage st{stage[tid >> 6], 0u};
    // i below every j of the block: no index test; the rest of the range (the diagonal blocks) tests i < j
    const uint32_t i_mid = i_hi < jb0 ? i_hi : (i_lo > jb0 ? i_lo : jb0);
    for (uint32_t i = i_lo; i < i_mid; ++i) pairs_one_i<DIM, false>(a, st, lane, cj, jr, ms, i, thr);
    for (uint32_t i = i_mid; i < i_hi; ++i) pairs_one_i<DIM, true>(a, st, lane, cj, jr, ms, i, thr);
    stage_flush(a, st, lane);
}

// ------------------------------------------------------------------------------------------------
// 3. check_motion per candidate pair (from = the newer milestone j, to = the older one i: prm.rs:134)

template <int DIM>
__global__ __launch_bounds__(256) void prm_edge_kernel(DevParams p, PrmArgs a, uint32_t n_cand) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cand) return;
    const uint2 pr = a.cand[c];
    double from[DIM], to[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        from[k] = a.ms[(size_t)pr.x * DIM + k];
        to[k] = a.ms[(size_t)pr.y * DIM + k];
    }
    if (!motion_valid_seq<DIM>(p, from, to)) return;
    const uint32_t slot = atomicAdd(&a.state->n_keys, 2u);
    a.keys[slot] = ((uint64_t)pr.x << 32) | pr.y;       // i in j's list
    a.keys[slot + 1] = ((uint64_t)pr.y << 32) | pr.x;   // j in i's list (prm.rs:143-145)
}

// CSR from the sorted directed keys: offsets by binary search, neighbours = low words
__global__ __launch_bounds__(256) void prm_csr_kernel(const uint64_t* sorted, uint32_t n_keys, uint32_t n_nodes,
                                                       uint32_t* offsets, uint32_t* nbrs) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n_keys) nbrs[t] = (uint32_t)sorted[t];
    if (t <= n_nodes) {
        const uint64_t key = (uint64_t)t << 32;
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

void launch_prm_pairs(const DevParams& p, const PrmArgs& a, uint32_t j0, uint32_t j1, double thr, hipStream_t s) {
    if (j1 <= j0 || j1 < 2) return;
    const uint32_t jblocks = (j1 - j0 + kPairJB - 1) / kPairJB;
    const uint32_t ichunks = (j1 - 1 + kPairIC - 1) / kPairIC;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_pairs_kernel<D>, dim3(ichunks, jblocks), dim3(kPairThreads), 0, s, a, (const double*)a.ms, j0,
                           j1, thr);
    });
}

void launch_prm_edges(const DevParams& p, const PrmArgs& a, uint32_t n_cand, hipStream_t s) {
    if (n_cand == 0) return;
    dim_dispatch(p.dim, [&](auto d) {
        constexpr int D = decltype(d)::value;
        hipLaunchKernelGGL(prm_edge_kernel<D>, dim3((n_cand + 255) / 256), dim3(256), 0, s, p, a, n_cand);
    });
}

hipError_t prm_sort_keys(void* tmp, size_t& tmp_bytes, uint64_t* in, uint64_t* out, uint32_t n_keys, uint32_t n_nodes,
                         hipStream_t s) {
    uint32_t bits = 33;
    while (bits < 64 && ((uint64_t)n_nodes >> (bits - 32)) != 0) ++bits;  // high word < n_nodes
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, (size_t)n_keys, 0u, bits, s);
}

void launch_prm_csr(const uint64_t* sorted, uint32_t n_keys, uint32_t n_nodes, uint32_t* offsets, uint32_t* nbrs,
                    hipStream_t s) {
    const uint32_t work = n_keys > n_nodes + 1 ? n_keys : n_nodes + 1;
    hipLaunchKernelGGL(prm_csr_kernel, dim3((work + 255) / 256), dim3(256), 0, s, sorted, n_keys, n_nodes, offsets, nbrs);
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
