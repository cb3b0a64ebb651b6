// rrt_resident32.hip -- the register-resident RRT grow kernel with a binary32 SCREEN in the scanners.
//
// Same pipeline as rrt_resident.hip (8 scanner waves + 1 resolver wave per problem, LDS query ring, no
// workgroup barrier in steady state) plus a tenth wave that only draws the queries (640 threads).  The main
// difference is what the scanner waves hold and compute:
//   * the tree sits in VGPRs as binary32 roundings of the binary64 nodes (63 VGPRs for 10,240 nodes in R^3
//     instead of 126), and the O(n) scan of rrt.rs:187-196 runs in packed binary32 (v_pk_add/mul/fma_f32: two
//     queries per instruction) with the slot number folded into the low 5 bits of the squared distance, so the
//     per-lane bookkeeping is one v_and_or, one v_med3_u32 (second smallest) and one v_min_u32 per (node, query);
//   * that scan only SCREENS.  Each wave publishes its smallest and second smallest key; the resolver bounds the
//     true binary64 distance of every screened node from its key (error model below) and accepts the scan's
//     winner only when the runner-up is provably farther.  The winner's distance, the steer, the motion check
//     and the comparison with the nodes committed after the scan's snapshot are then computed in binary64 from
//     the binary64 node exactly as in rrt_resident.hip.  When the screen cannot separate winner and runner-up
//     (~1e-4 of queries) the resolver runs the reference's own loop over the binary64 tree in HBM / L2.
// The result is bit-identical to the binary64 kernels; only the work to find it shrinks.
//
// Error model (u = 2^-24; M = largest coordinate magnitude among the bounds, the goal centre and the tree: every
// query is a sample inside the bounds or the goal centre, every new node a convex combination of two of those):
//   e_k = fl32(fl32(q_k) - fl32(c_k))             |e_k - (q_k - c_k)| <= u|q_k| + u|c_k| + u|e_k|  <= 4.1 u M
//   s   = fma(e_2,e_2, fma(e_1,e_1, e_0*e_0))     s = |e|^2 (1 + eta) + zeta, |eta| <= D u, |zeta| <= D 2^-126
//   key = bits(s) with the low 5 bits replaced     value(key) <= s < value(key) (1 + 2^-18)
// so the true distance d of a node with key value v obeys
//   sqrt(v) (1 - R) - A  <=  d  <=  sqrt(v) (1 + R) + A,   A = sqrt(D) 4.1 u M + 1e-18,  R = 2^-19 + (D + 2) u.
// The kernel uses 2A and 2R (+ 2^-21 for taking sqrt(v) in binary32).  A winner is accepted when  sqrt(v2)(1 - 2R) - 2A  >  sqrt(v1)(1 + 2R) + 2A : every
// other screened node is then farther by at least ~1e-7 M, eleven orders of magnitude above the rounding of the
// binary64 post-sqrt compare of rrt.rs:192, so strict-'<' / lowest-index semantics cannot be involved.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"

namespace oxhip {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kKeyInf = 0x7F80001Fu;   // +inf with slot 31: "no node"
constexpr uint32_t kSlotMask = 31u;
#ifndef OXHIP_RING32
#define OXHIP_RING32 32
#endif
constexpr int kRing32 = OXHIP_RING32;        // queries in flight (power of two)
#ifndef OXHIP_S32
#define OXHIP_S32 22
#define OXHIP_C32 14
#endif
constexpr int kPipeThreads32 = kScanThreads + 128;   // + the resolver wave + the sampler wave
#ifndef OXHIP_BATCH32
#define OXHIP_BATCH32 8
#endif
constexpr int kBatch32 = OXHIP_BATCH32;             // queries one scanner pass covers = queries the resolver handles side by side
constexpr int kRowLanes = 64 / kBatch32;            // resolver lanes per query: a DPP row (16) or half-row (8)
constexpr uint32_t kRowMask = (1u << kRowLanes) - 1u;
static_assert(kBatch32 == 4 || kBatch32 == 8, "a query's lane group is a DPP row or half-row");
static_assert(kRowLanes >= kScanWaves, "one lane per scanner wave's result");
#ifndef OXHIP_SCAN_PRIO
#define OXHIP_SCAN_PRIO 16   // a scanner wave whose lead over the resolver is below this many queries runs at raised priority (0: off)
#endif
#ifndef OXHIP_OLDER32
#define OXHIP_OLDER32 0
#endif
constexpr bool kOlder32 = OXHIP_OLDER32 != 0;        // which waves hold the extra rows (Layout)
constexpr int kS32 = OXHIP_S32, kC32 = OXHIP_C32;   // register rows of the heavy / of every scanner wave

struct alignas(16) WavePub32 {   // one wave's screen result for one query (one 16-byte LDS store)
    uint32_t k1;   // smallest key of the wave
    uint32_t k2;   // second smallest key of the wave (with multiplicity)
    uint32_t i1;   // node index of k1
    uint32_t nc;   // tree size this scan covered (the wave's snapshot of `committed`)
};

template <int DIM>
struct alignas(16) QSlot32 {
    double q[DIM];
    uint64_t pos_after;  // stream position after this query's draws
    float qf[4];         // fl32(q): what the scanners screen with (one 16-byte LDS read)
};

template <int DIM>
struct PipeShared32 {
    uint32_t rng_buf[16][64];
    QSlot32<DIM> qring[kRing32];
    WavePub32 pub[kRing32][kScanWaves];
    uint32_t wave_done[kScanWaves];      // queries each scanner wave has published so far (monotonic, one plain store per pass)
    double newn[64][DIM];
    double obs[DIM + 1][64];
    uint32_t sampled, resolved, committed, stop_flag;
    uint32_t mabs_bits;                  // bits of the largest |fl32(coordinate)| the scanners loaded
};

template <int DIM>
__device__ __forceinline__ double node_coord32(const PipeShared32<DIM>& sh, const double* tree, size_t cap, uint32_t n_start,
                                               uint32_t n_now, int k, uint32_t i) {
    if (i >= n_start && i + 64u >= n_now) return sh.newn[i & 63][k];
    return __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t f32_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bits_f32(uint32_t v) { return __builtin_bit_cast(float, v); }

// Wave-wide unsigned minima of four independent values, one v_min_u32_dpp per value and step (the compiler's
// lowering of update_dpp + min costs three VALU instructions per step).  The four chains interleave, so a value's next
// step is three instructions behind its last write and the VALU-write -> DPP-read hazard (2 wait states) needs no
// padding; the leading s_nop covers whatever produced the inputs.  Lanes a step gives no source keep their value.
#define OXHIP_MIN4_STEP(ctrl)                   \
    "v_min_u32_dpp %0, %0, %0 " ctrl "\n"       \
    "v_min_u32_dpp %1, %1, %1 " ctrl "\n"       \
    "v_min_u32_dpp %2, %2, %2 " ctrl "\n"       \
    "v_min_u32_dpp %3, %3, %3 " ctrl "\n"
__device__ __forceinline__ void wave_min4_u32(uint32_t (&v)[4]) {
    uint32_t a = v[0], b = v[1], c = v[2], d = v[3];
    asm("s_nop 1\n"
        OXHIP_MIN4_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        OXHIP_MIN4_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        OXHIP_MIN4_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
        OXHIP_MIN4_STEP("row_mirror row_mask:0xf bank_mask:0xf")
        OXHIP_MIN4_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        OXHIP_MIN4_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    v[0] = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    v[1] = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    v[2] = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
    v[3] = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
}
#undef OXHIP_MIN4_STEP

// per-lane screen state of one query
struct Screen {
    uint32_t b1;   // smallest key
    uint32_t h2;   // second smallest key (with multiplicity)
};
__device__ __forceinline__ void screen_push(Screen& v, float s, uint32_t slot) {
    const uint32_t key = (f32_bits(s) & ~kSlotMask) | slot;
    v.h2 = umed3(key, v.b1, v.h2);
    v.b1 = key < v.b1 ? key : v.b1;
}

// minima over a query's lane group (every lane of the group ends up with it): 16 lanes = a DPP row, 8 = a half-row
__device__ __forceinline__ uint32_t grp_min_u32(uint32_t v) {
    v = dpp_umin_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_umin_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_umin_step<0x141, 0xf>(v);   // row_half_mirror
    if (kRowLanes == 16) v = dpp_umin_step<0x140, 0xf>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ double grp_min_f64(double v) {
    v = dpp_min_step<0xB1, 0xf>(v);
    v = dpp_min_step<0x4E, 0xf>(v);
    v = dpp_min_step<0x141, 0xf>(v);
    if (kRowLanes == 16) v = dpp_min_step<0x140, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t grp_ballot(bool pred, uint32_t row) {
    return (uint32_t)(__ballot(pred) >> (kRowLanes * row)) & kRowMask;
}

// The resolver's view of one query's screen, for the 16-lane DPP row the caller sits in (lanes sub < 8 read the
// eight waves' results).  Row-uniform outputs.
struct RowScreen {
    uint32_t cand;    // the screen's winner (node index; meaningful when clear)
    uint32_t bmin;    // oldest scan snapshot among the eight waves
    bool clear;       // the winner is provably the nearest of all screened nodes
};
struct Margins {
    double a2;        // 2A
    double r_lo;      // 1 - 2R
    double r_hi;      // 1 + 2R
    bool usable;      // M small enough for binary32 squares
};
__device__ __forceinline__ RowScreen row_screen(const WavePub32* pubs, uint32_t row, uint32_t sub, bool active, const Margins& mg) {
    const bool inS = active && sub < (uint32_t)kScanWaves;
    const WavePub32 mine = pubs[inS ? sub : 0];
    const uint32_t k1 = inS ? mine.k1 : kKeyInf;
    const uint32_t k2 = inS ? mine.k2 : kKeyInf;
    const uint32_t nc = inS ? mine.nc : 0xFFFFFFFFu;
    const uint32_t K1 = grp_min_u32(k1);
    const uint32_t eq = grp_ballot(inS && k1 == K1, row);
    const uint32_t wsub = eq ? (uint32_t)(__ffs((int)eq) - 1) : 0u;
    const uint32_t K2 = grp_min_u32((inS && sub == wsub) ? k2 : k1);
    RowScreen r;
    r.cand = (uint32_t)__shfl((int)mine.i1, (int)(kRowLanes * row + wsub), 64);
    r.bmin = grp_min_u32(nc);
    const float v1 = bits_f32(K1 & ~kSlotMask), v2 = bits_f32(K2 & ~kSlotMask);
    // v_sqrt_f32 (1 ulp; a subnormal argument may come back as 0, which the 1e-18 inside A covers): its 2^-22 is in r_lo / r_hi
    const double d1 = (double)__builtin_amdgcn_sqrtf(v1), d2 = (double)__builtin_amdgcn_sqrtf(v2);
    // (+inf second: d2 = +inf and the test holds; +inf or NaN first: it fails)
    r.clear = mg.usable && eq != 0 && (d2 * mg.r_lo - mg.a2 > d1 * mg.r_hi + mg.a2);
    return r;
}

template <int DIM, int S, int C, bool STAMP>
__global__ __launch_bounds__(kPipeThreads32) void rrt_resident32_kernel(DevParams p) {
    constexpr int D = DIM;
    static_assert(S <= 32, "the slot number lives in 5 key bits");

    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;

    __shared__ PipeShared32<DIM> sh;

    const ProblemState st0 = p.state[prob];
    if (p.stop_at_goal && st0.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    uint8_t* skip = p.skip + (size_t)prob * cap;
    const uint32_t budget = (uint32_t)p.budget;  // the host keeps a launch's budget below 2^31

    if (tid < (uint32_t)kScanWaves) sh.wave_done[tid] = 0;
    if (tid == 0) {
        sh.sampled = 0;
        sh.resolved = 0;
        sh.committed = st0.n_nodes;
        sh.stop_flag = 0;
        sh.mabs_bits = 0;
    }
    __syncthreads();

    if (wave < kScanWaves) {
        // ================================================================= scanner waves
        using Lay = Layout<S, C, kOlder32>;
        uint32_t n_local = st0.n_nodes;
        float tr[DIM][S];
        uint32_t mab = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t i = ((uint32_t)s < Lay::kCommon || Lay::heavy(wave)) ? Lay::node_index(wave, lane, (uint32_t)s) : kNoNode;
            const bool in_tree = i < n_local;
            const bool live = in_tree && skip[i] == 0;  // duplicates of a lower-index node never win: hold +inf
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                const float f = in_tree ? (float)tree[(size_t)k * cap + i] : 0.0f;
                const uint32_t ab = f32_bits(f) & 0x7FFFFFFFu;
                mab = ab > mab ? ab : mab;
                tr[k][s] = live ? f : __builtin_inff();
            }
        }
        __hip_atomic_fetch_max(&sh.mabs_bits, mab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();  // the resolver reads mabs_bits after this barrier (the second and last of the launch)

        uint64_t t_wait = 0, t_work = 0, t_pre = 0, t_scan = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;
        uint32_t seen_sampled = 0;
        for (uint32_t j = 0; j < budget; j += kBatch32) {
            const uint32_t nb = (budget - j < (uint32_t)kBatch32) ? (budget - j) : (uint32_t)kBatch32;
            const uint32_t need = j + nb;
            // wait until the pass's queries are sampled (implies their ring slots were consumed kRing32 queries ago)
            for (uint32_t spins = 0; seen_sampled < need; ++spins) {
                if (lds_peek(&sh.stop_flag) != 0 || spins > kMaxSpins) break;  // every spin is bounded
                seen_sampled = uni(lds_peek(&sh.sampled));
                if (seen_sampled < need) __builtin_amdgcn_s_sleep(2);
            }
            if (seen_sampled < need) break;  // stop requested
#if OXHIP_SCAN_PRIO
            // two scanner waves share a SIMD and the older one wins every issue conflict: it races ahead, then idles at the
            // ring, while the younger one -- the wave the resolver ends up waiting for -- crawls.  The wave with the
            // smaller lead over the resolver takes the higher priority for this pass.
            if (j - uni(lds_peek(&sh.resolved)) < (uint32_t)OXHIP_SCAN_PRIO) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }
            // absorb the nodes committed since this wave's last snapshot (the owner lane takes each)
            const uint32_t nc = uni(lds_peek(&sh.committed));
            for (uint32_t i = n_local; i < nc; ++i) {
                uint32_t owner_thread, sl;
                Lay::locate(i, owner_thread, sl);
                if ((owner_thread >> 6) != wave) continue;   // another wave's node: skip the slot ladder below
                const bool mine = tid == owner_thread;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (sl == (uint32_t)s) {
                        if (mine) {
#pragma unroll
                            for (int k = 0; k < D; ++k) tr[k][s] = (float)sh.newn[i & 63][k];   // +inf stays +inf
                        }
                    }
                }
            }
            n_local = nc;
            // the pass's queries, two per packed register
            f32x2 q[kBatch32 / 2][D];
#pragma unroll
            for (int b = 0; b < kBatch32; ++b) {
                const uint32_t slot = (j + ((uint32_t)b < nb ? (uint32_t)b : 0u)) & (kRing32 - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) q[b / 2][k][b % 2] = bits_f32(uni(f32_bits(sh.qring[slot].qf[k])));   // wave-uniform: scalar registers
            }
            const uint32_t nslots = Lay::slots_in_use(wave, nc);
            Screen sc[kBatch32];
#pragma unroll
            for (int b = 0; b < kBatch32; ++b) sc[b] = Screen{kKeyInf, kKeyInf};
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_pre += now - t_mark; t_mark = now; }
            // screen (rrt.rs:187-196 in binary32): groups of kGroup slots under one uniform branch
#pragma unroll
            for (int g0 = 0; g0 < S; g0 += group_len<S, C>(g0)) {
                if ((uint32_t)g0 < nslots) {
#pragma unroll
                    for (int s = g0; s < g0 + group_len<S, C>(g0); ++s) {
#pragma unroll
                        for (int bp = 0; bp < kBatch32 / 2; ++bp) {
                            f32x2 e = (f32x2){tr[0][s], tr[0][s]} - q[bp][0];
                            f32x2 acc = e * e;
#pragma unroll
                            for (int k = 1; k < D; ++k) {
                                e = (f32x2){tr[k][s], tr[k][s]} - q[bp][k];
                                acc = __builtin_elementwise_fma(e, e, acc);
                            }
                            screen_push(sc[2 * bp], acc[0], (uint32_t)s);
                            screen_push(sc[2 * bp + 1], acc[1], (uint32_t)s);
                        }
                    }
                }
            }
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_scan += now - t_mark; t_mark = now; }
            // reduce: the wave's smallest key, its lane, and the smallest of everything else
            uint32_t k1w[kBatch32], k2w[kBatch32];
            int wl[kBatch32];
#pragma unroll
            for (int b0 = 0; b0 < kBatch32; b0 += 4) {
                uint32_t t4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) t4[t] = sc[b0 + t].b1;
                wave_min4_u32(t4);
#pragma unroll
                for (int t = 0; t < 4; ++t) k1w[b0 + t] = t4[t];
            }
#pragma unroll
            for (int b = 0; b < kBatch32; ++b) {
                const uint64_t eqm = __ballot(sc[b].b1 == k1w[b]);
                wl[b] = __ffsll((unsigned long long)eqm) - 1;   // eqm != 0: the minimum is attained
            }
#pragma unroll
            for (int b0 = 0; b0 < kBatch32; b0 += 4) {
                uint32_t t4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) t4[t] = (int)lane == wl[b0 + t] ? sc[b0 + t].h2 : sc[b0 + t].b1;
                wave_min4_u32(t4);
#pragma unroll
                for (int t = 0; t < 4; ++t) k2w[b0 + t] = t4[t];
            }
#pragma unroll
            for (int b = 0; b < kBatch32; ++b) {
                if ((uint32_t)b < nb) {
                    const uint32_t slot = (j + (uint32_t)b) & (kRing32 - 1);
                    if (lane == 0) {
                        WavePub32 out;
                        out.k1 = k1w[b];
                        out.k2 = k2w[b];
                        out.i1 = Lay::node_index(wave, (uint32_t)wl[b], k1w[b] & kSlotMask);
                        out.nc = nc;
                        sh.pub[slot][wave] = out;
                    }
                }
            }
            if (lane == 0) lds_post(&sh.wave_done[wave], need);   // after the records (LDS is in order within a wave)
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) {
            p.dbg[16 + wave] = t_wait;
            p.dbg[24 + wave] = t_work + t_pre + t_scan;
            if (wave == 5) { p.dbg[8] = t_pre; p.dbg[9] = t_scan; p.dbg[10] = t_work; }
        }
        return;
    }

    if (wave == (uint32_t)kScanWaves + 1u) {
        // ================================================================= sampler wave
        // rrt.rs:177-184 for the whole launch, ahead of everybody: the queries depend on the RNG stream only, never on
        // the tree, so they are drawn by a wave of their own and the resolver's critical path starts at the scan results.
        // A ring slot is reused only after its previous tenant was resolved (`resolved`, posted by the resolver).
        double goal_c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
        __syncthreads();  // the launch's second barrier (see the scanners)
        RngWindow rng;
        rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st0.draws);
        uint64_t t_busy = 0;
        uint32_t js = 0;
        while (js < budget) {
            uint32_t jr_seen = 0;
            bool go = false;
            for (uint32_t spins = 0; spins <= kMaxSpins; ++spins) {   // every spin is bounded
                if (lds_peek(&sh.stop_flag) != 0) break;
                jr_seen = uni(lds_peek(&sh.resolved));
                if (js - jr_seen <= (uint32_t)(kRing32 / 2)) { go = true; break; }   // half the ring is free: refill it
                __builtin_amdgcn_s_sleep(2);
            }
            if (!go) break;  // stop requested (or a protocol bug: the resolver's own guard reports it)
            const uint64_t t0 = STAMP ? (uint64_t)clock64() : 0;
            uint32_t m = jr_seen + kRing32 - js;  // free ring slots
            if (m > budget - js) m = budget - js;
            // keep the batch's words inside the LDS window: refill (64 blocks from the current position) when short
            const uint64_t need_hi = rng.pos + (uint64_t)m * (1 + D) + 64;
            if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
                rng.base_blk = uni64(rng.pos >> 3);
                uint32_t o[16];
                chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
                for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
            }
            if (!sample_batch<DIM, kRing32, QSlot32<DIM>>(rng, p, goal_c, m, lane, sh.qring, js)) {
                for (uint32_t b = 0; b < m; ++b) {  // (never expected) a redraw ran past the window: one by one
                    double qn[D];
                    sample_state<D, false>(rng, p, DIM, goal_c, qn);
                    QSlot32<DIM>& qs = sh.qring[(js + b) & (kRing32 - 1)];
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) qs.q[k] = qn[k];
                        qs.pos_after = rng.pos;
                    }
                }
            }
            if (lane < m) {   // the binary32 copies the scanners read (LDS is in order within this wave)
                QSlot32<DIM>& qs = sh.qring[(js + lane) & (kRing32 - 1)];
#pragma unroll
                for (int k = 0; k < D; ++k) qs.qf[k] = (float)qs.q[k];
            }
            js += m;
            if (lane == 0) lds_post(&sh.sampled, js);
            if (STAMP) t_busy += (uint64_t)clock64() - t0;
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) p.dbg[0] = t_busy;
        return;
    }

    // ===================================================================== resolver wave
    __builtin_amdgcn_s_setprio(3);  // the youngest wave of its SIMD would otherwise queue behind two scanners
    ProblemState st = st0;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    const double goal_thr = p.goal_thr[prob];
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
    const bool extras = nobs > ns64;  // spheres beyond the first 64 and every box: always stepped, never filtered
    double oc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) oc[k] = lane < ns64 ? p.sph_c[(size_t)k * p.n_spheres + lane] : 0.0;
    const double othr = lane < ns64 ? p.sph_thr[lane] : -1.0;
    const double ofilt = lane < ns64 ? p.sph_filt[lane] : -1.0;
#pragma unroll
    for (int k = 0; k < D; ++k) sh.obs[k][lane] = oc[k];
    sh.obs[D][lane] = ofilt;

    uint64_t draws_done = st.draws;
    uint32_t n = st.n_nodes;
    uint32_t jr = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    const uint32_t row = lane / (uint32_t)kRowLanes, sub = lane % (uint32_t)kRowLanes;
    uint64_t t_wait = 0, t_work = 0, t_comb = 0, n_amb = 0, n_seq = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;

    __syncthreads();  // pairs with the scanners' second barrier: mabs_bits is final
    Margins mg;
    {
        // M: the tree as loaded (binary32 roundings, hence the 1 + 2^-23), the bounds and the goal centre
        double m = (double)bits_f32(lds_peek(&sh.mabs_bits)) * (1.0 + 0x1p-23);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            m = fmax(m, fmax(fabs(p.lo[k]), fabs(p.hi[k])));
            m = fmax(m, fabs(goal_c[k]));
        }
        m = unid(m) * 1.001;   // interpolation rounding over any chain of inserts
        const double u = 0x1p-24;
        mg.usable = m < 1e15;  // also false for NaN / inf
        mg.a2 = 2.0 * (sqrt((double)D) * 4.1 * u * m + 1e-18);
        const double r2 = 2.0 * (0x1p-19 + (double)(D + 2) * u) + 0x1p-21;   // + the binary32 square root of row_screen
        mg.r_lo = 1.0 - r2;
        mg.r_hi = 1.0 + r2;
    }

    // ---- one query, the reference's sequential semantics in full (tails, batch conflicts, near-ties):
    //      candidates = the screen's winner + every node committed after the oldest scan snapshot
    auto resolve_one = [&](uint32_t jq, uint32_t& nearest, double (&q_new)[D], bool& dup) -> bool {
        const uint32_t slot = jq & (kRing32 - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = unid(sh.qring[slot].q[k]);
        // every row evaluates the same query's screen: the outputs are wave-uniform
        const RowScreen rs = row_screen(sh.pub[slot], row, sub, true, mg);
        const bool clear = __builtin_amdgcn_readfirstlane(rs.clear ? 1 : 0) != 0;
        const uint32_t cand = uni(rs.cand);
        const uint32_t base_min = uni(rs.bmin);
        double cc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) cc[k] = clear ? unid(node_coord32<DIM>(sh, tree, cap, st0.n_nodes, n, k, cand < n ? cand : 0u)) : 0.0;
        const bool inS = lane == 0;   // lane 0 carries the screen's winner with its binary64 d2
        const double pb = (inS && clear) ? dist2<D>(cc, q, DIM) : __builtin_inf();
        const uint32_t pidxS = inS ? cand : kNoNode;
        // the ring entry of this lane holds the latest node i with (i & 63) == lane
        uint32_t pidx = kNoNode;
        if (n > lane) {
            const uint32_t i = lane + (((n - 1u - lane) >> 6) << 6);
            if (i >= base_min) pidx = i;
        }
        double pn[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pn[k] = sh.newn[lane][k];
        const bool pv = pidx != kNoNode;
        const double d2p = pv ? dist2<D>(pn, q, DIM) : __builtin_inf();
        const double g = wave_min_f64(d2p < pb ? d2p : pb);
        const uint32_t hb = hi32(g) + 1;
        const bool nearS = inS && clear && hi32(pb) <= hb;
        const bool nearP = pv && hi32(d2p) <= hb;
        const uint64_t mS = __ballot(nearS), mP = __ballot(nearP);
        const bool from_scan = mS != 0;
        const int wl = from_scan ? 0 : (mP ? (__ffsll((unsigned long long)mP) - 1) : 0);
        nearest = from_scan ? cand : (uint32_t)__builtin_amdgcn_readlane((int)pidx, wl);
        // unambiguous iff the screen was clear and every near candidate is that one node
        const bool amb = !clear || (mS == 0 && mP == 0) || __ballot((nearS && pidxS != nearest) || (nearP && pidx != nearest)) != 0;
        double q_near[D];
        double dist_or_g;
        if (!amb) {
#pragma unroll
            for (int k = 0; k < D; ++k) q_near[k] = from_scan ? cc[k] : unid(sh.newn[wl][k]);
            dist_or_g = g;
            dup = g == 0.0;
        } else {
            // the reference's own loop -- post-sqrt compare with lowest-index ties -- over the persistent
            // binary64 copy of the tree in global memory
            if (STAMP) ++n_amb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            Exact e{__builtin_inf(), kNoNode};
            for (uint32_t i = lane; i < n; i += 64) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    c[k] = __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double d = sqrt(dist2<D>(c, q, DIM));
                if (d < e.dist) { e.dist = d; e.idx = i; }
            }
            e = exact_wave_reduce(e);
            nearest = uni(e.idx);
#pragma unroll
            for (int k = 0; k < D; ++k)
                q_near[k] = unid(__hip_atomic_load(&tree[(size_t)k * cap + nearest], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            dist_or_g = unid(e.dist);
            dup = dist_or_g == 0.0;
        }
        steer<DIM>(p, amb, dist_or_g, q_near, q, q_new);
        if (nobs == 0) return true;
        double mid[D];
        lerp<DIM>(q_near, q_new, 0.5, mid, DIM);
        if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid)) == 0 && !extras) return true;
        return motion_lanes<DIM>(p, lane, q_near, q_new, oc, othr, ofilt, ns64);
    };

    while (jr < budget) {
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }

        // the scanners publish kBatch32 queries per pass: resolve them as one batch
        const uint32_t nbq = (budget - jr < (uint32_t)kBatch32) ? (budget - jr) : (uint32_t)kBatch32;
        uint32_t spins = 0;
        while (__ballot(lane < (uint32_t)kScanWaves && lds_peek(&sh.wave_done[lane & (kScanWaves - 1)]) < jr + nbq) != 0 &&
               spins <= kMaxSpins) {   // some scanner wave has not published this batch yet
            __builtin_amdgcn_s_sleep(1);
            ++spins;
        }
        if (spins > kMaxSpins) { stop = 4; break; }  // OXHIP_STOP_INTERNAL: a scanner never published (bug guard)
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }

        // ---- row-parallel phase: lane group r (a DPP row or half-row) works on query jr + r against the tree of n0 nodes
        const uint32_t n0 = n;
        const bool active = row < nbq;
        const uint32_t slot_r = (jr + (active ? row : 0u)) & (kRing32 - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = sh.qring[slot_r].q[k];
        const uint64_t pos_after_r = sh.qring[slot_r].pos_after;
        const RowScreen rs = row_screen(sh.pub[slot_r], row, sub, active, mg);
        const uint32_t bmin_r = rs.bmin;
        if (__ballot(active && (n0 - bmin_r > 64u || bmin_r > n0)) != 0) { stop = 4; break; }  // ring would have wrapped (bug guard)
        const bool clear_r = active && rs.clear;
        // the screen's winner with its binary64 coordinates and d2 (lane sub == 0 of the row carries it)
        double cc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) cc[k] = node_coord32<DIM>(sh, tree, cap, st0.n_nodes, n0, k, (clear_r && rs.cand < n0) ? rs.cand : 0u);
        const bool inS = clear_r && sub == 0;
        const double pb = inS ? dist2<D>(cc, q, DIM) : __builtin_inf();
        const uint32_t pidxS = rs.cand;
        // nodes committed after the oldest snapshot (at most a few): lanes of the row stride over them
        Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
        if (active) {
            for (uint32_t i = bmin_r + sub; i < n0; i += (uint32_t)kRowLanes) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.newn[i & 63][k];
                scan_push(pd, dist2<D>(c, q, DIM), i);  // ascending i: ties keep the lower index
            }
        }
        const double g_r = grp_min_f64(pd.b1 < pb ? pd.b1 : pb);
        const uint32_t hb = hi32(g_r) + 1;
        const bool nearS = inS && hi32(pb) <= hb;
        const bool nearP = active && pd.slot != kNoNode && hi32(pd.b1) <= hb;
        const uint32_t rowS = grp_ballot(nearS, row);
        const uint32_t rowP = grp_ballot(nearP, row);
        const bool from_scan = rowS != 0;
        const uint32_t wsub = from_scan ? 0u : (rowP ? (uint32_t)(__ffs((int)rowP) - 1) : 0u);
        const int src_lane = (int)(kRowLanes * row + wsub);
        const uint32_t wP = (uint32_t)__shfl((int)pd.slot, src_lane, 64);
        const uint32_t nearest_r = from_scan ? pidxS : wP;
        // ambiguous iff the screen could not name a winner, or a second node is near the binary64 minimum
        const bool amb_l = (nearP && pd.slot != nearest_r) || (active && pd.h2 <= hb);
        const bool amb_r = !clear_r || grp_ballot(amb_l, row) != 0 || (rowS == 0 && rowP == 0);
        double q_near[D], qn[D], mid[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q_near[k] = from_scan ? cc[k] : sh.newn[nearest_r & 63][k];
        steer<DIM>(p, false, g_r, q_near, q, qn);
        lerp<DIM>(q_near, qn, 0.5, mid, DIM);
        bool maybe_l = false;
        if (nobs > 0) {
            for (uint32_t o = sub; o < 64; o += (uint32_t)kRowLanes) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                maybe_l = maybe_l || sphere_maybe_hit<DIM>(c, sh.obs[D][o], mid);
            }
        }
        const bool maybe_r = extras || grp_ballot(maybe_l, row) != 0;
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_comb += now - t_mark; t_mark = now; }

        // ---- batch commit: when no query of a full batch is ambiguous, no node accepted earlier in the batch comes
        //      near a later query, the node cap is not reached inside the batch and no goal hit has to stop the launch,
        //      the four iterations are independent given the row results, and the per-query bookkeeping below (the
        //      "sequential phase", ~900 cycles per query of scalar code) collapses into one pass.  Anything else falls
        //      through to the sequential phase, which is the reference's order literally.
        if (nbq == (uint32_t)kBatch32 && __ballot(amb_r) == 0) {
            // motion checks of the rows the midpoint filter could not clear (is_valid is pure: the order is free)
            uint32_t okmask = (1u << kBatch32) - 1u;
            if (nobs > 0) {
#pragma unroll
                for (int r = 0; r < kBatch32; ++r) {
                    if (__builtin_amdgcn_readlane(maybe_r ? 1 : 0, kRowLanes * r) != 0) {
                        double a[D], bq[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) { a[k] = readlane_f64(q_near[k], kRowLanes * r); bq[k] = readlane_f64(qn[k], kRowLanes * r); }
                        if (!motion_lanes<DIM>(p, lane, a, bq, oc, othr, ofilt, ns64)) okmask &= ~(1u << r);
                    }
                }
            }
            const bool ins = !p.freeze;
            const uint32_t cnt = ins ? (uint32_t)__popc(okmask) : 0u;
            const bool ok_l = ((okmask >> row) & 1u) != 0;
            const bool dup_l = g_r == 0.0;
            bool special = ins && n + cnt >= p.max_nodes;   // a later query of the batch might have to stop at the cap
            if (ins && !special) {
                // would a node accepted earlier in the batch be (nearly) as close to this row's query as its nearest?
                bool conflict_l = false;
#pragma unroll
                for (int a = 0; a < kBatch32 - 1; ++a) {
                    double ca[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) ca[k] = readlane_f64(qn[k], kRowLanes * a);
                    const bool a_new = ((okmask >> a) & 1u) != 0 && readlane_f64(g_r, kRowLanes * a) != 0.0;   // accepted and not a duplicate
                    conflict_l = conflict_l || (a_new && row > (uint32_t)a && hi32(dist2<D>(ca, q, DIM)) <= hb);
                }
                const bool hit_l = ok_l && dist2<D>(qn, goal_c, DIM) <= goal_thr;
                const uint64_t hits = __ballot(hit_l && sub == 0);
                special = __ballot(conflict_l) != 0 || (hits != 0 && p.stop_at_goal);
                if (!special) {
                    const uint32_t idx = n + (uint32_t)__popc(okmask & ((1u << row) - 1u));
                    if (ok_l && sub == 0) {
                        // 6. insert (rrt.rs:213-217), as in the sequential phase
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            sh.newn[idx & 63][k] = dup_l ? __builtin_inf() : qn[k];
                            tree[(size_t)k * cap + idx] = qn[k];
                        }
                        parent[idx] = (int32_t)nearest_r;
                        skip[idx] = dup_l ? 1 : 0;
                    }
                    if (hits != 0 && st.goal_node < 0)   // 7. goal test (rrt.rs:220-223): the first hit in query order
                        st.goal_node = (int32_t)__builtin_amdgcn_readlane((int)idx, __ffsll((unsigned long long)hits) - 1);
                    n += cnt;
                    if (lane == 0) lds_post(&sh.committed, n);
                }
            }
            if (!special) {
                // every lane group digests its own iteration (in parallel); the chain over the batch is one multiply-add each
                const uint64_t g_l = iter_digest<D>(nearest_r, qn, DIM, ok_l);
                uint64_t h = st.checksum;
#pragma unroll
                for (int r = 0; r < kBatch32; ++r)
                    h = chk_push(h, ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(g_l >> 32), kRowLanes * r) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)g_l, kRowLanes * r));
                st.checksum = h;
                st.iterations += kBatch32;
                st.accepted += (uint64_t)__popc(okmask);
                draws_done = uni64((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_r, kRowLanes * (kBatch32 - 1)) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_r >> 32), kRowLanes * (kBatch32 - 1)) << 32));
                if (lane == 0) lds_post(&sh.resolved, jr + (uint32_t)kBatch32);   // the sampler may hand the slots out again
                jr += (uint32_t)kBatch32;
                if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
                continue;
            }
        }

        // ---- sequential phase: commit in query order; a node committed earlier in this batch that is
        //      closer (or near-tied) to a later query forces that query through resolve_one
        if (STAMP) ++n_seq;
        double cn[kBatch32][D];   // coordinates the scanners will hold for the nodes committed in this batch
        bool cn_valid[kBatch32];
#pragma unroll
        for (int b = 0; b < kBatch32; ++b) cn_valid[b] = false;
        bool leave = false;
        uint32_t processed = 0;
#pragma unroll
        for (int b = 0; b < kBatch32; ++b) {
            if (!leave && (uint32_t)b < nbq) {
                if (!p.freeze && n >= p.max_nodes) { stop = 2; leave = true; }
            }
            if (!leave && (uint32_t)b < nbq) {
                const int l0 = kRowLanes * b;
                const double g_b = readlane_f64(g_r, l0);
                double q_b[D];
#pragma unroll
                for (int k = 0; k < D; ++k) q_b[k] = readlane_f64(q[k], l0);
                bool redo = __builtin_amdgcn_readlane(amb_r ? 1 : 0, l0) != 0;
#pragma unroll
                for (int c = 0; c < b; ++c)
                    if (cn_valid[c] && hi32(dist2<D>(cn[c], q_b, DIM)) <= hi32(g_b) + 1) redo = true;
                uint32_t nearest;
                double q_new[D];
                bool ok, dup;
                if (redo) {
                    ok = resolve_one(jr + (uint32_t)b, nearest, q_new, dup);
                } else {
                    nearest = (uint32_t)__builtin_amdgcn_readlane((int)nearest_r, l0);
#pragma unroll
                    for (int k = 0; k < D; ++k) q_new[k] = readlane_f64(qn[k], l0);
                    dup = g_b == 0.0;
                    ok = true;
                    if (nobs > 0 && __builtin_amdgcn_readlane(maybe_r ? 1 : 0, l0) != 0) {
                        double qnr[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) qnr[k] = readlane_f64(q_near[k], l0);
                        ok = motion_lanes<DIM>(p, lane, qnr, q_new, oc, othr, ofilt, ns64);
                    }
                }
                st.checksum = uni64(chk_push(st.checksum, iter_digest<D>(nearest, q_new, DIM, ok)));
                st.iterations++;
                draws_done = uni64((uint64_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_r, l0) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_r >> 32), l0) << 32));
                bool hit = false;
                if (ok) {
                    st.accepted++;
                    if (!p.freeze) {
                        // 6. insert (rrt.rs:213-217): LDS hand-off to the owning scanner lane + HBM copy.  A node at
                        // distance 0 from its nearest node repeats that node's coordinates, and the strict '<' of
                        // rrt.rs:192 can never prefer it over the lower index: the scanners keep +inf for it.
                        const uint32_t i = n;
#pragma unroll
                        for (int k = 0; k < D; ++k) cn[b][k] = dup ? __builtin_inf() : q_new[k];
                        cn_valid[b] = true;
                        if (lane == (i & 63)) {
#pragma unroll
                            for (int k = 0; k < D; ++k) {
                                sh.newn[i & 63][k] = cn[b][k];
                                tree[(size_t)k * cap + i] = q_new[k];
                            }
                            parent[i] = (int32_t)nearest;
                            skip[i] = dup ? 1 : 0;
                        }
                        ++n;
                        if (lane == 0) lds_post(&sh.committed, n);
                        // 7. goal test (rrt.rs:220-223)
                        if (dist2<D>(q_new, goal_c, DIM) <= goal_thr) {
                            if (st.goal_node < 0) st.goal_node = (int32_t)i;
                            hit = true;
                        }
                    }
                }
                if (lane == 0) lds_post(&sh.resolved, jr + (uint32_t)b + 1);   // the sampler may hand the slot out again
                ++processed;
                if (hit && p.stop_at_goal) { stop = 0; leave = true; }
            }
        }
        jr += processed;
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        if (leave) break;
    }
    if (lane == 0) {
        lds_post(&sh.stop_flag, 1);
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
        if (STAMP && p.dbg && prob == 0) {
            p.dbg[5] = n_seq; p.dbg[1] = t_wait; p.dbg[2] = t_work; p.dbg[3] = t_comb; p.dbg[4] = n_amb; p.dbg[7] = st.iterations;
        }
    }
}

// instantiations: (dim, slots) -> capacity as in rrt_resident.hip
static int pick_slots32(uint32_t cap) {
    const uint32_t need = (cap + kScanThreads - 1) / kScanThreads;
    if (need <= 4) return 4;
    if (cap <= Layout<kS32, kC32, kOlder32>::kCapacity) return kS32;
    return 0;
}

bool resident32_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pick_slots32(cap) != 0; }

void launch_rrt_resident32(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kPipeThreads32);
    const int s = pick_slots32(p.cap);
#define OXHIP_LAUNCH(DIM_, S_, C_)                                                                               \
    do {                                                                                                      \
        if (p.dbg) hipLaunchKernelGGL((rrt_resident32_kernel<DIM_, S_, C_, true>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((rrt_resident32_kernel<DIM_, S_, C_, false>), grid, block, 0, stream, p);         \
    } while (0)
    if (p.dim == 3) {
        if (s == 4) OXHIP_LAUNCH(3, 4, 4); else OXHIP_LAUNCH(3, kS32, kC32);
    } else {
        if (s == 4) OXHIP_LAUNCH(2, 4, 4); else OXHIP_LAUNCH(2, kS32, kC32);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
