// rrt_lanes.hip -- the register-resident RRT grow kernel with a LANE-PER-QUERY resolver.
//
// Same idea as rrt_resident32.hip -- the tree of one planning problem lives in the vector registers of eight scanner
// waves as binary32 roundings, the O(n) scan of rrt.rs:187-196 is a packed-binary32 SCREEN, and everything that enters
// a result is computed in binary64 from the binary64 nodes -- but the wave that turns screen results into iterations is
// organised the other way round.  rrt_resident32's resolver gives every query a group of 8 lanes and handles 8 queries per
// pass: steer (a binary64 sqrt and a division), the midpoint filter, the checksum chain are then paid per 8 queries, and that
// wave, not the scan, bounds the kernel (about 1,100 cycles per iteration).  Here lane j of the resolver IS
// query jr + j: up to 64 iterations are resolved side by side, so one sqrt / division / filter pass serves 64 queries.
//
// Sequential semantics (iteration k sees exactly the tree left by iterations < k, rrt.rs:170-225) are kept by committing,
// per round, the longest prefix of the 64 lanes whose results cannot have been changed by the nodes the lanes before them
// insert (a new node changes a later query's result only if it is at least as close as that query's nearest node: one d2
// per pair, lane-parallel), and re-resolving the rest in the next round against the grown tree.  With inserts suppressed
// (the steady measurement) every round commits all its lanes; while a tree is small a round commits ~sqrt(2n) lanes.
// A lane whose winner the screen cannot prove, or whose binary64 candidates are within a rounding of each other, ends the
// prefix and is resolved alone by the reference's literal loop over the binary64 tree (post-sqrt compare, lowest index).
//
// Waves: 8 scanners (64 x 20 register rows = 10,240 nodes), 1 resolver, 1 sampler (draws the queries ahead: they depend on
// the RNG stream only).  LDS rings: 128 queries in flight, the last 256 committed nodes.  No workgroup barrier in steady state.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"

namespace oxhip {

typedef float lf32x2 __attribute__((ext_vector_type(2)));

constexpr int kLanesThreads = kScanThreads + 128;   // + the resolver wave + the sampler wave
constexpr int kQRing = 128;                         // queries in flight (power of two)
constexpr int kNRing = 256;                         // committed nodes kept in LDS (power of two, >= kQRing + 64)
constexpr int kPassQ = 8;                           // queries one scanner pass covers
constexpr uint32_t kDepthGrow = 64;                 // queries sampled ahead of the resolver while inserts are on
constexpr uint32_t kLKeyInf = 0x7F80001Fu;          // +inf with slot 31: "no node"
constexpr uint32_t kLSlotMask = 31u;
#ifndef OXHIP_LANES_PRIO
#define OXHIP_LANES_PRIO 16
#endif

struct alignas(16) LanePub {   // one wave's screen result for one query (one 16-byte LDS record)
    uint32_t k1;   // smallest key of the wave
    uint32_t k2;   // second smallest key of the wave (with multiplicity)
    uint32_t i1;   // node index of k1
    uint32_t nc;   // tree size this scan covered (the wave's snapshot of `committed`)
};

template <int DIM>
struct LanesShared {
    uint32_t rng_buf[16][64];
    double q[DIM][kQRing];               // the queries, coordinate-major: resolver lane j reads q[k][slot_j] conflict-free
    uint64_t pos_after[kQRing];          // stream position after each query's draws
    float qf[kQRing][4];                 // fl32(q): what the scanners screen with (one 16-byte uniform read per query)
    LanePub pub[kScanWaves][kQRing];     // wave-major: resolver lane j reads pub[w][slot_j] conflict-free
    double newn[DIM][kNRing];            // the last kNRing committed nodes, node i at i & (kNRing - 1); +inf for skipped duplicates
    double obs[DIM + 2][64];             // first 64 spheres: centre, validity threshold, filter threshold
    uint32_t wave_done[kScanWaves];      // queries each scanner wave has published (monotonic)
    uint32_t sampled, resolved, committed, stop_flag;
    uint32_t heartbeat;                  // bumped by the resolver while it works: waiters only give up when it stands still
    uint32_t mabs_bits;                  // bits of the largest |fl32(coordinate)| the scanners loaded
};

__device__ __forceinline__ uint32_t lf32_bits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float lbits_f32(uint32_t v) { return __builtin_bit_cast(float, v); }

// wave-wide unsigned minima of four independent values (see rrt_resident32.hip: one v_min_u32_dpp per value and step)
#define OXHIP_LMIN4_STEP(ctrl)                  \
    "v_min_u32_dpp %0, %0, %0 " ctrl "\n"       \
    "v_min_u32_dpp %1, %1, %1 " ctrl "\n"       \
    "v_min_u32_dpp %2, %2, %2 " ctrl "\n"       \
    "v_min_u32_dpp %3, %3, %3 " ctrl "\n"
__device__ __forceinline__ void lanes_min4_u32(uint32_t (&v)[4]) {
    uint32_t a = v[0], b = v[1], c = v[2], d = v[3];
    asm("s_nop 1\n"
        OXHIP_LMIN4_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        OXHIP_LMIN4_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        OXHIP_LMIN4_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
        OXHIP_LMIN4_STEP("row_mirror row_mask:0xf bank_mask:0xf")
        OXHIP_LMIN4_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        OXHIP_LMIN4_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    v[0] = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
    v[1] = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    v[2] = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
    v[3] = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
}
#undef OXHIP_LMIN4_STEP

struct LScreen {
    uint32_t b1;   // smallest key
    uint32_t h2;   // second smallest key (with multiplicity)
};
__device__ __forceinline__ void lscreen_push(LScreen& v, float s, uint32_t slot) {
    const uint32_t key = (lf32_bits(s) & ~kLSlotMask) | slot;
    v.h2 = umed3(key, v.b1, v.h2);
    v.b1 = key < v.b1 ? key : v.b1;
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

__device__ __forceinline__ uint64_t below_mask(uint32_t lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint64_t first_n_mask(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

// Lane-parallel sampling of m <= 64 consecutive queries into the coordinate-major ring: sample_batch of
// rrt_resident_common.hpp (rrt.rs:177-184 + rvss.rs:233-249) with this kernel's ring layout and the fl32 copies.
template <int DIM>
__device__ __forceinline__ bool sample_lanes(RngWindow& rng, const DevParams& p, const double* goal_c, uint32_t m,
                                             uint32_t lane, LanesShared<DIM>& sh, uint32_t js) {
    const uint64_t win_lo = rng.base_blk * 8;
    const uint64_t pos0 = rng.pos;
    if (pos0 < win_lo || pos0 + (uint64_t)m * (1 + DIM) > win_lo + 512) return false;
    const uint32_t rel0 = (uint32_t)(pos0 - win_lo);
    const bool act = lane < m;
    const bool always_goal = p.p_int == ~0ull;
    auto word = [&](uint32_t rel) -> uint64_t {
        const uint32_t a = rel0 + rel, bl = a >> 3, w = (a & 7u) * 2u;
        return ((uint64_t)rng.buf[w + 1][bl] << 32) | rng.buf[w][bl];
    };
    uint64_t goal_mask = always_goal ? ~0ull : 0ull;
    uint32_t off = 0;
    if (!always_goal) {
        const uint64_t below = below_mask(lane);
        for (uint32_t round = 0; round <= m; ++round) {
            off = act ? (1u + DIM) * lane - (uint32_t)DIM * (uint32_t)__popcll(goal_mask & below) : 0u;
            const uint64_t now = __ballot(act && word(off) < p.p_int);
            if (now == goal_mask) break;
            goal_mask = now;
        }
    }
    const bool goal = (goal_mask >> lane) & 1ull;
    double q[DIM];
    bool redraw = false;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const uint64_t bits = (word(act && !goal ? off + 1u + (uint32_t)k : 0u) >> 12) | 0x3FF0000000000000ull;
        const double v01 = __longlong_as_double((long long)bits) - 1.0;
        double res = v01 * p.scale[k];
        res = res + p.lo[k];
        redraw = redraw || !(res < p.hi[k]);
        q[k] = goal ? goal_c[k] : res;
    }
    if (__ballot(act && !goal && redraw) != 0) return false;
    const uint32_t cnt = always_goal ? 0u : (goal ? 1u : 1u + (uint32_t)DIM);
    if (act) {
        const uint32_t slot = (js + lane) & (kQRing - 1);
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            sh.q[k][slot] = q[k];
            sh.qf[slot][k] = (float)q[k];
        }
        sh.pos_after[slot] = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}

struct LMargins {
    double a2;        // 2A
    double r_lo;      // 1 - 2R
    double r_hi;      // 1 + 2R
    bool usable;      // M small enough for binary32 squares
};

template <int DIM, int S, int C, bool STAMP>
__global__ __launch_bounds__(kLanesThreads) void rrt_lanes_kernel(DevParams p) {
    constexpr int D = DIM;
    static_assert(S <= 32, "the slot number lives in 5 key bits");
    static_assert(DIM <= 4, "qf holds four floats per query");
    // Scanner waves 0 and 4 share their SIMD with the resolver: they hold C rows, the other six S (rrt_resident_common.hpp)
    using Lay = Layout<S, C, false>;

    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;

    __shared__ LanesShared<DIM> sh;

    const ProblemState st0 = p.state[prob];
    if (p.stop_at_goal && st0.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    uint8_t* skip = p.skip + (size_t)prob * cap;
    const uint32_t budget = (uint32_t)p.budget;  // the host keeps a launch's budget below 2^31
    const uint32_t depth = p.freeze ? (uint32_t)kQRing : kDepthGrow;

    if (tid < (uint32_t)kScanWaves) sh.wave_done[tid] = 0;
    if (tid == 0) {
        sh.sampled = 0;
        sh.resolved = 0;
        sh.committed = st0.n_nodes;
        sh.stop_flag = 0;
        sh.heartbeat = 0;
        sh.mabs_bits = 0;
    }
    __syncthreads();

    if (wave < kScanWaves) {
        // ================================================================= scanner waves
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
                const uint32_t ab = lf32_bits(f) & 0x7FFFFFFFu;
                mab = ab > mab ? ab : mab;
                tr[k][s] = live ? f : __builtin_inff();
            }
        }
        __hip_atomic_fetch_max(&sh.mabs_bits, mab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();  // the resolver reads mabs_bits after this barrier (the second and last of the launch)

        uint32_t seen_sampled = 0;
        uint64_t t_wait = 0, t_work = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;
        for (uint32_t j = 0; j < budget; j += kPassQ) {
            const uint32_t nb = (budget - j < (uint32_t)kPassQ) ? (budget - j) : (uint32_t)kPassQ;
            const uint32_t need = j + nb;
            // wait until the pass's queries are sampled; give up only when the resolver's heartbeat stands still
            uint32_t hb_seen = lds_peek(&sh.heartbeat);
            for (uint32_t spins = 0; seen_sampled < need; ++spins) {
                if (lds_peek(&sh.stop_flag) != 0) break;
                if (spins > kMaxSpins) {
                    const uint32_t hb = lds_peek(&sh.heartbeat);
                    if (hb == hb_seen) break;
                    hb_seen = hb;
                    spins = 0;
                }
                seen_sampled = uni(lds_peek(&sh.sampled));
                if (seen_sampled < need) __builtin_amdgcn_s_sleep(2);
            }
            if (seen_sampled < need) break;  // stop requested
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }
#if OXHIP_LANES_PRIO
            // two scanner waves (w and w ^ 4) share a SIMD and the older one wins every issue conflict: it would race ahead
            // and idle at the ring while the younger one -- the wave everybody ends up waiting for -- crawls.  The wave that
            // is not ahead of its partner takes the higher priority for this pass: the pair stays interleaved.
            if (j <= uni(lds_peek(&sh.wave_done[wave ^ 4u]))) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
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
                            for (int k = 0; k < D; ++k) tr[k][s] = (float)sh.newn[k][i & (kNRing - 1)];   // +inf stays +inf
                        }
                    }
                }
            }
            n_local = nc;
            // the pass's queries, two per packed register
            lf32x2 q[kPassQ / 2][D];
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) {
                const uint32_t slot = (j + ((uint32_t)b < nb ? (uint32_t)b : 0u)) & (kQRing - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) q[b / 2][k][b % 2] = lbits_f32(uni(lf32_bits(sh.qf[slot][k])));   // wave-uniform: scalar registers
            }
            const uint32_t nslots = Lay::slots_in_use(wave, nc);
            LScreen sc[kPassQ];
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) sc[b] = LScreen{kLKeyInf, kLKeyInf};
            // screen (rrt.rs:187-196 in binary32): groups of kGroup slots under one uniform branch
#pragma unroll
            for (int g0 = 0; g0 < S; g0 += group_len<S, C>(g0)) {
                if ((uint32_t)g0 < nslots) {
#pragma unroll
                    for (int s = g0; s < g0 + group_len<S, C>(g0); ++s) {
#pragma unroll
                        for (int bp = 0; bp < kPassQ / 2; ++bp) {
                            lf32x2 e = (lf32x2){tr[0][s], tr[0][s]} - q[bp][0];
                            lf32x2 acc = e * e;
#pragma unroll
                            for (int k = 1; k < D; ++k) {
                                e = (lf32x2){tr[k][s], tr[k][s]} - q[bp][k];
                                acc = __builtin_elementwise_fma(e, e, acc);
                            }
                            lscreen_push(sc[2 * bp], acc[0], (uint32_t)s);
                            lscreen_push(sc[2 * bp + 1], acc[1], (uint32_t)s);
                        }
                    }
                }
            }
            // reduce: the wave's smallest key, its lane, and the smallest of everything else
            uint32_t k1w[kPassQ], k2w[kPassQ];
            int wl[kPassQ];
#pragma unroll
            for (int b0 = 0; b0 < kPassQ; b0 += 4) {
                uint32_t t4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) t4[t] = sc[b0 + t].b1;
                lanes_min4_u32(t4);
#pragma unroll
                for (int t = 0; t < 4; ++t) k1w[b0 + t] = t4[t];
            }
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) {
                const uint64_t eqm = __ballot(sc[b].b1 == k1w[b]);
                wl[b] = __ffsll((unsigned long long)eqm) - 1;   // eqm != 0: the minimum is attained
            }
#pragma unroll
            for (int b0 = 0; b0 < kPassQ; b0 += 4) {
                uint32_t t4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) t4[t] = (int)lane == wl[b0 + t] ? sc[b0 + t].h2 : sc[b0 + t].b1;
                lanes_min4_u32(t4);
#pragma unroll
                for (int t = 0; t < 4; ++t) k2w[b0 + t] = t4[t];
            }
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) {
                if ((uint32_t)b < nb) {
                    const uint32_t slot = (j + (uint32_t)b) & (kQRing - 1);
                    if (lane == 0) {
                        LanePub out;
                        out.k1 = k1w[b];
                        out.k2 = k2w[b];
                        out.i1 = Lay::node_index(wave, (uint32_t)wl[b], k1w[b] & kLSlotMask);
                        out.nc = nc;
                        sh.pub[wave][slot] = out;
                    }
                }
            }
            if (lane == 0) lds_post(&sh.wave_done[wave], need);   // after the records (LDS is in order within a wave)
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) { p.dbg[16 + wave] = t_wait; p.dbg[24 + wave] = t_work; }
        return;
    }

    if (wave == (uint32_t)kScanWaves + 1u) {
        // ================================================================= sampler wave
        // rrt.rs:177-184 for the whole launch, ahead of everybody: the queries depend on the RNG stream only, never on
        // the tree.  A ring slot is reused only after its previous tenant was resolved (`resolved`, posted by the resolver).
        double goal_c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
        __syncthreads();  // the launch's second barrier (see the scanners)
        RngWindow rng;
        rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st0.draws);
        uint32_t js = 0;
        while (js < budget) {
            uint32_t jr_seen = 0;
            bool go = false;
            uint32_t hb_seen = lds_peek(&sh.heartbeat);
            for (uint32_t spins = 0;; ++spins) {
                if (lds_peek(&sh.stop_flag) != 0) break;
                if (spins > kMaxSpins) {   // give up only when the resolver's heartbeat stands still
                    const uint32_t hb = lds_peek(&sh.heartbeat);
                    if (hb == hb_seen) break;
                    hb_seen = hb;
                    spins = 0;
                }
                jr_seen = uni(lds_peek(&sh.resolved));
                if (js - jr_seen <= depth / 2) { go = true; break; }   // half the window is free: refill it
                __builtin_amdgcn_s_sleep(2);
            }
            if (!go) break;  // stop requested (or a protocol bug: the resolver's own guard reports it)
            uint32_t m = jr_seen + depth - js;  // free window slots
            if (m > 64u) m = 64u;
            if (m > 8u) m -= (js + m) & 7u;     // the scanners consume whole passes of 8: end the batch on a pass boundary
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
            if (!sample_lanes<DIM>(rng, p, goal_c, m, lane, sh, js)) {
                for (uint32_t b = 0; b < m; ++b) {  // (never expected) a redraw ran past the window: one by one
                    double qn[D];
                    sample_state<D, false>(rng, p, DIM, goal_c, qn);
                    const uint32_t slot = (js + b) & (kQRing - 1);
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            sh.q[k][slot] = qn[k];
                            sh.qf[slot][k] = (float)qn[k];
                        }
                        sh.pos_after[slot] = rng.pos;
                    }
                }
            }
            js += m;
            if (lane == 0) lds_post(&sh.sampled, js);
        }
        return;
    }

    // ===================================================================== resolver wave: lane j <-> query jr + j
    __builtin_amdgcn_s_setprio(3);  // the youngest wave of its SIMD would otherwise queue behind two scanners
    ProblemState st = st0;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    const double goal_thr = p.goal_thr[prob];
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
    const bool extras = nobs > ns64;  // spheres beyond the first 64 and every box: always stepped, never filtered
    // sphere `lane` in this lane's registers (the whole-wave motion check of the exact path) and in LDS (the lane-parallel checks)
    double oc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) oc[k] = lane < ns64 ? p.sph_c[(size_t)k * p.n_spheres + lane] : 0.0;
    const double othr = lane < ns64 ? p.sph_thr[lane] : -1.0;
    const double ofilt = lane < ns64 ? p.sph_filt[lane] : -1.0;
#pragma unroll
    for (int k = 0; k < D; ++k) sh.obs[k][lane] = oc[k];
    sh.obs[D][lane] = othr;
    sh.obs[D + 1][lane] = ofilt;

    // P^lane for the batched checksum (H <- H P^m + sum_j g_j P^(m-1-j)); P^64 for a full batch
    uint64_t pw = 1;
    {
        uint64_t base = kFnvPrime;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            if ((lane >> b) & 1u) pw *= base;
            base *= base;
        }
    }
    const uint64_t pw64 = uni64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pw >> 32), 63) << 32 |
                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pw, 63)) * kFnvPrime);

    uint64_t draws_done = st.draws;
    uint32_t n = st.n_nodes;
    uint32_t jr = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    uint64_t n_amb = 0, n_rounds = 0, n_lanes = 0, n_cut_conflict = 0, t_wait = 0, t_work = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;

    __syncthreads();  // pairs with the scanners' second barrier: mabs_bits is final
    LMargins mg;
    {
        // M: the tree as loaded (binary32 roundings, hence the 1 + 2^-23), the bounds and the goal centre
        double m = (double)lbits_f32(lds_peek(&sh.mabs_bits)) * (1.0 + 0x1p-23);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            m = fmax(m, fmax(fabs(p.lo[k]), fabs(p.hi[k])));
            m = fmax(m, fabs(goal_c[k]));
        }
        m = unid(m) * 1.001;   // interpolation rounding over any chain of inserts
        const double u = 0x1p-24;
        mg.usable = m < 1e15;  // also false for NaN / inf
        mg.a2 = 2.0 * (sqrt((double)D) * 4.1 * u * m + 1e-18);
        const double r2 = 2.0 * (0x1p-19 + (double)(D + 2) * u) + 0x1p-21;   // + the binary32 square root of the clear test
        mg.r_lo = 1.0 - r2;
        mg.r_hi = 1.0 + r2;
    }

    // coordinates of node i as the resolver may read them: the LDS ring for the young ones (their global stores may still be
    // in flight; duplicates hold +inf there, but a duplicate is never anybody's nearest node), the persistent binary64 copy
    // in global memory for nodes this wave wrote at least 64 commits ago or that predate the launch
    auto node_coord = [&](int k, uint32_t i, uint32_t n_now) -> double {
        const bool young = i >= st0.n_nodes && i + (uint32_t)(kNRing - 64) >= n_now;
        const double g = __hip_atomic_load(&tree[(size_t)k * cap + (young ? 0u : i)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double l = sh.newn[k][i & (kNRing - 1)];
        return young ? l : g;
    };

    while (true) {
        if (jr >= budget) { stop = 1; break; }
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }

        // ---- how many queries have all eight scanners published?
        // A round costs about the same for 8 lanes as for 64 (the filter, the sqrt and the division are per round), and this
        // wave shares its SIMD with two scanners: it waits for a good part of a full round -- half the window the sampler
        // keeps ahead -- unless the budget ends sooner.
        uint32_t avail = 0;
        {
            uint32_t spins = 0;
            for (;;) {
                // never wait for more than the scanners are going to publish without further sampling: whole passes of 8
                // of what has been sampled (the sampler refills by its own rule and may be waiting for this wave)
                const uint32_t sampled_now = uni(lds_peek(&sh.sampled));
                const uint32_t coming = (sampled_now >= budget ? budget : (sampled_now & ~7u)) - jr;
                uint32_t want = depth / 2;
                if (want > coming) want = coming;
                const uint32_t d = lane < (uint32_t)kScanWaves ? lds_peek(&sh.wave_done[lane & (kScanWaves - 1)]) : 0xFFFFFFFFu;
                const uint32_t done_all = wave_min_u32(d);
                avail = done_all - jr;
                if ((avail != 0 && avail >= want) || spins > kMaxSpins) break;
                __builtin_amdgcn_s_sleep(4);
                ++spins;
            }
            if (avail == 0) { stop = 4; break; }  // OXHIP_STOP_INTERNAL: a scanner never published (bug guard)
        }
        uint32_t m = avail < 64u ? avail : 64u;
        if (m > budget - jr) m = budget - jr;
        if (lane == 0) lds_bump(&sh.heartbeat);
        if (STAMP) { ++n_rounds; n_lanes += m; uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }

        // ---- parallel phase: lane j resolves query jr + j against the tree of n nodes
        const bool act = lane < m;
        const uint32_t slot = (jr + (act ? lane : 0u)) & (kQRing - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = sh.q[k][slot];
        const uint64_t pos_after_l = sh.pos_after[slot];
        // the eight waves' screens: smallest key, the wave that holds it, the smallest of everything else, oldest snapshot
        uint32_t K1 = kLKeyInf, K2 = kLKeyInf, cand = 0, bmin = 0xFFFFFFFFu;
        {
            LanePub rec[kScanWaves];
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) rec[w] = sh.pub[w][slot];
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                K1 = rec[w].k1 < K1 ? rec[w].k1 : K1;
                bmin = rec[w].nc < bmin ? rec[w].nc : bmin;
            }
            bool taken = false;
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                const bool win = !taken && rec[w].k1 == K1;   // the first wave attaining the minimum
                const uint32_t other = win ? rec[w].k2 : rec[w].k1;
                K2 = other < K2 ? other : K2;
                cand = win ? rec[w].i1 : cand;
                taken = taken || win;
            }
        }
        if (__ballot(act && (n - bmin > (uint32_t)(kNRing - 64) || bmin > n)) != 0) { stop = 4; break; }  // ring would have wrapped (bug guard)
        bool clear;
        {
            const float v1 = lbits_f32(K1 & ~kLSlotMask), v2 = lbits_f32(K2 & ~kLSlotMask);
            // v_sqrt_f32 (1 ulp; a subnormal argument may come back as 0, which the 1e-18 inside A covers): its 2^-22 is in r_lo / r_hi
            const double d1 = (double)__builtin_amdgcn_sqrtf(v1), d2 = (double)__builtin_amdgcn_sqrtf(v2);
            // (+inf second: d2 = +inf and the test holds; +inf or NaN first: it fails)
            clear = act && mg.usable && K1 != kLKeyInf && (d2 * mg.r_lo - mg.a2 > d1 * mg.r_hi + mg.a2);
        }
        // the screen's winner with its binary64 coordinates and d2
        const uint32_t cand_i = (clear && cand < n) ? cand : 0u;
        double cc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) cc[k] = node_coord(k, cand_i, n);
        const double pb = clear ? dist2<D>(cc, q, DIM) : __builtin_inf();
        // nodes committed after the oldest snapshot: every lane folds the ones its own query has not seen
        Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
        {
            const uint32_t lo = wave_min_u32(act ? bmin : 0xFFFFFFFFu);
            for (uint32_t i = lo; i < n; ++i) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.newn[k][i & (kNRing - 1)];
                const double d = dist2<D>(c, q, DIM);
                if (act && i >= bmin) scan_push(pd, d, i);  // ascending i: ties keep the lower index
            }
        }
        const double g = pd.b1 < pb ? pd.b1 : pb;
        const uint32_t hb = hi32(g) + 1;
        const bool nearS = clear && hi32(pb) <= hb;
        const bool nearP = pd.slot != kNoNode && hi32(pd.b1) <= hb;
        const uint32_t nearest = nearS ? cand : pd.slot;
        // ambiguous iff the screen could not name a winner, or a second node is within a rounding of the binary64 minimum
        const bool amb = act && (!clear || (nearP && pd.slot != nearest) || pd.h2 <= hb || (!nearS && !nearP));
        double q_near[D], qn[D], mid[D];
        {
            const uint32_t ni = (nearest == kNoNode ? 0u : nearest) & (kNRing - 1);
#pragma unroll
            for (int k = 0; k < D; ++k) q_near[k] = nearS ? cc[k] : sh.newn[k][ni];
        }
        steer<DIM>(p, false, g, q_near, q, qn);
        const bool dup = g == 0.0;
        // check_motion (rrt.rs:90-116): the midpoint filter names the spheres the segment can touch at all ...
        bool bad = false;
        if (nobs > 0) {
            lerp<DIM>(q_near, qn, 0.5, mid, DIM);
            // (eight spheres per trip: their LDS reads -- wave-uniform addresses, broadcast -- are issued together, so the
            // loop runs at the arithmetic's pace instead of one LDS round trip per sphere; slots beyond ns64 hold -1 thresholds)
            uint32_t maybe_lo = 0, maybe_hi = 0;
            for (uint32_t o0 = 0; o0 < ns64; o0 += 8) {
                double c[8][D], f[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#pragma unroll
                    for (int k = 0; k < D; ++k) c[t][k] = sh.obs[k][o0 + t];
                    f[t] = sh.obs[D + 1][o0 + t];
                }
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) bits |= sphere_maybe_hit<DIM>(c[t], f[t], mid) ? (1u << t) : 0u;
                if (o0 < 32) maybe_lo |= bits << o0; else maybe_hi |= bits << (o0 - 32);
            }
            const uint64_t maybe = ((uint64_t)maybe_hi << 32) | maybe_lo;
            const bool need = act && !amb && (maybe != 0 || extras);
            if (__ballot(need) != 0) {
                // ... and every lane steps through its own motion against just those (is_valid is pure: testing all states
                // equals the reference's first-invalid early exit)
                const double dist = sqrt(dist2<DIM>(q_near, qn, DIM));
                const uint32_t nsteps = num_steps_u32(dist, p.res);
                const uint32_t steps_l = need ? (nsteps <= 1 ? 1u : nsteps) : 0u;
                const uint32_t smax = wave_max_u32(steps_l);
                const double dn = (double)nsteps;
                for (uint32_t s = 1; s <= smax && s != 0; ++s) {
                    const bool on = s <= steps_l;
                    double x[D];
                    {
                        const double t = (double)s / dn;
                        double xi[D];
                        lerp<DIM>(q_near, qn, t, xi, DIM);
#pragma unroll
                        for (int k = 0; k < D; ++k) x[k] = nsteps <= 1 ? qn[k] : xi[k];   // num_steps <= 1: is_valid(to) only
                    }
                    uint64_t rem = on ? maybe : 0ull;
                    while (__ballot(rem != 0) != 0) {
                        const bool has = rem != 0;
                        const uint32_t o = has ? (uint32_t)(__ffsll((unsigned long long)rem) - 1) : 0u;
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                        bad = bad || (has && !(dist2<D>(c, x, DIM) > sh.obs[D][o]));
                        rem &= rem - 1;
                    }
                    for (uint32_t jx = ns64; jx < nobs; ++jx) bad = bad || (on && obstacle_hit<DIM>(p, DIM, x, jx));
                    if ((s & 63u) == 0 && lane == 0) lds_bump(&sh.heartbeat);
                }
            }
        }
        const bool ok = act && !bad;
        const bool ins = !p.freeze;

        // ---- the prefix this round may commit
        uint32_t cut = m;
        int32_t stop_after = -1;
        const uint64_t ambm = __ballot(amb);
        if (ambm != 0) cut = (uint32_t)(__ffsll((unsigned long long)ambm) - 1);
        const uint64_t okm = __ballot(ok);
        uint64_t hitm = 0;
        if (ins) {
            // node cap: query j is processed only while the tree has room (rrt.rs has no cap; checked before any draw of the iteration)
            const uint64_t capm = __ballot(act && n + (uint32_t)__popcll(okm & below_mask(lane)) >= p.max_nodes);
            if (capm != 0) {
                const uint32_t c = (uint32_t)(__ffsll((unsigned long long)capm) - 1);
                if (c <= cut) { cut = c; stop_after = 2; }
            }
            hitm = __ballot(ok && dist2<D>(qn, goal_c, DIM) <= goal_thr);
            if (p.stop_at_goal && hitm != 0) {
                const uint32_t c = (uint32_t)__ffsll((unsigned long long)hitm);   // first hit lane + 1
                if (c <= cut) { cut = c; stop_after = 0; }
            }
            // a node accepted earlier in the round that is (nearly) as close to a later query as that query's nearest
            // node changes that query's result: the prefix ends before the first such query
            const uint64_t newm = __ballot(ok && !dup);
            for (uint64_t rest = newm; rest != 0; rest &= rest - 1) {
                const int i = __ffsll((unsigned long long)rest) - 1;
                if ((uint32_t)i + 1u >= cut) break;
                double ca[D];
#pragma unroll
                for (int k = 0; k < D; ++k) ca[k] = readlane_f64(qn[k], i);
                const uint64_t cm = __ballot(act && lane > (uint32_t)i && hi32(dist2<D>(ca, q, DIM)) <= hb);
                if (cm != 0) {
                    const uint32_t c = (uint32_t)(__ffsll((unsigned long long)cm) - 1);
                    if (c < cut) { cut = c; stop_after = -1; if (STAMP) ++n_cut_conflict; }
                }
            }
        }

        // ---- commit lanes [0, cut) in query order
        if (cut > 0) {
            const uint64_t cutm = first_n_mask(cut);
            const bool mine = lane < cut;
            if (ins) {
                const uint32_t idx = n + (uint32_t)__popcll(okm & below_mask(lane));
                if (mine && ok) {
                    // insert (rrt.rs:213-217): LDS hand-off to the owning scanner lane + the persistent copy.  A node at
                    // distance 0 from its nearest node repeats that node's coordinates, and the strict '<' of rrt.rs:192
                    // can never prefer it over the lower index: the scanners keep +inf for it.
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        sh.newn[k][idx & (kNRing - 1)] = dup ? __builtin_inf() : qn[k];
                        tree[(size_t)k * cap + idx] = qn[k];
                    }
                    parent[idx] = (int32_t)nearest;
                    skip[idx] = dup ? 1 : 0;
                }
                // goal test (rrt.rs:220-223): the first hit in query order
                const uint64_t hits = hitm & cutm;
                if (hits != 0 && st.goal_node < 0)
                    st.goal_node = (int32_t)__builtin_amdgcn_readlane((int)idx, __ffsll((unsigned long long)hits) - 1);
                n += (uint32_t)__popcll(okm & cutm);
                if (lane == 0) lds_post(&sh.committed, n);
            }
            // checksum: H <- H P^cut + sum_{j < cut} g_j P^(cut-1-j)
            {
                const uint64_t gd = iter_digest<D>(nearest, qn, DIM, ok);
                const int src = mine ? (int)(cut - 1u - lane) : 0;
                const uint64_t w = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(pw >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)pw, src, 64);
                const uint64_t sum = wave_sum_u64(mine ? gd * w : 0ull);
                const uint64_t pc = cut >= 64u ? pw64
                                               : uni64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pw >> 32), (int)cut) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pw, (int)cut));
                st.checksum = st.checksum * pc + sum;
            }
            st.iterations += cut;
            st.accepted += (uint64_t)__popcll(okm & cutm);
            draws_done = uni64((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_l, (int)(cut - 1u)) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_l >> 32), (int)(cut - 1u)) << 32));
            jr += cut;
            if (lane == 0) lds_post(&sh.resolved, jr);   // the sampler may hand the slots out again
        }
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        if (stop_after >= 0) { stop = stop_after; break; }
        if (cut == m || ambm == 0 || (uint32_t)(__ffsll((unsigned long long)ambm) - 1) != cut) continue;

        // ---- the lane at `cut` is ambiguous: that one query by the reference's own loop -- post-sqrt compare with
        //      lowest-index ties -- over the persistent binary64 copy of the tree in global memory
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }
        {
            if (STAMP) ++n_amb;
            const uint32_t slot1 = jr & (kQRing - 1);
            double q1[D];
#pragma unroll
            for (int k = 0; k < D; ++k) q1[k] = unid(sh.q[k][slot1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            Exact e{__builtin_inf(), kNoNode};
            for (uint32_t i = lane; i < n; i += 64) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    c[k] = __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double d = sqrt(dist2<D>(c, q1, DIM));
                if (d < e.dist) { e.dist = d; e.idx = i; }
            }
            e = exact_wave_reduce(e);
            const uint32_t nearest1 = uni(e.idx);
            double qn1[D], q_near1[D];
#pragma unroll
            for (int k = 0; k < D; ++k)
                q_near1[k] = unid(__hip_atomic_load(&tree[(size_t)k * cap + nearest1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const double dist1 = unid(e.dist);
            const bool dup1 = dist1 == 0.0;
            steer<DIM>(p, true, dist1, q_near1, q1, qn1);
            bool ok1 = true;
            if (nobs > 0) {
                double mid1[D];
                lerp<DIM>(q_near1, qn1, 0.5, mid1, DIM);
                if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid1)) != 0 || extras)
                    ok1 = motion_lanes<DIM>(p, lane, q_near1, qn1, oc, othr, ofilt, ns64);
            }
            st.checksum = uni64(chk_push(st.checksum, iter_digest<D>(nearest1, qn1, DIM, ok1)));
            st.iterations++;
            draws_done = uni64(sh.pos_after[slot1]);
            bool hit1 = false;
            if (ok1) {
                st.accepted++;
                if (!p.freeze) {
                    const uint32_t i = n;
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            sh.newn[k][i & (kNRing - 1)] = dup1 ? __builtin_inf() : qn1[k];
                            tree[(size_t)k * cap + i] = qn1[k];
                        }
                        parent[i] = (int32_t)nearest1;
                        skip[i] = dup1 ? 1 : 0;
                    }
                    ++n;
                    if (lane == 0) lds_post(&sh.committed, n);
                    if (dist2<D>(qn1, goal_c, DIM) <= goal_thr) {
                        if (st.goal_node < 0) st.goal_node = (int32_t)i;
                        hit1 = true;
                    }
                }
            }
            jr += 1;
            if (lane == 0) lds_post(&sh.resolved, jr);
            if (hit1 && p.stop_at_goal) { stop = 0; break; }
        }
    }
    if (lane == 0) {
        lds_post(&sh.stop_flag, 1);
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
        if (STAMP && p.dbg && prob == 0) {
            p.dbg[4] = n_amb; p.dbg[5] = n_rounds; p.dbg[6] = n_lanes; p.dbg[7] = st.iterations; p.dbg[12] = n_cut_conflict; p.dbg[1] = t_wait; p.dbg[2] = t_work;
        }
    }
}

#ifndef OXHIP_LANES_S
#define OXHIP_LANES_S 21
#define OXHIP_LANES_C 18
#endif
constexpr int kLS = OXHIP_LANES_S, kLC = OXHIP_LANES_C;   // register rows of the six heavy / of every scanner wave

static int pick_slots_lanes(uint32_t cap) {
    const uint32_t need = (cap + kScanThreads - 1) / kScanThreads;
    if (need <= 4) return 4;
    if (cap <= Layout<kLS, kLC, false>::kCapacity) return kLS;
    return 0;
}

bool lanes_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pick_slots_lanes(cap) != 0; }

void launch_rrt_lanes(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kLanesThreads);
    const int s = pick_slots_lanes(p.cap);
#define OXHIP_LAUNCH(DIM_, S_, C_)                                                                            \
    do {                                                                                                      \
        if (p.dbg) hipLaunchKernelGGL((rrt_lanes_kernel<DIM_, S_, C_, true>), grid, block, 0, stream, p);     \
        else hipLaunchKernelGGL((rrt_lanes_kernel<DIM_, S_, C_, false>), grid, block, 0, stream, p);          \
    } while (0)
    if (p.dim == 3) {
        if (s == 4) OXHIP_LAUNCH(3, 4, 4); else OXHIP_LAUNCH(3, kLS, kLC);
    } else {
        if (s == 4) OXHIP_LAUNCH(2, 4, 4); else OXHIP_LAUNCH(2, kLS, kLC);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
