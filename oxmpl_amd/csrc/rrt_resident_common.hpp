// rrt_resident_common.hpp -- pieces shared by the register-resident kernels (rrt_resident.hip,
// rrt_pruned.hip): DPP reductions, the scan state, the scanner/resolver LDS ring, the tree layout,
// steer / motion check / conservative filter, the lane-parallel sampler.
#pragma once

#include "rrt_device.hpp"

namespace oxhip {


// ---- wave64 min of an f64 with DPP (VALU only, no LDS crossbar); result is wave-uniform
__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bound3(double g) {
    return __longlong_as_double((long long)((uint64_t)__double_as_longlong(g) + 3));
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)__double2hiint(v); }

// Per-lane scan state.  d2 >= +0, so the high dword orders like the value: h2 tracks the second
// smallest HIGH DWORD (with multiplicity) in one v_med3_u32 per node.  Two d2 whose square roots
// could coincide differ by <= 3 ulps, hence their high dwords differ by <= 1: `h2 <= hi(b1) + 1`
// is a conservative (never missing, ~1e-6 false-positive) near-tie detector.
struct Scan {
    double b1;      // smallest d2 of this lane
    uint32_t slot;  // its slot (lowest index: slots are visited in increasing index order)
    uint32_t h2;    // second smallest high dword
};
__device__ __forceinline__ void scan_push(Scan& v, double d, uint32_t s) {
    v.h2 = umed3(hi32(d), hi32(v.b1), v.h2);
    const bool lt = d < v.b1;  // strict: the earlier (lower) index keeps exact ties
    v.b1 = lt ? d : v.b1;
    v.slot = lt ? s : v.slot;
}

template <int DIM>
struct alignas(16) WavePub {  // one wave's nearest-neighbour candidate (head = one 16-byte LDS store)
    double b1;        // its smallest d2
    uint32_t i1;      // lowest index attaining it
    uint32_t amb_nc;  // bit 0: the wave saw another d2 whose high dword is within 1 of b1's;
                      // bits 1..: the tree size this scan covered (the wave's snapshot of `committed`)
    double c[DIM];    // the candidate's coordinates (written by the owning lane)
};
template <int DIM>
struct WaveExact {
    double dist;
    uint32_t idx;
    uint32_t pad;
    double c[DIM];
};

// lane-predicated store of the owning lane's slot `slot` (wave-uniform) into LDS.  Register arrays
// cannot be indexed dynamically (the compiler would demote the whole tree to scratch), so the slot
// is matched by uniform branches ending in compile-time indices: groups of 4, then the slot.
template <int DIM, int S>
__device__ __forceinline__ void store_slot(const double (&tr)[DIM][S], uint32_t slot, bool mine, double* dst) {
    const uint32_t grp = slot >> 2, sub = slot & 3u;
#pragma unroll
    for (int g = 0; g < (S + 3) / 4; ++g) {
        if (grp == (uint32_t)g) {  // uniform
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * g + t < S) {
                    if (sub == (uint32_t)t) {  // uniform
                        if (mine) {
#pragma unroll
                            for (int k = 0; k < DIM; ++k) dst[k] = tr[k][4 * g + t];
                        }
                    }
                }
            }
        }
    }
}

// Node -> (scanner thread, register slot).  Waves w, w+4, w+8 share a SIMD (waves are dealt to the
// four SIMDs cyclically), so scanner waves 0 and 4 sit beside the resolver (wave 8) and would be
// ~20 % slower than the other six.  The S = 21 layout therefore gives them 17 slots and the other
// six waves 21: rows 0..16 span all 512 threads, rows 17..20 only the 384 threads of waves
// 1,2,3,5,6,7 (17*512 + 4*384 = 10,240 nodes).  Smaller instantiations use the plain even layout.
// (C = rows every wave holds; the binary32 kernel, whose scans are cheaper relative to the resolver's work, uses 22 / 14.)
// OLDER = true: the extra rows go to waves 0..3 instead -- the older wave of each SIMD, which wins every issue conflict
// with its younger neighbour (w + 4) and would otherwise idle at the ring while that one is still scanning.
template <int S, int C = (S == 21 ? 17 : S), bool OLDER = false>
struct Layout {
    static constexpr bool kUneven = (C != S);
    static constexpr uint32_t kCommon = (uint32_t)C;   // rows every wave holds
    static constexpr uint32_t kHeavyThreads = OLDER ? 256 : 384;
    static constexpr uint32_t kCapacity = (uint32_t)C * 512u + (uint32_t)(S - C) * kHeavyThreads;
    __device__ static __forceinline__ bool heavy(uint32_t wave) { return !kUneven || (OLDER ? wave < 4u : (wave & 3u) != 0); }
    __device__ static __forceinline__ uint32_t node_index(uint32_t wave, uint32_t lane, uint32_t slot) {
        if (slot < kCommon) return slot * 512u + wave * 64u + lane;
        const uint32_t hw = OLDER ? wave : wave - 1u - (wave > 4u ? 1u : 0u);  // waves 1,2,3,5,6,7 -> 0..5
        return kCommon * 512u + (slot - kCommon) * kHeavyThreads + hw * 64u + lane;
    }
    __device__ static __forceinline__ void locate(uint32_t i, uint32_t& thread, uint32_t& slot) {
        if (i < kCommon * 512u) { thread = i & 511u; slot = i >> 9; return; }
        const uint32_t r = i - kCommon * 512u, c = r % kHeavyThreads, hw = c >> 6;
        slot = kCommon + r / kHeavyThreads;
        thread = OLDER ? c : (hw + 1u + (hw >= 3u ? 1u : 0u)) * 64u + (c & 63u);
    }
    __device__ static __forceinline__ uint32_t slots_in_use(uint32_t wave, uint32_t n) {
        if (n <= kCommon * 512u) return (n + 511u) >> 9;
        return heavy(wave) ? kCommon + (n - kCommon * 512u + kHeavyThreads - 1u) / kHeavyThreads : kCommon;
    }
};

// ------------------------------------------------------------------------------------------
// Asynchronous pipeline.  kScanWaves scanner waves own the tree (node i in thread i % 512, slot
// i / 512) and stream queries from an LDS ring without ever meeting at a workgroup barrier; one
// resolver wave samples the queries ahead, consumes the scanners' per-query results in order,
// covers the nodes committed after a scan's snapshot from its own lanes (the last 64 nodes, one
// per lane), steers, checks the motion and commits.  Sequential semantics are the resolver's:
// iteration k sees exactly the tree left by iterations < k, as in rrt.rs:170-225.
// ------------------------------------------------------------------------------------------
constexpr int kScanWaves = 8;
constexpr int kScanThreads = kScanWaves * 64;        // 512
constexpr int kPipeThreads = kScanThreads + 64;      // + the resolver wave
constexpr int kRing = 16;                            // queries in flight (power of two)
constexpr int kBatch = 4;                            // queries one scanner pass covers
constexpr int kGroup = 4;                            // slots per uniform branch of the scan
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
constexpr uint32_t kMaxSpins = 1u << 22;             // ~0.5 s of polling: turns a protocol bug into an error, not a hang

template <int DIM>
struct QSlot {
    double q[DIM];
    uint64_t pos_after;  // stream position after this query's draws
};

template <int DIM>
struct PipeShared {
    uint32_t rng_buf[16][64];
    QSlot<DIM> qring[kRing];
    WavePub<DIM> pub[kRing][kScanWaves];
    uint32_t done[kRing];                // scanner waves that have published this slot
    double newn[64][DIM];                // the last 64 committed nodes, node i at i & 63 (+inf for skipped duplicates)
    double obs[DIM + 1][64];             // first 64 spheres for the row-parallel filter: centre, filter threshold
    uint32_t sampled;                    // queries sampled so far   (monotonic)
    uint32_t resolved;                   // queries resolved so far  (monotonic)
    uint32_t committed;                  // tree size                (monotonic)
    uint32_t stop_flag;                  // resolver -> scanners: leave
};

// LDS executes one wave's instructions in order and is a single pipeline per CU, so "write data,
// then write flag" / "read flag, then read data" need compiler ordering only.
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) {
    uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_post(uint32_t* p, uint32_t v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_bump(uint32_t* p) {
    asm volatile("" ::: "memory");
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// steer (rrt.rs:199-208): q_new = q when within max_distance of q_near, else the point at max_distance.
// `have_dist`: g_or_dist is the post-sqrt distance (exact path); otherwise it is d2 and the test
// sqrt(d2) > max_distance is the exact threshold compare d2 > t_steer.
template <int DIM>
__device__ __forceinline__ void steer(const DevParams& p, bool have_dist, double g_or_dist, const double q_near[DIM],
                                      const double q[DIM], double q_new[DIM]) {
    const bool far = have_dist ? (g_or_dist > p.max_distance) : (g_or_dist > p.t_steer);
    if (far) {
        const double md = have_dist ? g_or_dist : sqrt(g_or_dist);
        const double t = p.max_distance / md;
        lerp<DIM>(q_near, q, t, q_new, DIM);
    } else {
#pragma unroll
        for (int k = 0; k < DIM; ++k) q_new[k] = q[k];
    }
}

// check_motion (rrt.rs:90-116) by the whole resolver wave: one obstacle per lane (registers), the
// interpolated states in sequence, stopping at the first invalid one like the reference.
template <int DIM>
__device__ __forceinline__ bool motion_full(const DevParams& p, uint32_t lane, const double q_near[DIM], const double q_new[DIM],
                                            const double oc[DIM], double othr, uint32_t ns64) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const double dist = sqrt(dist2<DIM>(q_near, q_new, DIM));
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    bool bad = false;
    if (nsteps <= 1) {
        bad = !(dist2<DIM>(oc, q_new, DIM) > othr);
        for (uint32_t j = ns64 + lane; j < nobs; j += 64) bad = bad || obstacle_hit<DIM>(p, DIM, q_new, j);
    } else {
        const double dn = (double)nsteps;
        const double tl = (double)(lane + 1) / dn;  // lane s-1 holds s / nsteps (one division for all steps)
        for (uint32_t s = 1; s <= nsteps; ++s) {
            const double t = (s <= 64) ? readlane_f64(tl, (int)(s - 1)) : ((double)s / dn);
            double x[DIM];
            lerp<DIM>(q_near, q_new, t, x, DIM);
            bad = bad || !(dist2<DIM>(oc, x, DIM) > othr);
            for (uint32_t j = ns64 + lane; j < nobs; j += 64) bad = bad || obstacle_hit<DIM>(p, DIM, x, j);
            if (__ballot(bad) != 0) break;
        }
    }
    return __ballot(bad) == 0;
}

// check_motion (rrt.rs:90-116) for a steered motion (length <= max_distance), lane-parallel over the STEPS: the midpoint
// filter below first names the spheres the segment can touch at all (usually one or two of 64), then lane s-1 tests
// interpolated state s against just those -- one pass instead of one pass per step.  is_valid is pure and the filter only
// ever removes spheres that provably cannot be hit, so the verdict equals the reference's first-invalid early exit.
// Spheres beyond the first 64 and all boxes are tested against every step, unfiltered.
template <int DIM>
__device__ __forceinline__ bool sphere_maybe_hit(const double c[DIM], double filt, const double mid[DIM]);
template <int DIM>
__device__ __forceinline__ bool motion_lanes(const DevParams& p, uint32_t lane, const double q_near[DIM], const double q_new[DIM],
                                             const double oc[DIM], double othr, double ofilt, uint32_t ns64) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const double dist = sqrt(dist2<DIM>(q_near, q_new, DIM));
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    bool bad = false;
    if (nsteps <= 1) {
        bad = !(dist2<DIM>(oc, q_new, DIM) > othr);
        for (uint32_t j = ns64 + lane; j < nobs; j += 64) bad = bad || obstacle_hit<DIM>(p, DIM, q_new, j);
        return __ballot(bad) == 0;
    }
    double mid[DIM];
    lerp<DIM>(q_near, q_new, 0.5, mid, DIM);
    const uint64_t cand = __ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid));
    const double dn = (double)nsteps;
    for (uint64_t base = 0; base < nsteps; base += 64) {
        const uint64_t s = base + lane + 1;
        const bool act = s <= nsteps;
        const double t = (double)(uint32_t)(act ? s : 1u) / dn;
        double x[DIM];
        lerp<DIM>(q_near, q_new, t, x, DIM);
        for (uint64_t m = cand; m != 0; m &= m - 1) {
            const int j = __ffsll((unsigned long long)m) - 1;
            double c[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) c[k] = readlane_f64(oc[k], j);
            // (read across lanes OUTSIDE the short-circuit below: sphere j's own lane is usually not one of the lanes with a
            // step to test, and a cross-lane read under their narrower exec mask returns whatever the register allocator left
            // in lane j -- with `othr` spilled, as in the R^4..R^6 lane-per-query kernels, a stale temporary)
            const double thr_j = readlane_f64(othr, j);
            bad = bad || (act && !(dist2<DIM>(c, x, DIM) > thr_j));
        }
        for (uint32_t j = ns64; j < nobs; ++j) bad = bad || (act && obstacle_hit<DIM>(p, DIM, x, j));
        if (__ballot(bad) != 0) break;
    }
    return __ballot(bad) == 0;
}

// Conservative midpoint filter (exactness: DESIGN.md section 3): every interpolated state lies within
// max_distance/2 of the segment midpoint, so d2(centre, mid) > (r + max_distance/2 + margin)^2 proves
// a sphere cannot be hit.  The filter never decides a motion invalid; it only skips provably valid work.
template <int DIM>
__device__ __forceinline__ bool sphere_maybe_hit(const double c[DIM], double filt, const double mid[DIM]) {
    return !(dist2<DIM>(c, mid, DIM) > filt);
}

// min over each 16-lane DPP row (every lane ends up with its row's minimum)
__device__ __forceinline__ double row_min_f64(double v) {
    v = dpp_min_step<0xB1, 0xf>(v);
    v = dpp_min_step<0x4E, 0xf>(v);
    v = dpp_min_step<0x141, 0xf>(v);
    v = dpp_min_step<0x140, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {
    v = dpp_umin_step<0xB1, 0xf>(v);
    v = dpp_umin_step<0x4E, 0xf>(v);
    v = dpp_umin_step<0x141, 0xf>(v);
    v = dpp_umin_step<0x140, 0xf>(v);
    return v;
}

// Lane-parallel sampling of m <= 64 consecutive queries (rrt.rs:177-184 + rvss.rs:233-249): lane l
// produces query js + l.  A query starts where the earlier ones stopped drawing: a goal sample
// takes 1 word, a uniform sample 1 + DIM.  So the word offset of lane l is
// (1+DIM)*l - DIM*popcount(goal lanes below l): the lanes iterate "read my Bernoulli word at the
// offset implied by the current goal mask -> ballot the new goal mask" to its fix-point (one extra
// round per goal sample in the batch).  A rejected range draw (res >= hi, probability ~2^-53) or a
// read past the LDS word window makes the function return false with nothing written; the caller
// then samples that batch sequentially.
template <int DIM, int RING = kRing, class QS = QSlot<DIM>>
__device__ __forceinline__ bool sample_batch(RngWindow& rng, const DevParams& p, const double* goal_c, uint32_t m,
                                             uint32_t lane, QS* qring, uint32_t js) {
    const uint64_t win_lo = rng.base_blk * 8;
    const uint64_t pos0 = rng.pos;
    if (pos0 < win_lo || pos0 + (uint64_t)m * (1 + DIM) > win_lo + 512) return false;
    const uint32_t rel0 = (uint32_t)(pos0 - win_lo);  // first word of the batch inside the window
    const bool act = lane < m;
    const bool always_goal = p.p_int == ~0ull;
    auto word = [&](uint32_t rel) -> uint64_t {       // rel < 512 by the check above
        const uint32_t a = rel0 + rel, bl = a >> 3, w = (a & 7u) * 2u;
        return ((uint64_t)rng.buf[w + 1][bl] << 32) | rng.buf[w][bl];
    };
    uint64_t goal_mask = always_goal ? ~0ull : 0ull;
    uint32_t off = 0;
    if (!always_goal) {
        const uint64_t below = (1ull << lane) - 1ull;
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
        QS& qs = qring[(js + lane) & (RING - 1)];
#pragma unroll
        for (int k = 0; k < DIM; ++k) qs.q[k] = q[k];
        qs.pos_after = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}

// scan groups: kGroup slots per uniform branch, never straddling the common / heavy-only boundary
template <int S, int C = (S == 21 ? 17 : S)>
__host__ __device__ constexpr int group_len(int g0) {
    const int common = C;
    int len = kGroup;
    if (g0 < common && g0 + len > common) len = common - g0;
    if (g0 + len > S) len = S - g0;
    return len;
}

}  // namespace oxhip
