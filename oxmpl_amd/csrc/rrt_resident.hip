// rrt_resident.hip -- register-resident RRT grow kernel for gfx950 (the fast path).
//
// A 10,240-node R^3 tree is 240 KB: too big for LDS (160 KB) but it fits the 512 KB vector
// register file of one CU.  One 576-thread workgroup per problem: 8 scanner waves hold the tree in
// VGPRs (126 VGPRs for R^3) and stream queries from an LDS ring, so the O(n) nearest-neighbour scan
// of rrt.rs:187-196 reads no memory at all and is bounded by f64 VALU issue; one resolver wave
// samples ahead, consumes the scanners' candidates in order, steers, checks the motion and commits.
// No workgroup barrier in steady state (see "Asynchronous pipeline" below).  The SoA tree in HBM
// is only the persistent copy: written once per insert, read once per launch.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

// ---- wave64 min of an f64 with DPP (VALU only, no LDS crossbar); result is wave-uniform
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min_step(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl = false + old = own value: lanes without a source keep their own value
    int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    // plain v_min_f64: the operands are squared distances (never NaN), so the canonicalising
    // v_max_f64 x,x that fmin() would add in front of every step is dead weight
    double o = __hiloint2double(ohi, olo), r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));
    return r;
}

__device__ __forceinline__ double wave_min_f64(double v) {
    v = dpp_min_step<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_min_step<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_min_step<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_min_step<0x140, 0xf>(v);  // row_mirror          -> every lane holds its row's min
    v = dpp_min_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_min_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave min
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

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
template <int S>
struct Layout {
    static constexpr bool kUneven = (S == 21);
    static constexpr uint32_t kCommon = kUneven ? 17u : (uint32_t)S;   // rows every wave holds
    static constexpr uint32_t kHeavyThreads = 384;
    static constexpr uint32_t kCapacity = kUneven ? (17u * 512u + 4u * 384u) : (uint32_t)S * 512u;
    __device__ static __forceinline__ bool heavy(uint32_t wave) { return !kUneven || (wave & 3u) != 0; }
    __device__ static __forceinline__ uint32_t node_index(uint32_t wave, uint32_t lane, uint32_t slot) {
        if (slot < kCommon) return slot * 512u + wave * 64u + lane;
        const uint32_t hw = wave - 1u - (wave > 4u ? 1u : 0u);  // waves 1,2,3,5,6,7 -> 0..5
        return kCommon * 512u + (slot - kCommon) * kHeavyThreads + hw * 64u + lane;
    }
    __device__ static __forceinline__ void locate(uint32_t i, uint32_t& thread, uint32_t& slot) {
        if (i < kCommon * 512u) { thread = i & 511u; slot = i >> 9; return; }
        const uint32_t r = i - kCommon * 512u, c = r % kHeavyThreads, hw = c >> 6;
        slot = kCommon + r / kHeavyThreads;
        thread = (hw + 1u + (hw >= 3u ? 1u : 0u)) * 64u + (c & 63u);
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

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin_step(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_umin_step<0xB1, 0xf>(v);
    v = dpp_umin_step<0x4E, 0xf>(v);
    v = dpp_umin_step<0x141, 0xf>(v);
    v = dpp_umin_step<0x140, 0xf>(v);
    v = dpp_umin_step<0x142, 0xa>(v);
    v = dpp_umin_step<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
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
template <int DIM>
__device__ __forceinline__ bool sample_batch(RngWindow& rng, const DevParams& p, const double* goal_c, uint32_t m,
                                             uint32_t lane, QSlot<DIM>* qring, uint32_t js) {
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
        QSlot<DIM>& qs = qring[(js + lane) & (kRing - 1)];
#pragma unroll
        for (int k = 0; k < DIM; ++k) qs.q[k] = q[k];
        qs.pos_after = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}

// scan groups: kGroup slots per uniform branch, never straddling the common / heavy-only boundary
template <int S>
__host__ __device__ constexpr int group_len(int g0) {
    const int common = (int)Layout<S>::kCommon;
    int len = kGroup;
    if (g0 < common && g0 + len > common) len = common - g0;
    if (g0 + len > S) len = S - g0;
    return len;
}

template <int DIM, int S, bool STAMP>
__global__ __launch_bounds__(kPipeThreads) void rrt_resident_kernel(DevParams p) {
    constexpr int D = DIM;
    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;

    __shared__ PipeShared<DIM> sh;

    const ProblemState st0 = p.state[prob];
    if (p.stop_at_goal && st0.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    uint8_t* skip = p.skip + (size_t)prob * cap;
    const uint32_t budget = (uint32_t)p.budget;  // the host keeps a launch's budget below 2^31

    if (tid < kRing) sh.done[tid] = 0;
    if (tid == 0) {
        sh.sampled = 0;
        sh.resolved = 0;
        sh.committed = st0.n_nodes;
        sh.stop_flag = 0;
    }
    __syncthreads();  // the only workgroup barrier of the launch

    if (wave < kScanWaves) {
        // ================================================================= scanner waves
        using Lay = Layout<S>;
        uint32_t n_local = st0.n_nodes;
        double tr[DIM][S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t i = ((uint32_t)s < Lay::kCommon || Lay::heavy(wave)) ? Lay::node_index(wave, lane, (uint32_t)s) : kNoNode;
            const bool live = i < n_local && skip[i] == 0;  // duplicates of a lower-index node never win: hold +inf
#pragma unroll
            for (int k = 0; k < DIM; ++k) tr[k][s] = live ? tree[(size_t)k * cap + i] : __builtin_inf();
        }
        uint64_t t_wait = 0, t_work = 0, t_pre = 0, t_scan = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;
        uint32_t seen_sampled = 0;
        for (uint32_t j = 0; j < budget; j += kBatch) {
            // one pass scans kBatch queries (the tail pass may hold one): fixed costs are shared, the two
            // reductions are independent dependency chains
            const uint32_t nb = (budget - j < (uint32_t)kBatch) ? (budget - j) : (uint32_t)kBatch;
            const uint32_t need = j + nb;
            // wait until the pass's queries are sampled (implies their ring slots were consumed kRing queries ago)
            for (uint32_t spins = 0; seen_sampled < need; ++spins) {
                if (lds_peek(&sh.stop_flag) != 0 || spins > kMaxSpins) break;  // every spin is bounded
                seen_sampled = uni(lds_peek(&sh.sampled));
                if (seen_sampled < need) __builtin_amdgcn_s_sleep(2);
            }
            if (seen_sampled < need) break;  // stop requested
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }
            // absorb the nodes committed since this wave's last snapshot (the owner lane takes each)
            const uint32_t nc = uni(lds_peek(&sh.committed));
            for (uint32_t i = n_local; i < nc; ++i) {
                uint32_t owner_thread, sl;
                Lay::locate(i, owner_thread, sl);
                const bool mine = tid == owner_thread;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (sl == (uint32_t)s) {
                        if (mine) {
#pragma unroll
                            for (int k = 0; k < D; ++k) tr[k][s] = sh.newn[i & 63][k];
                        }
                    }
                }
            }
            n_local = nc;
            double q[kBatch][D];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const uint32_t slot = (j + ((uint32_t)b < nb ? (uint32_t)b : 0u)) & (kRing - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) q[b][k] = unid(sh.qring[slot].q[k]);
            }
            // nearest neighbour over this wave's nodes (rrt.rs:187-196), d2 compare
            const uint32_t nslots = Lay::slots_in_use(wave, nc);
            Scan sc[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) sc[b] = Scan{__builtin_inf(), 0u, 0xFFFFFFFFu};
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_pre += now - t_mark; t_mark = now; }
            // slots are visited in groups of kGroup under ONE uniform branch: inside a group the code is
            // straight-line, so the scheduler interleaves kGroup x kBatch independent sub/mul/add chains
            // (empty slots hold +inf and can never win)
#pragma unroll
            for (int g0 = 0; g0 < S; g0 += group_len<S>(g0)) {
                if ((uint32_t)g0 < nslots) {
#pragma unroll
                    for (int s = g0; s < g0 + group_len<S>(g0); ++s) {
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = tr[k][s];
#pragma unroll
                        for (int b = 0; b < kBatch; ++b) scan_push(sc[b], dist2<D>(c, q[b], DIM), (uint32_t)s);
                    }
                }
            }
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_scan += now - t_mark; t_mark = now; }
            // reduce: branch-free for the whole batch, so the kBatch DPP chains / ballots interleave
            double wmin[kBatch];
            int wl[kBatch];
            uint32_t wslot[kBatch], wamb[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) wmin[b] = wave_min_f64(sc[b].b1);
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const uint64_t eqm = __ballot(sc[b].b1 == wmin[b]);
                wl[b] = eqm ? (__ffsll((unsigned long long)eqm) - 1) : 0;
                wslot[b] = __builtin_amdgcn_readlane(sc[b].slot, wl[b]);
                const uint32_t hb = hi32(wmin[b]) + 1;
                const bool amb_l = ((int)lane != wl[b] && hi32(sc[b].b1) <= hb) || (sc[b].h2 <= hb);
                wamb[b] = __ballot(amb_l) != 0 ? 1u : 0u;
            }
            // publish: the owning lane stores the coordinates, lane 0 the 16-byte head, then the slot is counted
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                if ((uint32_t)b < nb) {
                    const uint32_t slot = (j + (uint32_t)b) & (kRing - 1);
                    WavePub<DIM>& out = sh.pub[slot][wave];
                    store_slot<DIM, S>(tr, wslot[b], (int)lane == wl[b], out.c);
                    if (lane == 0) {
                        out.b1 = wmin[b];
                        out.i1 = Lay::node_index(wave, (uint32_t)wl[b], wslot[b]);
                        out.amb_nc = (nc << 1) | wamb[b];
                        lds_bump(&sh.done[slot]);
                    }
                }
            }
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) {
            p.dbg[16 + wave] = t_wait;
            p.dbg[24 + wave] = t_work + t_pre + t_scan;
            if (wave == 5) { p.dbg[8] = t_pre; p.dbg[9] = t_scan; p.dbg[10] = t_work; }
        }
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
    // this lane's obstacle (unused lanes hold a sphere that can never be hit), also mirrored to LDS
    double oc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) oc[k] = lane < ns64 ? p.sph_c[(size_t)k * p.n_spheres + lane] : 0.0;
    const double othr = lane < ns64 ? p.sph_thr[lane] : -1.0;
    const double ofilt = lane < ns64 ? p.sph_filt[lane] : -1.0;
#pragma unroll
    for (int k = 0; k < D; ++k) sh.obs[k][lane] = oc[k];
    sh.obs[D][lane] = ofilt;

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint64_t draws_done = st.draws;
    uint32_t n = st.n_nodes;
    uint32_t js = 0, jr = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    const uint32_t row = lane >> 4, sub = lane & 15;
    uint64_t t_wait = 0, t_work = 0, t_samp = 0, t_comb = 0, n_amb = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;

    // ---- one query, the reference's sequential semantics in full (tails, batch conflicts, near-ties):
    //      candidates = the 8 scanner waves' + every node committed after the oldest scan snapshot
    auto resolve_one = [&](uint32_t jq, uint32_t& nearest, double (&q_new)[D], bool& dup) -> bool {
        const uint32_t slot = jq & (kRing - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = unid(sh.qring[slot].q[k]);
        const bool inS = lane < (uint32_t)kScanWaves;
        const WavePub<DIM>& mine = sh.pub[slot][inS ? lane : 0];
        const double pb = inS ? mine.b1 : __builtin_inf();
        const uint32_t pan = inS ? mine.amb_nc : 0xFFFFFFFFu;
        const uint32_t pamb = inS ? (pan & 1u) : 0u;
        const uint32_t pidxS = inS ? mine.i1 : kNoNode;
        const uint32_t base_min = wave_min_u32(pan >> 1);  // oldest snapshot among the 8 scans
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
        const bool nearS = inS && hi32(pb) <= hb;
        const bool nearP = pv && hi32(d2p) <= hb;
        const uint64_t mS = __ballot(nearS), mP = __ballot(nearP);
        const bool from_scan = mS != 0;
        const int wl = from_scan ? (__ffsll((unsigned long long)mS) - 1) : (mP ? (__ffsll((unsigned long long)mP) - 1) : 0);
        nearest = from_scan ? (uint32_t)__builtin_amdgcn_readlane((int)pidxS, wl)
                            : (uint32_t)__builtin_amdgcn_readlane((int)pidx, wl);
        // unambiguous iff every near candidate is that one node and no near wave saw a second near node
        const bool amb = __ballot((nearS && (pidxS != nearest || pamb != 0)) || (nearP && pidx != nearest)) != 0;
        double q_near[D];
        double dist_or_g;
        if (!amb) {
#pragma unroll
            for (int k = 0; k < D; ++k) q_near[k] = from_scan ? unid(sh.pub[slot][wl].c[k]) : unid(sh.newn[wl][k]);
            dist_or_g = g;
            dup = g == 0.0;
        } else {
            // rare (~1e-6 of queries): the reference's own loop -- post-sqrt compare with lowest-index ties --
            // over the persistent copy of the tree in global memory
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
        return motion_full<DIM>(p, lane, q_near, q_new, oc, othr, ns64);
    };

    while (jr < budget) {
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }
        // sample ahead (rrt.rs:177-184); a query may only reuse a ring slot after its previous tenant was resolved
        if (js < budget && js - jr <= (uint32_t)(kRing / 2)) {
            uint32_t m = jr + kRing - js;  // free ring slots
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
            if (!sample_batch<DIM>(rng, p, goal_c, m, lane, sh.qring, js)) {
                for (uint32_t b = 0; b < m; ++b) {  // (never expected) a redraw ran past the window: one by one
                    double qn[D];
                    sample_state<D, false>(rng, p, DIM, goal_c, qn);
                    QSlot<DIM>& qs = sh.qring[(js + b) & (kRing - 1)];
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) qs.q[k] = qn[k];
                        qs.pos_after = rng.pos;
                    }
                }
            }
            js += m;
            if (lane == 0) lds_post(&sh.sampled, js);
        }
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_samp += now - t_mark; t_mark = now; }

        // the scanners publish kBatch queries per pass: resolve them as one batch
        const uint32_t nbq = (budget - jr < (uint32_t)kBatch) ? (budget - jr) : (uint32_t)kBatch;
        uint32_t spins = 0;
        for (uint32_t b = 0; b < nbq; ++b) {
            while (uni(lds_peek(&sh.done[(jr + b) & (kRing - 1)])) < (uint32_t)kScanWaves && spins <= kMaxSpins) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
            }
        }
        if (spins > kMaxSpins) { stop = 4; break; }  // OXHIP_STOP_INTERNAL: a scanner never published (bug guard)
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }

        // ---- row-parallel phase: DPP row r (16 lanes) works on query jr + r against the tree of n0 nodes
        const uint32_t n0 = n;
        const bool active = row < nbq;
        const uint32_t slot_r = (jr + (active ? row : 0u)) & (kRing - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = sh.qring[slot_r].q[k];
        const uint64_t pos_after_r = sh.qring[slot_r].pos_after;
        const bool inS = active && sub < (uint32_t)kScanWaves;
        const WavePub<DIM>& mine = sh.pub[slot_r][inS ? sub : 0];
        const double pb = inS ? mine.b1 : __builtin_inf();
        const uint32_t pan = inS ? mine.amb_nc : 0xFFFFFFFFu;
        const uint32_t pidxS = inS ? mine.i1 : kNoNode;
        const uint32_t bmin_r = row_min_u32(pan >> 1);  // oldest scan snapshot for this row's query
        if (__ballot(active && (n0 - bmin_r > 64u || bmin_r > n0)) != 0) { stop = 4; break; }  // ring would have wrapped (bug guard)
        // nodes committed after that snapshot (at most a few): lanes of the row stride over them
        Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
        if (active) {
            for (uint32_t i = bmin_r + sub; i < n0; i += 16) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.newn[i & 63][k];
                scan_push(pd, dist2<D>(c, q, DIM), i);  // ascending i: ties keep the lower index
            }
        }
        const double g_r = row_min_f64(pd.b1 < pb ? pd.b1 : pb);
        const uint32_t hb = hi32(g_r) + 1;
        const bool nearS = inS && hi32(pb) <= hb;
        const bool nearP = active && pd.slot != kNoNode && hi32(pd.b1) <= hb;
        const uint32_t rowS = (uint32_t)(__ballot(nearS) >> (16 * row)) & 0xFFFFu;
        const uint32_t rowP = (uint32_t)(__ballot(nearP) >> (16 * row)) & 0xFFFFu;
        const bool from_scan = rowS != 0;
        const uint32_t wsub = from_scan ? (uint32_t)(__ffs((int)rowS) - 1) : (rowP ? (uint32_t)(__ffs((int)rowP) - 1) : 0u);
        const int src_lane = (int)(16 * row + wsub);
        const uint32_t wS = (uint32_t)__shfl((int)pidxS, src_lane, 64), wP = (uint32_t)__shfl((int)pd.slot, src_lane, 64);
        const uint32_t nearest_r = from_scan ? wS : wP;
        // ambiguous iff a second node is near: another near candidate, a near wave that saw a second near node,
        // or a lane whose second-best recent node is near too
        const bool amb_l = (nearS && (pidxS != nearest_r || (pan & 1u) != 0)) || (nearP && pd.slot != nearest_r) ||
                           (active && pd.h2 <= hb);
        const bool amb_r = ((uint32_t)(__ballot(amb_l) >> (16 * row)) & 0xFFFFu) != 0 || (rowS == 0 && rowP == 0);
        double q_near[D], qn[D], mid[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q_near[k] = from_scan ? sh.pub[slot_r][wsub].c[k] : sh.newn[nearest_r & 63][k];
        steer<DIM>(p, false, g_r, q_near, q, qn);
        lerp<DIM>(q_near, qn, 0.5, mid, DIM);
        bool maybe_l = false;
        if (nobs > 0) {
            for (uint32_t o = sub; o < 64; o += 16) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                maybe_l = maybe_l || sphere_maybe_hit<DIM>(c, sh.obs[D][o], mid);
            }
        }
        const bool maybe_r = extras || (((uint32_t)(__ballot(maybe_l) >> (16 * row)) & 0xFFFFu) != 0);
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_comb += now - t_mark; t_mark = now; }

        // ---- sequential phase: commit in query order; a node committed earlier in this batch that is
        //      closer (or near-tied) to a later query forces that query through resolve_one
        double cn[kBatch][D];   // coordinates the scanners will hold for the nodes committed in this batch
        bool cn_valid[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) cn_valid[b] = false;
        bool leave = false;
        uint32_t processed = 0;
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            if (!leave && (uint32_t)b < nbq) {
                if (!p.freeze && n >= p.max_nodes) { stop = 2; leave = true; }
            }
            if (!leave && (uint32_t)b < nbq) {
                const int l0 = 16 * b;
                const uint32_t slot = (jr + (uint32_t)b) & (kRing - 1);
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
                        ok = motion_full<DIM>(p, lane, qnr, q_new, oc, othr, ns64);
                    }
                }
                // bookkeeping (wave-uniform, on the scalar unit where the compiler can)
                uint64_t h = fnv_mix(st.checksum, (uint64_t)nearest);
#pragma unroll
                for (int k = 0; k < D; ++k) h = fnv_mix(h, uni64((uint64_t)__double_as_longlong(q_new[k])));
                st.checksum = fnv_mix(h, ok ? 1ull : 0ull);
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
                if (lane == 0) {
                    lds_post(&sh.done[slot], 0);                 // free the slot ...
                    lds_post(&sh.resolved, jr + (uint32_t)b + 1); // ... before the sampler may hand it out again
                }
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
            p.dbg[0] = t_samp; p.dbg[1] = t_wait; p.dbg[2] = t_work; p.dbg[3] = t_comb; p.dbg[4] = n_amb; p.dbg[7] = st.iterations;
        }
    }
}

// instantiations: (dim, slots) -> capacity 512 * slots nodes
static int pick_slots(uint32_t cap) {
    const uint32_t need = (cap + kScanThreads - 1) / kScanThreads;
    if (need <= 4) return 4;
    if (cap <= Layout<21>::kCapacity) return 21;
    return 0;
}

bool resident_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pick_slots(cap) != 0; }

void launch_rrt_resident(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kPipeThreads);
    const int s = pick_slots(p.cap);
#define OXHIP_LAUNCH(DIM_, S_)                                                                              \
    do {                                                                                                    \
        if (p.dbg) hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, true>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, false>), grid, block, 0, stream, p);         \
    } while (0)
    if (p.dim == 3) {
        if (s == 4) OXHIP_LAUNCH(3, 4); else OXHIP_LAUNCH(3, 21);
    } else {
        if (s == 4) OXHIP_LAUNCH(2, 4); else OXHIP_LAUNCH(2, 21);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
