// rrt_lanes.hip -- the register-resident RRT grow kernel: binary32 dot-product SCREEN + lane-per-query resolver.
//
// The tree of one planning problem lives in the vector registers of eight scanner waves, as in rrt_resident.hip, but
// both ends of the pipeline are rebuilt around what the chip actually issues (tools/valu_mix_bench.hip: every VALU
// instruction of the old screen -- packed f32, and_or, med3, min -- costs one ~1.9 ns issue slot per wave64):
//
// * Scanners: 2.5 instructions per (64-node register row, query) instead of 6.  A node is held as a = fl32(x - c0) (c0 =
//   the centre of the bounds) plus cc = fl32(|a|^2), a query as Q = -2 fl32(q - c0), and the screen value is
//       s' = cc + a . Q  =  |a - b|^2 - |b|^2          (b = fl32(q - c0): D packed fused multiply-adds for two queries),
//   which orders the nodes like their squared distance.  The only bookkeeping per (row, query) is one v_min_f32: a lane
//   keeps its smallest s', a wave publishes (smallest s', the lane that holds it, the smallest s' of its OTHER lanes).
// * Resolver: lane j IS query jr + j (up to 64 iterations resolved side by side; one sqrt / division / filter pass serves
//   64 queries).  A lane takes the winning scanner lane's own <= 24 nodes -- four consecutive indices per register-row
//   block, so 32-byte loads from the binary64 tree -- and the nodes committed after the scans' snapshots (LDS ring), computes
//   their distances in binary64 exactly as the reference does, and keeps the nearest (lowest index on ties, near-ties
//   flagged).  It ACCEPTS that node only if every node it did not look at is provably farther (error model below, E = u H^2 D (3D + 9)):
//       every other scanner lane's smallest s' exceeds K1 + 2.25 E  (K1 = the winning lane's; within the winning wave the scanners
//                                                                   test this themselves, against K1 + 2.5 E, with one ballot), and
//       g <= K1 + |b|^2 + E                                        (the best node the lane looked at is the one the screen saw);
//   then a node it did not look at has d^2 >= s' + |b|^2 - E > K1 + |b|^2 + 1.25 E >= g + E / 4.  Otherwise -- ~1e-3 of queries -- the query is resolved alone by the reference's literal loop over
//   the binary64 tree.  Everything that enters a result (distance, steer, motion check, tree, checksum) is binary64.
//
// Sequential semantics (iteration k sees exactly the tree left by iterations < k, rrt.rs:170-225) are kept by committing,
// per round, the longest prefix of the 64 lanes whose results cannot have been changed by the nodes the lanes before them
// insert (a new node changes a later query's result only if it is at least as close as that query's nearest node: one d2
// per pair, lane-parallel), and re-resolving the rest in the next round against the grown tree.  With inserts suppressed
// (the steady measurement) every round commits all its lanes; while a tree is small a round commits ~sqrt(2n) lanes.
//
// Error model (u = 2^-24; H = largest |coordinate - c0| among the bounds, the goal centre and the tree; R^2 <= D H^2):
//   a = fl32(x - c0), b = fl32(q - c0):  |(a - b) - (x - q)|_k <= 2 u' H,  so  | |a - b| - d | <= 2 u' sqrt(D) H  and, with
//   |a - b| <= 2 sqrt(D) H,   d^2 >= |a - b|^2 - 8 u' D H^2;
//   cc = fl32(|a|^2) and the D fused multiply-adds round values of magnitude <= 3 R^2:  |s' + |b|^2 - |a - b|^2| <= (3D + 1) u R^2.
//   Hence every node with screen value s' has  d^2 >= s' + |b|^2 - E,  E = u H^2 D (3D + 9); the kernel uses 2E.
//
// Waves: 8 scanners (six hold S register rows, the two that share a SIMD with the resolver C), 1 resolver, 1 sampler (draws
// the queries ahead: they depend on the RNG stream only).  LDS rings: 128 queries in flight, the last 256 committed nodes.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"
#include "lanes_reduce.hpp"
#include "lane_query_common.hpp"

namespace oxhip {

constexpr int kLanesThreads = kScanThreads + 128;   // + the resolver wave + the sampler wave
#ifndef OXHIP_LANES_QRING
#define OXHIP_LANES_QRING 128
#endif
constexpr int kQRing = OXHIP_LANES_QRING;           // queries in flight (power of two)
constexpr int kNRing = 2 * kQRing;                  // committed nodes kept in LDS (power of two)
// The resolver stages up to 64 would-be nodes in the slots n .. n + 63, i.e. over the nodes n - kNRing ..; a scanner may still
// have to absorb nodes as old as n - (kQRing + one pass of 8 queries): the ring must hold both.
static_assert((kQRing & (kQRing - 1)) == 0 && kNRing >= kQRing + 64 + 8, "ring sizes");
#ifndef OXHIP_LANES_PASS3
#define OXHIP_LANES_PASS3 8
#endif
template <int DIM> struct LanesPass { static constexpr int Q = DIM >= 6 ? 4 : (DIM <= 3 ? OXHIP_LANES_PASS3 : 8); };   // queries one scanner pass covers (register budget)
#ifndef OXHIP_LANES_PAIR
#define OXHIP_LANES_PAIR 1            // 0: (diagnosis) two-lane cases go to the whole-tree path like three-way ones
#endif
#ifndef OXHIP_LANES_TRANSPOSE_MAX
#define OXHIP_LANES_TRANSPOSE_MAX 32  // rounds of at most this many lanes use the transposed sphere pre-filter (0: never)
#endif
#ifndef OXHIP_DEPTH_GROW
#define OXHIP_DEPTH_GROW 128   // (64: 402 M it/s growing configs[1]; 128: 437-441 M; a 256-query ring with 192 / 256: 440 / 427 M)
#endif
constexpr uint32_t kDepthGrow = OXHIP_DEPTH_GROW;                 // queries sampled ahead of the resolver while inserts are on
#ifndef OXHIP_WINDOW_DIV
#define OXHIP_WINDOW_DIV 8   // while a tree is small the query window is (tree size / this), at least two passes
#endif
#ifndef OXHIP_LANES_PRIO
#define OXHIP_LANES_PRIO 16
#endif
#ifndef OXHIP_WANT_DIV_GROW
#define OXHIP_WANT_DIV_GROW 8   // while inserts are on the resolver starts a round when 1/this of the window is published
#endif
#ifndef OXHIP_LANES_FOLD
#define OXHIP_LANES_FOLD 8
#endif
#ifndef OXHIP_LANES_TRIP
#define OXHIP_LANES_TRIP(DIM) 2   // (R^5 / R^6 with one block per trip -- 32 registers less in the resolver, which spills there -- was measured: see DESIGN.md 5.5)
#endif
#ifndef OXHIP_LANES_QSGPR
#define OXHIP_LANES_QSGPR 0   // 1: the pass's queries in scalar registers (24 v_readfirstlane per pass); 0: in vector register pairs
#endif

struct alignas(16) LanePub {   // one wave's screen result for one query (one 16-byte LDS record)
    uint32_t k1;   // bits of the smallest screen value s' of the wave
    uint32_t k2;   // (unused)
    uint32_t th;   // scanner thread (wave * 64 + lane) that holds k1; bit 31: another lane of the wave is within 2.5 E of k1
    uint32_t nc;   // tree size this scan covered (the wave's snapshot of `committed`)
};

// Node -> (scanner thread, register row) in blocks of four rows: a thread's four rows of a block are four CONSECUTIVE node
// indices, so the resolver fetches a scanner lane's candidates with 32-byte loads.  Blocks 0 .. C/4-1 span all 512 scanner
// threads (2048 nodes each); blocks C/4 .. S/4-1 only the 384 threads of waves 1,2,3,5,6,7 (1536 nodes each): waves 0 and
// 4 share their SIMD with the resolver and hold fewer rows.
template <int S, int C>
struct Layout4 {
    static_assert(S % 4 == 0 && C % 4 == 0 && C <= S, "rows come in blocks of four");
    static constexpr uint32_t kCommonNodes = (uint32_t)C * 512u;
    static constexpr uint32_t kCapacity = kCommonNodes + (uint32_t)(S - C) * 384u;
    __device__ static __forceinline__ bool heavy(uint32_t wave) { return (wave & 3u) != 0; }
    __device__ static __forceinline__ uint32_t rows(uint32_t wave) { return heavy(wave) ? (uint32_t)S : (uint32_t)C; }
    // The six heavy waves take a block's 1536 nodes in the order 3, 2, 7, 6, 1, 5: a partly filled last block costs its
    // holders four more rows each, and waves 1 and 5 share their SIMD with the sampler wave (ChaCha12 + the draws: ~10
    // instructions per query), so they come last; 3 / 2 first puts the first two partial holders on different SIMDs.
    __device__ static __forceinline__ uint32_t hw_of_wave(uint32_t wave) { return (0x23500140u >> (4u * wave)) & 7u; }   // waves 1,2,3,5,6,7 -> 4,1,0,5,3,2
    __device__ static __forceinline__ uint32_t wave_of_hw(uint32_t hw) { return (0x516723u >> (4u * hw)) & 7u; }        // 0..5 -> 3,2,7,6,1,5
    // first node of block `blk` of scanner thread `th` (kNoNode when that wave does not hold the block)
    __device__ static __forceinline__ uint32_t block_base(uint32_t th, uint32_t blk) {
        if (blk < (uint32_t)(C / 4)) return blk * 2048u + th * 4u;
        const uint32_t wave = th >> 6;
        if (!heavy(wave)) return kNoNode;
        return kCommonNodes + (blk - (uint32_t)(C / 4)) * 1536u + (hw_of_wave(wave) * 64u + (th & 63u)) * 4u;
    }
    __device__ static __forceinline__ void locate(uint32_t i, uint32_t& thread, uint32_t& row) {
        if (i < kCommonNodes) { thread = (i & 2047u) >> 2; row = (i >> 11) * 4u + (i & 3u); return; }
        const uint32_t r = i - kCommonNodes, blk = r / 1536u, c = r % 1536u, ht = c >> 2, hw = ht >> 6;
        row = (uint32_t)C + blk * 4u + (c & 3u);
        thread = wave_of_hw(hw) * 64u + (ht & 63u);
    }
    // rows of `wave` that may hold a node when the tree has n nodes (a multiple of four: whole blocks)
    __device__ static __forceinline__ uint32_t rows_in_use(uint32_t wave, uint32_t n) {
        if (n <= kCommonNodes) {
            const uint32_t full = n >> 11, rem = n & 2047u;
            return 4u * full + (rem > wave * 256u ? 4u : 0u);
        }
        if (!heavy(wave)) return (uint32_t)C;
        const uint32_t r = n - kCommonNodes, full = r / 1536u, rem = r % 1536u;
        return (uint32_t)C + 4u * full + (rem > hw_of_wave(wave) * 256u ? 4u : 0u);
    }
};

template <int DIM>
struct LanesShared {
    uint32_t rng_buf[16][64];
    double q[DIM][kQRing];               // the queries, coordinate-major: resolver lane j reads q[k][slot_j] conflict-free
    uint64_t pos_after[kQRing];          // stream position after each query's draws
    float qf[DIM][kQRing];               // Q = -2 fl32(q - c0), coordinate-major: a scanner reads a query PAIR's coordinate with one 8-byte uniform read
    // the scanners' results, field by field and wave-major: resolver lane j reads x[w][slot_j] conflict-free, and a scanner
    // wave's lane 63 -- where the DPP reduction leaves its result -- stores its values without moving them anywhere first
    uint32_t pub_k1[kScanWaves][kQRing];   // bits of the smallest screen value s' of the wave
    uint32_t pub_th[kScanWaves][kQRing];   // scanner thread (wave * 64 + lane) that holds k1; bit 31: another lane of the wave is within 2.5 E of k1
    uint32_t pub_nc[kScanWaves][kQRing / 4];   // per pass: tree size the scan covered (the wave's snapshot of `committed`)
    double newn[DIM][kNRing];            // the last kNRing committed nodes, node i at i & (kNRing - 1); +inf for skipped duplicates
    double obs[DIM + 2][64];             // first 64 spheres: centre, validity threshold, filter threshold
    float newn32[kNRing][DIM < 4 ? 4 : 8];   // the same nodes as the scanners hold them: a = fl32(x - c0), then cc = fl32(|a|^2) (+inf: skipped)
    float obs32[64][DIM < 4 ? 4 : 8];        // the spheres for the binary32 pre-filter: fl32(centre - c0), then fl32(|.|^2)
    float obs32_thr[64];                     // ... and the filter threshold + error bound, rounded up (-1: no sphere)
    float fq[64][DIM < 4 ? 4 : 8];           // staging of the pre-filter: lane j's midpoint as -2 fl32(mid - c0), then |.|^2 rounded down
    uint32_t wave_done[kScanWaves];      // queries each scanner wave has published (monotonic)
    uint32_t sampled, resolved, committed, stop_flag;
    uint32_t heartbeat;                  // bumped by the resolver while it works: waiters only give up when it stands still
    uint32_t mabs_bits;                  // bits of the largest |fl32(coordinate - c0)| the scanners loaded
};

// Lane-parallel sampling of m <= 64 consecutive queries into the coordinate-major ring: sample_batch of
// rrt_resident_common.hpp (rrt.rs:177-184 + rvss.rs:233-249) with this kernel's ring layout and the fl32 copies.
template <int DIM>
__device__ __forceinline__ bool sample_lanes(RngWindow& rng, const DevParams& p, const double* goal_c, double goal_radius, const double* c0, uint32_t m,
                                             uint32_t lane, LanesShared<DIM>& sh, uint32_t js) {
    const uint64_t win_lo = rng.base_blk * 8;
    const uint64_t pos0 = rng.pos;
    if (pos0 < win_lo || pos0 + (uint64_t)m * (1 + DIM) > win_lo + 512) return false;
    const uint32_t rel0 = (uint32_t)(pos0 - win_lo);
    const bool act = lane < m;
    const bool always_goal = p.p_int == ~0ull;
    const bool disc = DIM == 2 && p.goal_sampler == OXHIP_GOAL_SAMPLE_UNIFORM_DISC;
    const uint32_t gw = disc ? 2u : 0u;   // words a goal sample draws after its Bernoulli word
    auto word = [&](uint32_t rel) -> uint64_t {
        const uint32_t a = rel0 + rel, bl = a >> 3, w = (a & 7u) * 2u;
        return ((uint64_t)rng.buf[w + 1][bl] << 32) | rng.buf[w][bl];
    };
    uint64_t goal_mask = always_goal ? ~0ull : 0ull;
    uint32_t off = act ? gw * lane : 0u;   // (every query is a goal sample: gw words each)
    if (!always_goal) {
        const uint64_t below = below_mask(lane);
        for (uint32_t round = 0; round <= m; ++round) {
            off = act ? (1u + DIM) * lane - ((uint32_t)DIM - gw) * (uint32_t)__popcll(goal_mask & below) : 0u;
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
    redraw = redraw && !goal;
    if (DIM == 2 && disc) {   // the disc sampler's two words follow the Bernoulli word (if one was drawn)
        const uint32_t base = act && goal ? off + (always_goal ? 0u : 1u) : 0u;
        double gx, gy;
        const bool okd = goal_disc_sample(word(base), word(base + 1u), goal_c, goal_radius, gx, gy);
        if (goal) { q[0] = gx; q[DIM >= 2 ? 1 : 0] = gy; redraw = !okd; }
    }
    if (__ballot(act && redraw) != 0) return false;
    const uint32_t cnt = always_goal ? gw : (goal ? 1u + gw : 1u + (uint32_t)DIM);
    if (act) {
        const uint32_t slot = (js + lane) & (kQRing - 1);
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            sh.q[k][slot] = q[k];
            sh.qf[k][slot] = -2.0f * (float)(q[k] - c0[k]);
        }
        sh.pos_after[slot] = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}

template <int DIM, int S, int ROW>
__device__ __forceinline__ void absorb_one(lf32x2 (&tr)[DIM][S / 2], lf32x2 (&tcc)[S / 2], bool mine, const float (&f)[DIM], float fcc) {
    if (mine) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) tr[k][ROW / 2][ROW % 2] = f[k];
        tcc[ROW / 2][ROW % 2] = fcc;
    }
}
// binary descent on the wave-uniform row number: log2(S) scalar branches, then the one row's predicated writes
template <int DIM, int S, int LO, int HI>
__device__ __forceinline__ void absorb_range(lf32x2 (&tr)[DIM][S / 2], lf32x2 (&tcc)[S / 2], uint32_t row, bool mine, const float (&f)[DIM], float fcc) {
    if constexpr (HI - LO == 1) {
        absorb_one<DIM, S, LO>(tr, tcc, mine, f, fcc);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (row < (uint32_t)MID) absorb_range<DIM, S, LO, MID>(tr, tcc, row, mine, f, fcc);
        else absorb_range<DIM, S, MID, HI>(tr, tcc, row, mine, f, fcc);
    }
}
template <int DIM, int S>
__device__ __forceinline__ void absorb_row(lf32x2 (&tr)[DIM][S / 2], lf32x2 (&tcc)[S / 2], uint32_t row, bool mine, const float (&f)[DIM], float fcc) {
    absorb_range<DIM, S, 0, S>(tr, tcc, row, mine, f, fcc);
}

template <int DIM, int S, int C, bool STAMP>
__global__ __launch_bounds__(kLanesThreads) void rrt_lanes_kernel(DevParams p) {
    constexpr int D = DIM;
    constexpr int kPassQ = LanesPass<DIM>::Q;
    static_assert(DIM <= 8, "qf holds eight floats per query");
    using Lay = Layout4<S, C>;

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
    double c0[D];   // the centre of the bounds: the screen works on coordinates relative to it
#pragma unroll
    for (int k = 0; k < D; ++k) c0[k] = 0.5 * p.lo[k] + 0.5 * p.hi[k];

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
        // rows in pairs: a packed instruction's operand is a 64-bit register pair anyway, and op_sel picks the half that is
        // broadcast to both halves of the operation (S x (D + 1) VGPRs of tree instead of twice that)
        uint32_t n_local = st0.n_nodes;
        lf32x2 tr[DIM][S / 2], tcc[S / 2];
        uint32_t mab = 0;
        const uint32_t my_rows = Lay::rows(wave);
#pragma unroll
        for (int blk = 0; blk < S / 4; ++blk) {
            const uint32_t ib = Lay::block_base(tid, (uint32_t)blk);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int s = 4 * blk + t;
                const uint32_t i = ib == kNoNode ? kNoNode : ib + (uint32_t)t;
                const bool in_tree = i < n_local;
                const bool live = in_tree && skip[in_tree ? i : 0u] == 0;  // duplicates of a lower-index node never win: hold +inf
                double sq = 0.0;
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    const float f = in_tree ? (float)(tree[(size_t)k * cap + (in_tree ? i : 0u)] - c0[k]) : 0.0f;
                    const uint32_t ab = lf32_bits(f) & 0x7FFFFFFFu;
                    mab = ab > mab ? ab : mab;
                    tr[k][s / 2][s % 2] = live ? f : 0.0f;
                    sq += (double)f * (double)f;
                }
                tcc[s / 2][s % 2] = live ? (float)sq : __builtin_inff();
            }
        }
        __hip_atomic_fetch_max(&sh.mabs_bits, mab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();  // the resolver reads mabs_bits after this barrier (the second and last of the launch)
        // 2.5 E, E as the resolver computes it (same H): a second lane whose smallest s' is within it makes the wave's result "not
        // proven" (the resolver needs 2.25 E; the sum k1 + 2.5 E rounds by at most half an ulp of 3 D H^2 = E / 6 in R^3, less above)
        float e4f;
        {
            double h = (double)lbits_f32(lds_peek(&sh.mabs_bits)) * (1.0 + 0x1p-23);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                h = fmax(h, fmax(fabs(p.lo[k] - c0[k]), fabs(p.hi[k] - c0[k])));
                h = fmax(h, fabs(p.goal_c[(size_t)prob * DIM + k] - c0[k]));
            }
            h = unid(h) * 1.001;
            const bool usable = h < 1e15 && fabs(c0[0]) < 1e300;
            e4f = usable ? f32_up(2.5 * (0x1p-24 * h * h * (double)(D * (3 * D + 9)) * 1.0001 + 1e-290)) : __builtin_inff();
        }

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
                if ((owner_thread >> 6) != wave) continue;   // another wave's node: skip the row ladder below
                const bool mine = tid == owner_thread;
                float f[D];   // the resolver wrote the node as the scanners hold it: a = fl32(x - c0), cc = fl32(|a|^2) (+inf: skipped duplicate)
#pragma unroll
                for (int k = 0; k < D; ++k) f[k] = sh.newn32[i & (kNRing - 1)][k];
                const float fcc = sh.newn32[i & (kNRing - 1)][D];
                // (register arrays cannot be indexed dynamically: the wave-uniform row number selects one of S compile-time copies)
                absorb_row<DIM, S>(tr, tcc, uni(sl), mine, f, fcc);
            }
            n_local = nc;
            // the pass's queries Q = -2 fl32(q - c0), two per packed register
            // (a pass starts on a multiple of its width, so a pair's two slots are adjacent and 8-byte aligned; in the last, short
            // pass the slots beyond the budget hold older queries: screened and never published)
            lf32x2 q[kPassQ / 2][D];
#pragma unroll
            for (int bp = 0; bp < kPassQ / 2; ++bp) {
                const uint32_t slot = (j + 2u * (uint32_t)bp) & (kQRing - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const lf32x2 v = *reinterpret_cast<const lf32x2*>(&sh.qf[k][slot]);
#if OXHIP_LANES_QSGPR
                    q[bp][k][0] = lbits_f32(uni(lf32_bits(v[0])));   // wave-uniform: scalar registers
                    q[bp][k][1] = lbits_f32(uni(lf32_bits(v[1])));
#else
                    q[bp][k] = v;   // every lane holds the same pair: a vector register pair, no v_readfirstlane
#endif
                }
            }
            uint32_t nrows = Lay::rows_in_use(wave, nc);
            if (nrows > my_rows) nrows = my_rows;
            float b1[kPassQ];
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) b1[b] = __builtin_inff();
            // screen (rrt.rs:187-196 as s' = cc + a . Q in binary32): blocks of four rows under one uniform branch
#pragma unroll
            for (int g0 = 0; g0 < S; g0 += 4) {
                if ((uint32_t)g0 < nrows) {
                    if constexpr (DIM <= 5) {
#pragma unroll
                    for (int sp = g0 / 2; sp < g0 / 2 + 2; ++sp) {
                        // a packed fused multiply-add covers the two ROWS of a register pair for one query (op_sel broadcasts the
                        // query's half of its pair) and one v_min3_f32 folds both rows into the query's minimum; four queries'
                        // chains side by side, so consecutive instructions are independent
#pragma unroll
                        for (int b0 = 0; b0 < kPassQ; b0 += 4) {
                            lf32x2 acc[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[t] = tcc[sp];
#pragma unroll
                            for (int k = 0; k < D; ++k) {
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const lf32x2 qp = q[(b0 + t) / 2][k];
                                    const lf32x2 qq = ((b0 + t) & 1) ? __builtin_shufflevector(qp, qp, 1, 1) : __builtin_shufflevector(qp, qp, 0, 0);
                                    acc[t] = __builtin_elementwise_fma(tr[k][sp], qq, acc[t]);
                                }
                            }
#pragma unroll
                            for (int t = 0; t < 4; ++t) vmin3_f32(b1[b0 + t], acc[t][0], acc[t][1]);
                        }
                    }
                    } else {
                    // R^6 (four queries per pass, 20 x 7 registers of tree): the form that needs four accumulator registers
                    // instead of eight -- the row is the broadcast half, a packed instruction covers a query PAIR, one v_min_f32
                    // per (row, query) -- because here every register saved is a spill saved (190 M it/s against 127 M)
#pragma unroll
                    for (int s = g0; s < g0 + 4; ++s) {
                        lf32x2 acc[kPassQ / 2];
#pragma unroll
                        for (int bp = 0; bp < kPassQ / 2; ++bp)
                            acc[bp] = (s & 1) ? __builtin_shufflevector(tcc[s / 2], tcc[s / 2], 1, 1) : __builtin_shufflevector(tcc[s / 2], tcc[s / 2], 0, 0);
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const lf32x2 a = (s & 1) ? __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 1, 1) : __builtin_shufflevector(tr[k][s / 2], tr[k][s / 2], 0, 0);
#pragma unroll
                            for (int bp = 0; bp < kPassQ / 2; ++bp) acc[bp] = __builtin_elementwise_fma(a, q[bp][k], acc[bp]);
                        }
#pragma unroll
                        for (int bp = 0; bp < kPassQ / 2; ++bp) {
                            vmin_f32(b1[2 * bp], acc[bp][0]);
                            vmin_f32(b1[2 * bp + 1], acc[bp][1]);
                        }
                    }
                    }
                }
            }
            // reduce (lanes_reduce.hpp): all of the pass's wave minima in ONE register -- lanes 8b .. 8b + 7 hold query b's --
            // 17 instructions for eight queries instead of eight 6-step reductions
            const float u = lanes_min_transposed<kPassQ>(b1);
            const float uthr = u + e4f;   // (+inf stays +inf; never NaN: see above)
            // per query ONE ballot: which lanes' smallest s' are within 2.5 E of the wave's?  The lane that holds the minimum is
            // always among them.  Exactly one: that lane, proven.  Exactly two: both are named (bits 0..8, bits 16..21 + bit 30) and
            // the resolver looks at both lanes' nodes, in whichever order.  More: not proven (bit 31), the resolver's whole-tree path.
            uint64_t nearm[kPassQ];   // (all the ballots first: eight independent compare results instead of a chain through vcc)
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) {
                const float thr = lbits_f32((uint32_t)__builtin_amdgcn_readlane((int)lf32_bits(uthr), 8 * b));
                nearm[b] = __ballot(!(b1[b] > thr));   // (minimum +inf -- the wave holds no node yet -- : every lane)
            }
            uint32_t thv = 0;
#pragma unroll
            for (int b = 0; b < kPassQ; ++b) {
                const int nnear = __popcll(nearm[b]);
                const uint32_t wl = nearm[b] ? (uint32_t)(__ffsll((unsigned long long)nearm[b]) - 1) : 0u;
                const uint64_t otherm = nearm[b] & (nearm[b] - 1ull);
                const uint32_t ol = (nnear == 2) ? (0x40000000u | ((uint32_t)(__ffsll((unsigned long long)otherm) - 1) << 16)) : 0u;
                const uint32_t t = (wave * 64u + wl) | (nnear != 1 ? 0x80000000u : 0u) | ol;
                asm("v_writelane_b32 %0, %1, %2" : "+v"(thv) : "s"(t), "n"(8 * b));   // lane 8b <- the wave-uniform record
            }
            if ((lane & 7u) == 0 && lane < 8u * (uint32_t)kPassQ) {   // lane 8b publishes query b (records of the slots beyond a short last pass are never read)
                const uint32_t slot = (j + (lane >> 3)) & (kQRing - 1);
                sh.pub_k1[wave][slot] = lf32_bits(u);
                sh.pub_th[wave][slot] = thv;
            }
            if (lane == 0) {
                sh.pub_nc[wave][((j & (kQRing - 1)) / (uint32_t)kPassQ) & (kQRing / 4 - 1)] = nc;
                lds_post(&sh.wave_done[wave], need);   // after the records (LDS is in order within a wave)
            }
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
        const double goal_radius = p.goal_r[prob];   // (the disc goal sampler scales by it)
        __syncthreads();  // the launch's second barrier (see the scanners)
        // while a tree grows everybody downstream waits for this wave: never queue behind the two scanner waves of this SIMD.
        // (Not in the steady measurement: ChaCha12 and the draws are ~10 instructions per query, and at top priority they
        // slowed the two scanners here -- the waves the other six then wait for: 953 -> 900 M it/s.)
        if (!p.freeze) __builtin_amdgcn_s_setprio(3);
        RngWindow rng;
        rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st0.draws);
        uint32_t js = 0;
        while (js < budget) {
            uint32_t jr_seen = 0, depth_now = depth;
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
                // While a tree is small a round commits few lanes (~sqrt(2n)) and every query in flight has to fold in every node
                // committed since its scan: the window follows the tree size (n / 8, at least two passes) up to `depth`.
                depth_now = depth;
                if (!p.freeze) {
                    const uint32_t dn = uni(lds_peek(&sh.committed)) / (uint32_t)OXHIP_WINDOW_DIV;
                    depth_now = dn < 2u * (uint32_t)kPassQ ? 2u * (uint32_t)kPassQ : (dn < depth ? dn : depth);
                }
                if (js - jr_seen + 2u * (uint32_t)kPassQ <= depth_now || js == jr_seen) { go = true; break; }   // two scanner passes' worth of the window is free: refill it
                __builtin_amdgcn_s_sleep(2);
            }
            if (!go) break;  // stop requested (or a protocol bug: the resolver's own guard reports it)
            uint32_t m = jr_seen + depth_now > js ? jr_seen + depth_now - js : (uint32_t)kPassQ;  // free window slots
            if (m > 64u) m = 64u;
            if (m > (uint32_t)kPassQ) m -= (js + m) & (uint32_t)(kPassQ - 1);     // the scanners consume whole passes: end the batch on a pass boundary
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
            if (!sample_lanes<DIM>(rng, p, goal_c, goal_radius, c0, m, lane, sh, js)) {
                for (uint32_t b = 0; b < m; ++b) {  // (never expected) a redraw ran past the window: one by one
                    double qn[D];
                    sample_state<D, false>(rng, p, DIM, goal_c, qn, goal_radius);
                    const uint32_t slot = (js + b) & (kQRing - 1);
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            sh.q[k][slot] = qn[k];
                            sh.qf[k][slot] = -2.0f * (float)(qn[k] - c0[k]);
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
    uint64_t n_wrap = 0, n_forced = 0, n_amb = 0, n_rounds = 0, n_lanes = 0, n_cut_conflict = 0, t_wait = 0, t_work = 0, t_exact = 0, n_tie = 0, n_memo = 0, n_pair = 0, n_fold_trips = 0, n_fold_exact = 0, n_conf_trips = 0, n_conf_exact = 0, t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = STAMP ? (uint64_t)clock64() : 0;
    const uint64_t t_begin = t_mark, rt_begin = STAMP ? (uint64_t)__builtin_amdgcn_s_memrealtime() : 0;

    __syncthreads();  // pairs with the scanners' second barrier: mabs_bits is final
    float sph_a[D], sph_cc = 0.0f, sph_thr32 = -1.0f;   // sphere `lane` as the binary32 pre-filter holds it
    LMargins mg;
    {
        // H: the tree as loaded (binary32 roundings, hence the 1 + 2^-23), the bounds and the goal centre, relative to c0
        double h = (double)lbits_f32(lds_peek(&sh.mabs_bits)) * (1.0 + 0x1p-23);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            h = fmax(h, fmax(fabs(p.lo[k] - c0[k]), fabs(p.hi[k] - c0[k])));
            h = fmax(h, fabs(goal_c[k] - c0[k]));
        }
        h = unid(h) * 1.001;   // interpolation rounding over any chain of inserts
        const double u = 0x1p-24;
        mg.usable = h < 1e15 && fabs(c0[0]) < 1e300;  // also false for NaN / inf
        mg.e2 = 2.0 * (u * h * h * (double)(D * (3 * D + 9)) * 1.0001 + 1e-290);
        // the spheres for the binary32 pre-filter of the motion check: same error model with H_f = max(H, |centre - c0|)
        double hs = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) hs = fmax(hs, lane < ns64 ? fabs(oc[k] - c0[k]) : 0.0);
        const double hf = fmax(h, wave_max_f64pos(hs));
        const double ef = 2.0 * (u * hf * hf * (double)(D * (3 * D + 9)) * 1.0001 + 1e-290);
        double sq = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float f = (float)(oc[k] - c0[k]);
            sh.obs32[lane][k] = f;
            sq += (double)f * (double)f;
        }
        sh.obs32[lane][D] = (float)sq;
        sph_cc = (float)sq;
        // a sphere is cleared when s' + |m|^2 > thr: thr = (filter threshold + 2 E_f), rounded up (and two ulps more for the sum)
        float thr = (float)((ofilt + ef) * (1.0 + 0x1p-21));
        thr = thr + fabsf(thr) * 0x1p-22f;
        sph_thr32 = (lane < ns64 && mg.usable && hf < 1e15 && ofilt >= 0.0) ? thr : (lane < ns64 ? __builtin_inff() : -1.0f);
        sh.obs32_thr[lane] = sph_thr32;
#pragma unroll
        for (int k = 0; k < D; ++k) sph_a[k] = sh.obs32[lane][k];
    }

    typedef double ldouble4 __attribute__((ext_vector_type(4)));
    constexpr int kTrip = OXHIP_LANES_TRIP(DIM);   // candidate blocks fetched per memory round trip
    constexpr int kFold = OXHIP_LANES_FOLD;       // ring nodes screened per trip of the fold
    // the last whole-tree answer: valid while the tree has not grown (see the exact path)
    uint32_t memo_n = 0xFFFFFFFFu, memo_idx = kNoNode;
    double memo_g = 0.0, memo_q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) memo_q[k] = 0.0;

    uint64_t t_pm = 0;
#define OXHIP_PHASE(IDX) do { if (STAMP) { const uint64_t now_ = (uint64_t)clock64(); t_ph[IDX] += now_ - t_pm; t_pm = now_; } } while (0)
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
                const uint32_t coming = (sampled_now >= budget ? budget : (sampled_now & ~(uint32_t)(kPassQ - 1))) - jr;
                uint32_t want = p.freeze ? depth / 2 : depth / OXHIP_WANT_DIV_GROW;
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
        if (STAMP) t_pm = (uint64_t)clock64();
        const bool act = lane < m;
        const uint32_t slot = (jr + (act ? lane : 0u)) & (kQRing - 1);
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = sh.q[k][slot];
        const uint64_t pos_after_l = sh.pos_after[slot];
        // this wave's own earlier stores to the tree have reached the cache the loads below read (same CU; a wait, no cache op)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // the eight waves' screens: the scanner lane that holds the smallest s', the smallest s' of every other lane, oldest snapshot
        float K1 = __builtin_inff(), K2 = __builtin_inff(), K3 = __builtin_inff();
        uint32_t wth = 0, wth2 = 0, bmin = 0xFFFFFFFFu;
        {
            LanePub rec[kScanWaves];
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                rec[w].k1 = sh.pub_k1[w][slot];
                rec[w].th = sh.pub_th[w][slot];
                rec[w].nc = sh.pub_nc[w][(slot / (uint32_t)kPassQ) & (kQRing / 4 - 1)];
            }
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                K1 = fminf(K1, lbits_f32(rec[w].k1));   // (never NaN: see the scanner)
                bmin = rec[w].nc < bmin ? rec[w].nc : bmin;
            }
            bool taken = false;
            uint32_t w1 = 0;
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                const bool win = !taken && lbits_f32(rec[w].k1) == K1;   // the first wave attaining the minimum
                K2 = win ? K2 : fminf(K2, lbits_f32(rec[w].k1));         // K2: the smallest s' of the OTHER waves
                wth = win ? rec[w].th : wth;
                w1 = win ? (uint32_t)w : w1;
                taken = taken || win;
            }
            // the runner-up wave and K3, the smallest s' of the remaining six
            taken = false;
#pragma unroll
            for (int w = 0; w < kScanWaves; ++w) {
                const bool second = !taken && (uint32_t)w != w1 && lbits_f32(rec[w].k1) == K2;
                K3 = (second || (uint32_t)w == w1) ? K3 : fminf(K3, lbits_f32(rec[w].k1));
                wth2 = second ? rec[w].th : wth2;
                taken = taken || second;
            }
        }
        if (__ballot(act && (n - bmin > (uint32_t)(kNRing - 64) || bmin > n)) != 0) { stop = 4; break; }  // ring would have wrapped (bug guard)
        const uint32_t wthread = wth & 0x1FFu;   // (bit 31 of the record: the wave could not separate its two best lanes)
        // the winning scanner lane's own nodes, in binary64 from the tree (four consecutive nodes per block of rows), then
        // the nodes committed after the oldest snapshot, from the LDS ring: ascending indices, so ties keep the lower index
        Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
        // The whole-tree path's last answer (below) is the exact nearest node of ITS query for the tree as it was then: a lane
        // with that very query -- the goal centre, drawn again and again: when the screen cannot decide it once it cannot the
        // next time either -- starts from that answer and only has to look at the nodes committed since (they are in the LDS
        // ring), instead of ending the round's prefix and scanning the whole tree every time.
        bool from_memo = act && memo_n <= n && n - memo_n <= ((p.dbg_flags & OXHIP_DEBUG_SHORT_MEMO) ? 8u : (uint32_t)(kNRing - 64));
#pragma unroll
        for (int k = 0; k < D; ++k) from_memo = from_memo && __double_as_longlong(q[k]) == __double_as_longlong(memo_q[k]);
        if (from_memo) {
            pd.b1 = memo_g; pd.slot = memo_idx;   // (stored only when the minimum was unique: no second node within a rounding)
            bmin = memo_n;
        }
        if (STAMP) n_memo += (uint64_t)__popcll(__ballot(from_memo));
        // When the screen leaves exactly TWO scanner lanes in question -- the winner and one other lane of its wave (named in
        // the record), or the winners of two waves with every third wave out of reach -- both lanes' nodes are looked at, which
        // again covers every node whose s' is within 2.25 E of K1; only three-way cases go to the whole-tree path.
        const double m225 = (double)K1 + 1.125 * mg.e2;
        const bool flag1 = (wth & 0x80000000u) != 0, waves_out = (double)K2 > m225;   // (NaN / inf: false)
        uint32_t wthread2 = 0xFFFFFFFFu;
        if (waves_out && (wth & 0x40000000u) != 0) wthread2 = (wth & 0x1C0u) | ((wth >> 16) & 63u);
        else if (!flag1 && !waves_out && (double)K3 > m225 && (wth2 & 0x80000000u) == 0) wthread2 = wth2 & 0x1FFu;
        const bool pair = OXHIP_LANES_PAIR && (p.dbg_flags & (OXHIP_DEBUG_PAIR_TO_WHOLE_TREE | OXHIP_DEBUG_ALL_WHOLE_TREE)) == 0 && act && !from_memo && mg.usable && wthread2 != 0xFFFFFFFFu;
        // (two blocks per trip: the tree of a whole batch does not fit the L2, a trip is a memory round trip)
#pragma nounroll
        for (int cpass = 0; cpass < 2; ++cpass) {   // (a tie between the two lanes' nodes is flagged by scan_push whatever the order)
        if (cpass == 1) {
            if (__ballot(pair) == 0) break;
            if (STAMP) ++n_pair;
        }
        const uint32_t thread_ = cpass == 0 ? wthread : (pair ? wthread2 : 0u);
        const bool on_ = cpass == 0 ? (act && !from_memo) : pair;
#pragma unroll
        for (int blk0 = 0; blk0 < S / 4; blk0 += kTrip) {
            uint32_t ib2[kTrip], il2[kTrip], sk2[kTrip];
            bool have2[kTrip];
#pragma unroll
            for (int h = 0; h < kTrip; ++h) {
                ib2[h] = blk0 + h < S / 4 ? Lay::block_base(thread_, (uint32_t)(blk0 + h)) : kNoNode;
                have2[h] = on_ && ib2[h] < bmin;      // (kNoNode fails; nodes >= bmin come from the ring below)
                il2[h] = have2[h] ? ib2[h] : 0u;
            }
            if (__ballot(have2[0] || have2[kTrip - 1]) == 0) continue;       // a small tree fills the first blocks only
            // the reference's sum, coordinate by coordinate (0.0 + x*x == x*x, then + y*y, ...): eight nodes' sums side by side
            double d8[kTrip][4];
#pragma unroll
            for (int h = 0; h < kTrip; ++h) sk2[h] = *reinterpret_cast<const uint32_t*>(skip + il2[h]);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                ldouble4 ck[kTrip];
#pragma unroll
                for (int h = 0; h < kTrip; ++h) ck[h] = *reinterpret_cast<const ldouble4*>(tree + (size_t)k * cap + il2[h]);
#pragma unroll
                for (int h = 0; h < kTrip; ++h) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double df = ck[h][t] - q[k];
                        const double sq = df * df;
                        d8[h][t] = k == 0 ? sq : d8[h][t] + sq;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < kTrip; ++h) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (have2[h] && ib2[h] + (uint32_t)t < bmin && ((sk2[h] >> (8 * t)) & 0xFFu) == 0) scan_push(pd, d8[h][t], ib2[h] + (uint32_t)t);
            }
        }
        }
        OXHIP_PHASE(0);   // combine + the winning lane's candidates
        // this lane's query as the screens see it: Q = -2 fl32(q - c0), |b|^2
        float Qf[D];
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float bk = (float)(q[k] - c0[k]);
            Qf[k] = -2.0f * bk;
            bb += (double)bk * (double)bk;
        }
        {
            // A ring node can become this lane's nearest node, or tie with it, only if its d2 is at most T = g0 (1 + 2^-19)
            // (g0 = the best so far); its screen value then obeys s' <= T - |b|^2 + E.  Everything else is skipped after
            // D fused multiply-adds; the few that pass get the reference's binary64 distance.
            float thr = (mg.usable && pd.b1 < 1e300) ? f32_up(pd.b1 * (1.0 + 0x1p-19) - bb + mg.e2) : __builtin_inff();
            const uint32_t lo = wave_min_u32(act ? bmin : 0xFFFFFFFFu);
            // (eight ring nodes per trip: their LDS reads -- wave-uniform addresses -- are issued together)
            for (uint32_t i0 = lo; i0 < n; i0 += kFold) {
                // which of the trip's nodes concern this lane at all: i0 + t in [bmin, n) -- one mask per trip instead of
                // two compares per node
                const uint32_t t_hi = n - i0 < (uint32_t)kFold ? n - i0 : (uint32_t)kFold;
                const uint32_t t_lo = bmin > i0 ? (bmin - i0 < (uint32_t)kFold ? bmin - i0 : (uint32_t)kFold) : 0u;
                const uint32_t valid = act ? (((1u << t_hi) - 1u) & ~((1u << t_lo) - 1u)) : 0u;
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < kFold; ++t) {
                    const uint32_t i = i0 + (uint32_t)t;
                    const float* nf = sh.newn32[i & (kNRing - 1)];   // (slots past n hold stale nodes: masked by `valid`)
                    float sp = nf[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sp = __builtin_fmaf(nf[k], Qf[k], sp);
                    screen_bit(bits, sp, thr);   // (NaN: look)
                }
                const uint32_t lookm = (__brev(bits) >> (32 - kFold)) & valid;
                if (STAMP) ++n_fold_trips;
                if (__ballot(lookm != 0) != 0) {
#pragma unroll
                    for (int t = 0; t < kFold; ++t) {
                        if (__ballot((lookm >> t) & 1u) != 0) {
                            if (STAMP) ++n_fold_exact;
                            const uint32_t i = i0 + (uint32_t)t;
                            double c[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) c[k] = sh.newn[k][i & (kNRing - 1)];
                            const double d = dist2<D>(c, q, DIM);
                            if ((lookm >> t) & 1u) scan_push(pd, d, i);  // ascending i: ties keep the lower index (+inf for skipped duplicates)
                        }
                    }
                    // a lane whose best just improved looks at fewer of the nodes to come (a lane that had no candidate at all
                    // -- its nearest node is itself young -- started with thr = +inf)
                    if (mg.usable && pd.b1 < 1e300) thr = f32_up(pd.b1 * (1.0 + 0x1p-19) - bb + mg.e2);
                }
            }
        }
        OXHIP_PHASE(1);   // ring fold
        // accept iff every node this lane did not look at -- the other scanner lanes' -- is provably farther: their smallest s'
        // exceed K1 + 2.25 E (other waves: K2; the winning wave's other lanes: its own 2.5 E ballot) and g <= K1 + |b|^2 + E, hence
        //   d_other^2 >= s'_other + |b|^2 - E  >  K1 + |b|^2 + 1.25 E  >=  g + E / 4
        bool clear = act && mg.usable && pd.slot != kNoNode && ((!flag1 && waves_out) || pair) &&
                     (pd.b1 <= (double)K1 + bb + 0.5 * mg.e2);   // (NaN anywhere: false)
        if (p.dbg_flags & OXHIP_DEBUG_ALL_WHOLE_TREE) clear = false;   // (tests: no screen verdict is trusted)
        clear = clear || from_memo;   // (the memoized answer + the fold over everything committed since = the whole tree)
        {
            // keep the answer current: the first such lane's result holds for the tree of n nodes
            const uint64_t mm_ = __ballot(from_memo && pd.h2 > hi32(pd.b1) + 1);
            if (mm_ != 0) {
                const int ml = __ffsll((unsigned long long)mm_) - 1;
                memo_g = readlane_f64(pd.b1, ml);
                memo_idx = (uint32_t)__builtin_amdgcn_readlane((int)pd.slot, ml);
                memo_n = n;
            }
        }
        const double g = pd.b1;
        const uint32_t hb = hi32(g) + 1;
        const uint32_t nearest = pd.slot;
        // ambiguous iff that proof fails, or a second candidate is within a rounding of the binary64 minimum
        const bool amb = act && (!clear || pd.h2 <= hb);
        double q_near[D], qn[D], mid[D];
        {
            const uint32_t ni = nearest == kNoNode ? 0u : nearest;
            const bool from_ring = ni >= bmin;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double gl = tree[(size_t)k * cap + (from_ring ? 0u : ni)];
                q_near[k] = from_ring ? sh.newn[k][ni & (kNRing - 1)] : gl;
            }
        }
        steer<DIM>(p, false, g, q_near, q, qn);
        const bool dup = g == 0.0;
        OXHIP_PHASE(2);   // nearest node's coordinates + steer
        // check_motion (rrt.rs:90-116): the midpoint filter names the spheres the segment can touch at all ...
        bool bad = false;
        uint64_t maybe_dbg = 0;
        if (nobs > 0) {
            lerp<DIM>(q_near, qn, 0.5, mid, DIM);
            // (binary32 first: a sphere whose screen value proves d2(centre, mid) > filter threshold is cleared after D fused
            // multiply-adds, an add and a compare; only the rest get the binary64 test.  Eight spheres per trip: their LDS
            // reads -- wave-uniform addresses, broadcast -- are issued together)
            float Qm[D], mm = 0.0f;
            {
                double mmd = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float mk = (float)(mid[k] - c0[k]);
                    Qm[k] = -2.0f * mk;
                    mmd += (double)mk * (double)mk;
                }
                mm = (float)(mmd * (1.0 - 0x1p-22));   // rounded down: errs towards "maybe"
            }
            uint32_t maybe_lo = 0, maybe_hi = 0;
            if (m <= (uint32_t)OXHIP_LANES_TRANSPOSE_MAX) {
            // Small rounds (growing trees), transposed: lane s holds sphere s; the round's midpoints are staged in LDS and visited
            // one by one (wave-uniform reads, four per trip), each costing D fused multiply-adds, an add, a compare (= the
            // 64-sphere mask of that query) and two v_writelane into the query's own lane -- a round of 10 lanes pays for 10
            // midpoints, not for 64 spheres.
            {
#pragma unroll
                for (int k = 0; k < D; ++k) sh.fq[lane][k] = Qm[k];
                sh.fq[lane][D] = mm;
                for (uint32_t j0 = 0; j0 < m; j0 += 4) {
                    uint64_t mk[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float* fqj = sh.fq[(j0 + (uint32_t)t) & 63u];
                        float sp = sph_cc;
#pragma unroll
                        for (int k = 0; k < D; ++k) sp = __builtin_fmaf(sph_a[k], fqj[k], sp);
                        mk[t] = __ballot(!(sp + fqj[D] > sph_thr32));   // (NaN / inf threshold: maybe; -1: no sphere)
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (j0 + (uint32_t)t < m) {
                            const uint32_t jl = uni(j0 + (uint32_t)t);
                            // (one scalar operand per VALU instruction on gfx9: the lane select goes through M0)
                            // (M0 is the compiler's: saved and restored around its use)
                            uint32_t m0_saved;
                            asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                                         : "+v"(maybe_lo), "+v"(maybe_hi), "=&s"(m0_saved) : "s"((uint32_t)mk[t]), "s"(jl), "s"((uint32_t)(mk[t] >> 32)));
                        }
                    }
                }
            }
            } else {
            // Full rounds: every lane walks the 64 spheres (eight LDS reads -- wave-uniform, broadcast -- per trip)
            for (uint32_t o0 = 0; o0 < ns64; o0 += 8) {
                uint32_t b8 = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float* of = sh.obs32[o0 + t];
                    float sp = of[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sp = __builtin_fmaf(of[k], Qm[k], sp);
                    screen_bit(b8, sp + mm, sh.obs32_thr[o0 + t]);   // !(sp + mm > thr)  (NaN / inf threshold: maybe; -1: no sphere)
                }
                const uint32_t bits = rev8(b8);
                if (o0 < 32) maybe_lo |= bits << o0; else maybe_hi |= bits << (o0 - 32);
            }
            }
            // the binary64 filter for the spheres the pre-filter left (per lane: usually none or one)
            {
                uint64_t rem = ((uint64_t)maybe_hi << 32) | maybe_lo;
                uint64_t keep = 0;
                while (__ballot(rem != 0) != 0) {
                    const bool has = rem != 0;
                    const uint32_t o = has ? (uint32_t)(__ffsll((unsigned long long)rem) - 1) : 0u;
                    double c[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                    if (has && sphere_maybe_hit<DIM>(c, sh.obs[D + 1][o], mid)) keep |= 1ull << o;
                    rem &= rem - 1;
                }
                maybe_lo = (uint32_t)keep;
                maybe_hi = (uint32_t)(keep >> 32);
            }
            const uint64_t maybe = ((uint64_t)maybe_hi << 32) | maybe_lo;
            OXHIP_PHASE(3);   // sphere filter
            if (STAMP) maybe_dbg = maybe;
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
        OXHIP_PHASE(4);   // motion check
        if (STAMP && p.dbg && (p.dbg_flags & OXHIP_DEBUG_AUDIT) != 0) {   // audit (diagnostic instantiation with OXHIP_DEBUG_AUDIT): an accepted motion whose end state lies inside one of the first 64 spheres
            bool inval = false;
            for (uint32_t o = 0; o < ns64; ++o) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                inval = inval || !(dist2<D>(c, qn, DIM) > sh.obs[D][o]);
            }
            const uint64_t badm = __ballot(act && !amb && !bad && inval);
            if (badm != 0 && lane == (uint32_t)(__ffsll((unsigned long long)badm) - 1)) {
                atomicAdd((unsigned long long*)&p.dbg[50], 1ull);
                p.dbg[51] = ((uint64_t)prob << 48) | ((uint64_t)m << 40) | ((uint64_t)lane << 32) | (st.iterations + lane);
                p.dbg[52] = ((uint64_t)maybe_dbg);
                p.dbg[53] = 1;
            }
        }
        const bool ok = act && !bad;
        const bool ins = !p.freeze;
        // this lane's new node as the scanners (and the binary32 screens) will hold it
        float nf_a[D], nf_cc;
        {
            double sq = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                nf_a[k] = (float)(qn[k] - c0[k]);
                sq += (double)nf_a[k] * (double)nf_a[k];
            }
            nf_cc = (float)sq;
        }

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
            // (binary32 first, as in the ring fold: the new node of lane i matters to lane j only if d2 <= g_j (1 + 2^-19))
            // The round's would-be new nodes are staged in the ring slots they will occupy (beyond `committed`: nobody reads
            // them yet, and the slots belong to nodes too old to matter); the t-th of them concerns the lanes behind its own,
            // i.e. the lanes with more than t inserting lanes below them.
            const uint64_t newm = __ballot(ok && !dup);
            const uint32_t rank = (uint32_t)__popcll(newm & below_mask(lane));
            const float thr_c = (mg.usable && g < 1e300) ? f32_up(g * (1.0 + 0x1p-19) - bb + mg.e2) : __builtin_inff();
            if (ok && !dup) {
                const uint32_t sl = (n + rank) & (kNRing - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) { sh.newn32[sl][k] = nf_a[k]; sh.newn[k][sl] = qn[k]; }
                sh.newn32[sl][D] = nf_cc;
            }
            const uint32_t n_new = (uint32_t)__popcll(newm);
            for (uint32_t t0 = 0; t0 < n_new; t0 += 8) {
                // lanes in front of the first inserting lane of this trip are out of reach; so is everything behind the cut
                // the staged nodes of this trip that lie in front of this lane: t < min(rank, n_new) - t0, for lanes inside the cut
                const uint32_t ahead = rank < n_new ? rank : n_new;
                const uint32_t t_hi = ahead > t0 ? (ahead - t0 < 8u ? ahead - t0 : 8u) : 0u;
                const uint32_t valid = (act && lane < cut) ? ((1u << t_hi) - 1u) : 0u;
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float* nf = sh.newn32[(n + t0 + (uint32_t)t) & (kNRing - 1)];
                    float sp = nf[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sp = __builtin_fmaf(nf[k], Qf[k], sp);
                    screen_bit(bits, sp, thr_c);
                }
                const uint32_t lookm = rev8(bits) & valid;
                if (STAMP) ++n_conf_trips;
                if (__ballot(lookm != 0) != 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (__ballot((lookm >> t) & 1u) != 0) {
                            if (STAMP) ++n_conf_exact;
                            const uint32_t sl = (n + t0 + (uint32_t)t) & (kNRing - 1);
                            double ca[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) ca[k] = sh.newn[k][sl];
                            const uint64_t cm = __ballot(((lookm >> t) & 1u) != 0 && hi32(dist2<D>(ca, q, DIM)) <= hb);
                            if (cm != 0) {
                                const uint32_t c = (uint32_t)(__ffsll((unsigned long long)cm) - 1);
                                if (c < cut) { cut = c; stop_after = -1; if (STAMP) ++n_cut_conflict; }
                            }
                        }
                    }
                }
                // the nodes of the next trip come from lanes at or behind this one's: nothing left to learn once they pass the cut
                const uint64_t behind = newm & ~first_n_mask(cut);
                if (t0 + 8 >= n_new - (uint32_t)__popcll(behind)) break;
            }
        }

        if ((p.dbg_flags & OXHIP_DEBUG_ONE_LANE_ROUNDS) != 0 && cut > 1) { cut = 1; stop_after = -1; if (STAMP) ++n_forced; }   // (tests: the rest is re-resolved)
        OXHIP_PHASE(5);   // prefix: cap, goal, conflicts
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
                        sh.newn32[idx & (kNRing - 1)][k] = dup ? 0.0f : nf_a[k];
                        tree[(size_t)k * cap + idx] = qn[k];
                    }
                    sh.newn32[idx & (kNRing - 1)][D] = dup ? __builtin_inff() : nf_cc;
                    parent[idx] = (int32_t)nearest;
                    skip[idx] = dup ? 1 : 0;
                }
                // goal test (rrt.rs:220-223): the first hit in query order
                const uint64_t hits = hitm & cutm;
                if (hits != 0 && st.goal_node < 0)
                    st.goal_node = (int32_t)__builtin_amdgcn_readlane((int)idx, __ffsll((unsigned long long)hits) - 1);
                if (STAMP && ((n ^ (n + (uint32_t)__popcll(okm & cutm))) & ~(uint32_t)(kNRing - 1)) != 0) ++n_wrap;
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
        OXHIP_PHASE(6);   // commit
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        if (stop_after >= 0) { stop = stop_after; break; }
        if (cut == m || ambm == 0 || (uint32_t)(__ffsll((unsigned long long)ambm) - 1) != cut) continue;

        // ---- the lane at `cut` is ambiguous: that one query over the WHOLE binary64 tree.  First by squared distances, the
        //      wave striding over the nodes (four in flight per lane): if exactly one node is within a rounding of the
        //      minimum, it is the reference's nearest node (sqrt is monotone).  Only a genuine near-tie -- two d2 that may
        //      share a correctly rounded root -- takes the reference's literal loop (post-sqrt compare, lowest index).
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }
        {
            if (STAMP) ++n_amb;
            const uint64_t t_e0 = STAMP ? (uint64_t)clock64() : 0;
            const uint32_t slot1 = jr & (kQRing - 1);
            double q1[D];
#pragma unroll
            for (int k = 0; k < D; ++k) q1[k] = unid(sh.q[k][slot1]);
            // (a fixed query on an unchanged tree has a fixed answer: the goal centre is drawn again and again -- goal_bias --
            // and when the screen cannot decide it once, it cannot decide it the next time either)
            bool same_q = memo_n == n;
#pragma unroll
            for (int k = 0; k < D; ++k) same_q = same_q && __double_as_longlong(q1[k]) == __double_as_longlong(memo_q[k]);
            double gmin = memo_g;
            uint32_t memo_hit_idx = memo_idx;
            bool tie = false;
            if (!same_q) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores to the tree have landed (same CU)
                Scan ps{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
                // a lane takes four consecutive nodes per 32-byte load and coordinate, four such groups per trip: 16 nodes per
                // lane in flight (the trees of a whole batch do not fit the L2: a trip is a DRAM / Infinity Cache round trip)
                for (uint32_t i0 = 4u * lane; i0 < n; i0 += 1024u) {
                    uint32_t sk4[4], il[4];
                    double d16[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t ib = i0 + 256u * (uint32_t)t;
                        il[t] = ib < n ? ib : 0u;   // (rows are padded to cap >= n rounded up to 1024)
                        sk4[t] = *reinterpret_cast<const uint32_t*>(skip + il[t]);
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        ldouble4 ck[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) ck[t] = *reinterpret_cast<const ldouble4*>(tree + (size_t)k * cap + il[t]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double df = ck[t][r] - q1[k];
                                const double sq = df * df;
                                d16[t][r] = k == 0 ? sq : d16[t][r] + sq;   // the reference's summation order
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t i = i0 + 256u * (uint32_t)t + (uint32_t)r;
                            if (i < n && ((sk4[t] >> (8 * r)) & 0xFFu) == 0) scan_push(ps, d16[t][r], i);   // ascending within the lane: ties keep the lower index
                        }
                    }
                }
                gmin = wave_min_f64(ps.b1);
                const uint32_t hbw = hi32(gmin) + 1;
                const uint64_t nearm = __ballot(ps.slot != kNoNode && hi32(ps.b1) <= hbw);
                tie = __popcll(nearm) != 1 || __ballot(ps.h2 <= hbw) != 0;
                memo_hit_idx = tie ? kNoNode : (uint32_t)__builtin_amdgcn_readlane((int)ps.slot, __ffsll((unsigned long long)(nearm | (1ull << 63))) - 1);
                if (!tie) {
                    memo_n = n; memo_g = gmin; memo_idx = memo_hit_idx;
#pragma unroll
                    for (int k = 0; k < D; ++k) memo_q[k] = q1[k];
                }
            } else if (STAMP) ++n_memo;
            uint32_t nearest1;
            double qn1[D], q_near1[D];
            bool dup1;
            if (!tie) {
                nearest1 = memo_hit_idx;
#pragma unroll
                for (int k = 0; k < D; ++k) q_near1[k] = unid(tree[(size_t)k * cap + nearest1]);
                dup1 = gmin == 0.0;
                steer<DIM>(p, false, gmin, q_near1, q1, qn1);
            } else {
                if (STAMP) ++n_tie;
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
                nearest1 = uni(e.idx);
#pragma unroll
                for (int k = 0; k < D; ++k)
                    q_near1[k] = unid(__hip_atomic_load(&tree[(size_t)k * cap + nearest1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const double dist1 = unid(e.dist);
                dup1 = dist1 == 0.0;
                steer<DIM>(p, true, dist1, q_near1, q1, qn1);
            }
            bool ok1 = true;
            if (nobs > 0) {
                double mid1[D];
                lerp<DIM>(q_near1, qn1, 0.5, mid1, DIM);
                if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid1)) != 0 || extras)
                    ok1 = motion_lanes<DIM>(p, lane, q_near1, qn1, oc, othr, ofilt, ns64);
            }
            if (STAMP && p.dbg && (p.dbg_flags & OXHIP_DEBUG_AUDIT) != 0) {   // the same audit for the one-query path
                const bool inval1 = lane < ns64 && !(dist2<D>(oc, qn1, DIM) > othr);
                if (ok1 && __ballot(inval1) != 0 && lane == 0) {
                    atomicAdd((unsigned long long*)&p.dbg[50], 1ull);
                    p.dbg[51] = ((uint64_t)prob << 48) | st.iterations;
                    p.dbg[52] = __ballot(inval1);
                    p.dbg[53] = 2 | (tie ? 4 : 0) | (same_q ? 8 : 0);
                }
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
                        double sq1 = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const float f1 = (float)(qn1[k] - c0[k]);
                            sq1 += (double)f1 * (double)f1;
                            sh.newn[k][i & (kNRing - 1)] = dup1 ? __builtin_inf() : qn1[k];
                            sh.newn32[i & (kNRing - 1)][k] = dup1 ? 0.0f : f1;
                            tree[(size_t)k * cap + i] = qn1[k];
                        }
                        sh.newn32[i & (kNRing - 1)][D] = dup1 ? __builtin_inff() : (float)sq1;
                        parent[i] = (int32_t)nearest1;
                        skip[i] = dup1 ? 1 : 0;
                    }
                    ++n;
                    if (STAMP && (n & (uint32_t)(kNRing - 1)) == 0) ++n_wrap;
                    if (lane == 0) lds_post(&sh.committed, n);
                    if (dist2<D>(qn1, goal_c, DIM) <= goal_thr) {
                        if (st.goal_node < 0) st.goal_node = (int32_t)i;
                        hit1 = true;
                    }
                }
            }
            jr += 1;
            if (lane == 0) lds_post(&sh.resolved, jr);
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_exact += now - t_e0; t_mark = now; }
            if (hit1 && p.stop_at_goal) { stop = 0; break; }
        }
    }
    if (lane == 0) {
        lds_post(&sh.stop_flag, 1);
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
        if (STAMP && p.dbg) {   // over all problems: the largest exact-path count (with its problem id) and lifetime, and the sums
            atomicMax((unsigned long long*)&p.dbg[44], (unsigned long long)((n_amb << 32) | prob));
            atomicAdd((unsigned long long*)&p.dbg[45], (unsigned long long)n_amb);
            atomicMax((unsigned long long*)&p.dbg[46], (unsigned long long)((((uint64_t)clock64() - t_begin) << 16) | (prob & 0xFFFFu)));
            atomicAdd((unsigned long long*)&p.dbg[47], (unsigned long long)((uint64_t)clock64() - t_begin));
            atomicAdd((unsigned long long*)&p.dbg[48], (unsigned long long)n_pair);
            atomicAdd((unsigned long long*)&p.dbg[54], (unsigned long long)n_amb);
            atomicAdd((unsigned long long*)&p.dbg[55], (unsigned long long)n_memo);
            atomicAdd((unsigned long long*)&p.dbg[56], (unsigned long long)n_cut_conflict);
            atomicAdd((unsigned long long*)&p.dbg[57], (unsigned long long)n_wrap);
            atomicAdd((unsigned long long*)&p.dbg[58], (unsigned long long)n_pair);
            atomicAdd((unsigned long long*)&p.dbg[59], (unsigned long long)n_forced);
        }
        if (STAMP && p.dbg && prob == 0) {
            p.dbg[4] = n_amb; p.dbg[5] = n_rounds; p.dbg[6] = n_lanes; p.dbg[7] = st.iterations; p.dbg[12] = n_cut_conflict; p.dbg[1] = t_wait; p.dbg[2] = t_work; p.dbg[3] = t_exact; p.dbg[15] = n_tie; p.dbg[11] = n_memo;
            for (int i = 0; i < 8; ++i) p.dbg[32 + i] = t_ph[i];
            p.dbg[40] = n_fold_trips; p.dbg[41] = n_fold_exact; p.dbg[42] = n_conf_trips; p.dbg[43] = n_conf_exact;
            p.dbg[13] = (uint64_t)clock64() - t_begin; p.dbg[14] = (uint64_t)__builtin_amdgcn_s_memrealtime() - rt_begin;
        }
    }
}

#ifndef OXHIP_LANES_S
#define OXHIP_LANES_S 24
#define OXHIP_LANES_C 16
#endif
constexpr int kLS = OXHIP_LANES_S, kLC = OXHIP_LANES_C;   // register rows of the six heavy / of the two resolver-side scanner waves (R^2, R^3)
// R^4 .. R^6: a row costs DIM + 1 registers, so the rows are spread evenly (20 x 512 = 10,240 nodes); R^2 / R^3 also have a
// 40 / 32-row instantiation for trees of up to 20,480 / 16,384 nodes
template <int DIM> struct LanesShape { static constexpr int S = 20, C = 20, SB = 0, CB = 0; };
template <> struct LanesShape<2> { static constexpr int S = kLS, C = kLC, SB = 40, CB = 40; };
template <> struct LanesShape<3> { static constexpr int S = kLS, C = kLC, SB = 32, CB = 32; };

template <int DIM>
static int pick_rows_lanes(uint32_t cap) {
    if (cap <= 2048u) return 4;
    if (cap <= Layout4<LanesShape<DIM>::S, LanesShape<DIM>::C>::kCapacity) return LanesShape<DIM>::S;
    if (LanesShape<DIM>::SB != 0 && cap <= Layout4<LanesShape<DIM>::SB ? LanesShape<DIM>::SB : 4, LanesShape<DIM>::SB ? LanesShape<DIM>::CB : 4>::kCapacity)
        return LanesShape<DIM>::SB;
    return 0;
}

bool lanes_supported(uint32_t dim, uint32_t cap) {
    switch (dim) {
        case 2: return pick_rows_lanes<2>(cap) != 0;
        case 3: return pick_rows_lanes<3>(cap) != 0;
        case 4: return pick_rows_lanes<4>(cap) != 0;
        case 5: return pick_rows_lanes<5>(cap) != 0;
        case 6: return pick_rows_lanes<6>(cap) != 0;
        default: return false;
    }
}

template <int DIM, int S, int C>
static void launch_lanes_shape(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kLanesThreads);
    if (p.dbg) hipLaunchKernelGGL((rrt_lanes_kernel<DIM, S, C, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((rrt_lanes_kernel<DIM, S, C, false>), grid, block, 0, stream, p);
}
template <int DIM>
static void launch_lanes_dim(const DevParams& p, hipStream_t stream) {
    const int s = pick_rows_lanes<DIM>(p.cap);
    if (s == 4) launch_lanes_shape<DIM, 4, 4>(p, stream);
    else if (s == LanesShape<DIM>::S) launch_lanes_shape<DIM, LanesShape<DIM>::S, LanesShape<DIM>::C>(p, stream);
    else if constexpr (LanesShape<DIM>::SB != 0) launch_lanes_shape<DIM, LanesShape<DIM>::SB, LanesShape<DIM>::CB>(p, stream);
}

void launch_rrt_lanes(const DevParams& p, hipStream_t stream) {
    switch (p.dim) {
        case 2: launch_lanes_dim<2>(p, stream); break;
        case 3: launch_lanes_dim<3>(p, stream); break;
        case 4: launch_lanes_dim<4>(p, stream); break;
        case 5: launch_lanes_dim<5>(p, stream); break;
        case 6: launch_lanes_dim<6>(p, stream); break;
        default: break;
    }
}

}  // namespace oxhip
