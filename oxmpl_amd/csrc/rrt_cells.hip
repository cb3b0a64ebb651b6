// rrt_cells.hip -- RRT grow loop, ONE WAVE PER PLANNING PROBLEM, nearest neighbour through an exact cell grid (R^2, R^3).
//
// rrt_lanes.hip scans every node for every query (the reference's loop, rrt.rs:187-196, as a register-resident screen) and
// spends a whole CU on one problem.  This kernel does less work and needs no CU-wide choreography:
//
// * lane j of the wave IS query jr + j (as in rrt_lanes.hip's resolver): sample, nearest neighbour, steer, motion check,
//   commit of the longest prefix that keeps the reference's sequential semantics -- all inside one wave, no hand-off to
//   other waves, no workgroup barrier after the launch's first.
// * nearest neighbour: the nodes live in a uniform grid of cubic cells over the bounding box of the bounds, the goal centre
//   and the tree -- G = 2^floor(log2(n) / D) cells along the longest side, ~1 to 2^D nodes per cell, re-gridded whenever n
//   reaches the next power -- as one 64-byte BLOCK per cell in HBM / L2: a counter and seven entries (x, y, z, node), the
//   position inside the cell in 2^-16 cell units (a fuller cell chains further blocks).  A lane reads its query's own cell,
//   then -- four blocks in flight at a time -- only those of the other 3^D - 1 cells whose box comes within d1 + 2A of the
//   query, keeps the smallest and the second smallest binary32 squared distance, and accepts the smallest iff
//     d1 + A < d2 - A        (no other visited node can be nearer or tie:  A = sqrt(D) delta, delta the bound on any stored
//                             coordinate's and the query's error in cell units: Gmax 2^-23 + the measured clamping error)
//     d1 + A < lb            (no unvisited node can: lb = distance from the query to the nearest open face of the block)
//   then the winner's distance, steer, motion check, tree and checksum are binary64 from the binary64 node, exactly as in
//   every other kernel.  When lb fails (a query deep inside an obstacle, a sparse tree) the WAVE searches the next shells of
//   cells for that one query; when the margin fails (1e-4 of queries) the query takes the whole-tree path: the reference's
//   d2 scan / literal loop over the binary64 tree.  Trees of up to 1,024 nodes are scanned node by node (uniform addresses).
// * inserting a node is one atomic increment of its cell's counter and an 8-byte store: the grid is always current, there is
//   no "nodes committed since the snapshot" to fold in, no rebuild, no sorted order to maintain.
// * with inserts suppressed (the steady measurement) the queries of a launch are independent: a problem's launch is cut into
//   `split` contiguous parts, one wave each (stream positions of the part starts from cells_prepare_kernel), whose checksum
//   polynomials and counters are summed with atomics; the last part to finish writes the problem's state.
//
// Everything that enters a result is binary64 in the reference's evaluation order; the grid only names a candidate.
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"
#include "lane_query_common.hpp"

namespace oxhip {

#ifndef OXHIP_CELLS_BRUTE
#define OXHIP_CELLS_BRUTE 512
#endif
constexpr uint32_t kBruteMax = OXHIP_CELLS_BRUTE;   // trees up to this size: every node, by index
constexpr uint32_t kFlatCap = 4096;                  // entries of a problem's flat list (oxhip_api.hip allocates them)
static_assert(kBruteMax <= kFlatCap, "the flat list holds the small trees");
constexpr int kCellsWaves = 4;                       // waves (= problems, or parts of problems) per workgroup
constexpr uint32_t kSelfSkipMax = 4;                 // up to this many parts, a part skips ahead to its start itself
constexpr uint32_t kMaxSplit = 64;                   // parts a frozen launch of one problem is cut into, at most (cell_part_pos's row length)
constexpr int kMaxShell = 6;                         // the cooperative search gives up beyond this ring (-> whole-tree path)
#ifndef OXHIP_CELLS_NB
#define OXHIP_CELLS_NB 4                             // neighbour cells in flight per trip
#endif
#ifndef OXHIP_ABLATE
#define OXHIP_ABLATE 0   // (timing experiments only: 1 no sphere pre-filter, 2 no neighbour cells, 4 no motion check, 8 first trip without faces)
#endif
#ifndef OXHIP_CELLS_WT_UNROLL
#define OXHIP_CELLS_WT_UNROLL 4
#endif
#ifndef OXHIP_CELLS_NB_FROZEN
#define OXHIP_CELLS_NB_FROZEN 3                      // ... of a frozen launch (three waves per SIMD hide the latency; 16 registers per cell in flight)
#endif
#ifndef OXHIP_CELLS_FILL_X2
#define OXHIP_CELLS_FILL_X2 7ull
#endif
#ifndef OXHIP_CELLS_TAIL
#define OXHIP_CELLS_TAIL 64                          // at most this many outstanding (query, cell) pairs: one pair per lane (0: off)
#endif

typedef float cfloat4 __attribute__((ext_vector_type(4)));
typedef uint32_t cuint4 __attribute__((ext_vector_type(4)));
typedef double cdouble4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kBlkEntries = 7;

struct CellGrid {   // wave-uniform
    double lo[3], inv_h;
    uint32_t G[3];
    uint32_t level, regrid_at;
    uint32_t pool_next;  // next free overflow block
    double delta_node;   // bound on |stored - true| of any node coordinate, cell units
};

template <int DIM>
struct CellsWaveLds {
    uint32_t rng_buf[16][64];
    double q[DIM][64];            // the round's queries, slot (jr + lane) & 63
    uint64_t pos_after[64];
    double newn[DIM][64];         // the round's would-be new nodes, by rank
    float newn32[64][4];          // ... as the dot-product pre-screen holds them: fl32(x - c0), fl32(|.|^2)
    uint32_t tail_pair[64];       // the tail pass: (owner lane | neighbour number << 8) of the k-th outstanding (query, cell) pair ...
    float tail_s1[64], tail_s2[64];   // ... and what lane k found in that cell
    uint32_t tail_i1[64];
};
template <int DIM>
struct CellsShared {
    double obs[DIM + 2][64];      // first 64 spheres: centre, validity threshold, filter threshold
    CellsWaveLds<DIM> w[kCellsWaves];
};

// Grid level for a tree of n nodes: level l has G_l ~ 2^(l/2) cells along the longest side (steps of sqrt 2, so a regrid
// thins the cells by 2^(D/2), not 2^D) and serves trees of up to 3.5 G_l^D nodes -- at most 3.5 nodes per cell on average (a
// block holds seven: the Poisson tail beyond is 3 %), ~1.2 right after a regrid (sparser cells fail the lb test too often).
__host__ __device__ __forceinline__ uint32_t cells_G(uint32_t level) {
    const uint32_t t[15] = {1u, 2u, 2u, 3u, 4u, 6u, 8u, 11u, 16u, 23u, 32u, 45u, 64u, 91u, 128u};
    return t[level < 14u ? level : 14u];
}
__host__ __device__ __forceinline__ uint32_t cells_level_cap(uint32_t level, int dim) {   // largest tree level `level` serves
    unsigned long long v = OXHIP_CELLS_FILL_X2;   // (twice the nodes per cell a level may reach)
    for (int k = 0; k < dim; ++k) v *= cells_G(level);
    v >>= 1;
    return v > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)v;
}
__device__ __forceinline__ uint32_t cells_level(uint32_t n, int dim, uint32_t level_max) {
    if (n <= kBruteMax) return 0u;
    uint32_t l = 1;
    while (l < level_max && cells_level_cap(l, dim) < n) ++l;
    return l;
}

template <int DIM>
__device__ __forceinline__ void grid_load(const CellMeta& m, CellGrid& g) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.lo[k] = unid(m.lo[k]);
        g.G[k] = uni(m.G[k]);
    }
    g.inv_h = unid(m.inv_h);
    g.level = uni(m.level);
    g.regrid_at = uni(m.regrid_at);
    g.pool_next = uni(m.pool_next);
    g.delta_node = (double)__builtin_bit_cast(float, uni(__builtin_bit_cast(uint32_t, m.delta_node)));
}

// Where node x goes: its cell, its block entry (position inside the cell in 2^-16 cell units, decoded at the bin centre),
// its position in cell units as binary32 (the flat list of small trees), and the error either representation makes.
template <int DIM>
__device__ __forceinline__ uint32_t cell_place(const CellGrid& g, const double x[DIM], uint32_t idx, uint64_t& entry, float tf[3], double& err) {
    uint32_t c[3] = {0u, 0u, 0u}, u[3] = {0u, 0u, 0u};
    err = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tf[k] = 0.0f;
        if (k < DIM) {
            const double t = (x[k] - g.lo[k]) * g.inv_h;
            tf[k] = (float)t;
            const double fl = floor(t);
            const uint32_t ci = fl > 0.0 ? (fl < (double)g.G[k] ? (uint32_t)fl : g.G[k] - 1u) : 0u;   // (NaN -> 0; never expected)
            const double fr = (t - (double)ci) * 65536.0;
            const uint32_t ui = fr > 0.0 ? (fr < 65535.0 ? (uint32_t)fr : 65535u) : 0u;
            const double dec = (double)ci + ((double)ui + 0.5) * 0x1p-16;
            err = fmax(err, g.level == 0 ? fabs((double)tf[k] - t) : fabs(dec - t));
            c[k] = ci;
            u[k] = ui;
        }
    }
    entry = (uint64_t)(u[0] | (u[1] << 16)) | ((uint64_t)(u[2] | (idx << 16)) << 32);
    return (c[2] * g.G[1] + c[1]) * g.G[0] + c[0];
}

// Append `entry` to cell `cell` for the lanes with `pred`: slot = the cell's counter, atomically (several lanes of a round may
// share a cell); slots beyond the head block go to chained blocks, one lane at a time (rare: a cell holds ~1 .. 2^D nodes).
// In two halves, so that a round's commit can put its reductions and its checksum between the atomic and the use of its result.
__device__ __forceinline__ uint32_t cells_insert_begin(CellBlock* blk, uint32_t cell, bool pred) {
    uint32_t slot = 0;
    if (pred) slot = __hip_atomic_fetch_add(&blk[cell].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return slot;
}
__device__ __forceinline__ void cells_insert_finish(CellBlock* blk, uint32_t& pool_next, uint32_t cell, uint64_t entry, bool pred, uint32_t slot, uint32_t lane) {
    if (pred && slot < kBlkEntries) blk[cell].e[slot] = entry;
    uint64_t over = __ballot(pred && slot >= kBlkEntries);
    while (over != 0) {
        const int j = __ffsll((unsigned long long)over) - 1;
        over &= over - 1;
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)cell, j), sj = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
        const uint64_t ej = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(entry >> 32), j) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)entry, j);
        uint32_t b = cj;
        for (uint32_t t = 0; t < sj / kBlkEntries; ++t) {
            // (this wave's earlier link stores have landed: a wait, not a cache flush -- only this wave ever touches the problem's blocks)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t nx = uni(__hip_atomic_load(&blk[b].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (nx == 0) {
                nx = pool_next++;
                if (lane == 0) {
                    blk[nx].count = 0;
                    blk[nx].next = 0;
                    __hip_atomic_store(&blk[b].next, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            b = nx;
        }
        if (lane == 0) blk[b].e[sj % kBlkEntries] = ej;
    }
}
__device__ __forceinline__ void cells_insert(CellBlock* blk, uint32_t& pool_next, uint32_t cell, uint64_t entry, bool pred, uint32_t lane) {
    const uint32_t slot = cells_insert_begin(blk, cell, pred);
    cells_insert_finish(blk, pool_next, cell, entry, pred, slot, lane);
}

// (Re)build the grid of problem `prob` for its n nodes: one wave.  Chooses the level from n, the box from the bounds, the
// goal centre and the nodes; files every node that is not a skipped duplicate.
template <int DIM>
__device__ __forceinline__ void cells_build(const DevParams& p, uint32_t prob, uint32_t n, uint32_t lane, const double* c0, CellGrid& g,
                                            uint32_t& mabs_bits) {
    const size_t cap = p.cap;
    const double* tree = p.tree + (size_t)prob * DIM * cap;
    const uint8_t* skip = p.skip + (size_t)prob * cap;
    CellBlock* blk = p.cell_blk + (size_t)prob * p.cell_blocks;
    cfloat4* flat = reinterpret_cast<cfloat4*>(p.cell_flat) + (size_t)prob * kFlatCap;
    cdouble4* xyz = reinterpret_cast<cdouble4*>(p.cell_xyz) + (size_t)prob * cap;
    double lo[DIM], hi[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double gc = p.goal_c[(size_t)prob * DIM + k];
        lo[k] = fmin(p.lo[k], gc);
        hi[k] = fmax(p.hi[k], gc);
    }
    uint32_t mab = 0;
    for (uint32_t i = lane; i < n; i += 64) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
            const double x = tree[(size_t)k * cap + i];
            lo[k] = fmin(lo[k], x);
            hi[k] = fmax(hi[k], x);
            const uint32_t ab = lf32_bits((float)(x - c0[k])) & 0x7FFFFFFFu;
            mab = ab > mab ? ab : mab;
        }
    }
    mabs_bits = wave_max_u32(mab);
    double wmax = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        lo[k] = wave_min_f64(lo[k]);
        hi[k] = -wave_min_f64(-hi[k]);
        wmax = fmax(wmax, hi[k] - lo[k]);
    }
    g.level = cells_level(n, DIM, p.cell_level_max);
    const uint32_t G = cells_G(g.level);
    g.inv_h = (double)G / wmax;   // (wmax > 0: create() refuses lo >= hi)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < DIM) {
            g.lo[k] = lo[k];
            uint32_t gk = (uint32_t)((hi[k] - lo[k]) * g.inv_h) + 1u;
            g.G[k] = gk < G ? gk : G;
        } else {
            g.lo[k] = 0.0;
            g.G[k] = 1u;
        }
    }
    g.regrid_at = n <= kBruteMax ? kBruteMax + 1u : (g.level < p.cell_level_max ? cells_level_cap(g.level, DIM) + 1u : 0xFFFFFFFFu);
    const uint32_t nc = g.G[0] * g.G[1] * g.G[2];
    if (g.level != 0)
        for (uint32_t c = lane; c < nc; c += 64) { blk[c].count = 0; blk[c].next = 0; }
    g.pool_next = nc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    double derr = 0.0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool in = i < n;
        double x[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) x[k] = tree[(size_t)k * cap + (in ? i : 0u)];
        float tf[3];
        double err;
        uint64_t entry;
        const uint32_t cell = cell_place<DIM>(g, x, i, entry, tf, err);
        if (in) xyz[i] = cdouble4{x[0], x[1], DIM >= 3 ? x[DIM - 1] : 0.0, 0.0};
        const bool keep = in && skip[in ? i : 0u] == 0;   // a duplicate of a lower-index node can never win (strict '<', rrt.rs:192)
        if (keep) derr = fmax(derr, err);
        if (g.level == 0) {
            if (in) flat[i] = keep ? cfloat4{tf[0], tf[1], tf[2], 0.0f} : cfloat4{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f};
        } else {
            cells_insert(blk, g.pool_next, cell, entry, keep, lane);
        }
    }
    derr = -wave_min_f64(-derr);
    g.delta_node = (double)f32_up(derr * (1.0 + 1e-9) + 1e-30);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

template <int DIM>
__device__ __forceinline__ void grid_store(CellMeta& m, const CellGrid& g, uint32_t n, uint32_t mabs_bits, uint32_t lane) {
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { m.lo[k] = g.lo[k]; m.G[k] = g.G[k]; }
        m.inv_h = g.inv_h;
        m.level = g.level;
        m.regrid_at = g.regrid_at;
        m.pool_next = g.pool_next;
        m.n_grid = n;
        m.delta_node = (float)g.delta_node;   // (exact: it came from a float)
        m.mabs_bits = mabs_bits;
        m.valid = 1u;
    }
}

// Lane-parallel sampling of m <= 64 consecutive queries (rrt.rs:177-184 + rvss.rs:233-249; sample_batch of
// rrt_resident_common.hpp with this kernel's ring).  STORE = false only advances the stream position (the fast-forward of
// cells_prepare_kernel).  Returns false, nothing written, when a range draw was rejected or the window is too short.
template <int DIM, bool STORE>
__device__ __forceinline__ bool cells_sample(RngWindow& rng, const DevParams& p, const double* goal_c, double goal_radius, uint32_t m, uint32_t lane,
                                             CellsWaveLds<DIM>* sh, uint32_t js) {
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
    if (STORE && act) {
        const uint32_t slot = (js + lane) & 63u;
#pragma unroll
        for (int k = 0; k < DIM; ++k) sh->q[k][slot] = q[k];
        sh->pos_after[slot] = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}

// draw (STORE) or skip the queries [js, js + m): window refill, the lane-parallel sampler, the sequential fallback
template <int DIM, bool STORE>
__device__ __forceinline__ void cells_sample_block(RngWindow& rng, const DevParams& p, const double* goal_c, double goal_radius, uint32_t m, uint32_t lane,
                                                   CellsWaveLds<DIM>* sh, uint32_t js) {
    // (exactly what cells_sample asks of the window: a fresh window of 512 words serves two rounds of 64 uniform samples in R^3
    //  most of the time -- any slack here means ChaCha blocks computed twice)
    const uint64_t need_hi = rng.pos + (uint64_t)m * (1 + DIM);
    if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
        rng.base_blk = uni64(rng.pos >> 3);
        uint32_t o[16];
        chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
    }
    if (!cells_sample<DIM, STORE>(rng, p, goal_c, goal_radius, m, lane, sh, js)) {
        for (uint32_t b = 0; b < m; ++b) {   // (never expected) a redraw, or a batch past the window: one by one
            double qn[DIM];
            sample_state<DIM, false>(rng, p, DIM, goal_c, qn, goal_radius);
            if (STORE && lane == 0) {
                const uint32_t slot = (js + b) & 63u;
#pragma unroll
                for (int k = 0; k < DIM; ++k) sh->q[k][slot] = qn[k];
                sh->pos_after[slot] = rng.pos;
            }
        }
    }
}

// inclusive prefix sum over the wave (DPP: shifts within the rows of 16, then the row totals)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_scan_step(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v = dpp_scan_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_scan_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_scan_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_scan_step<0x118, 0xf>(v);   // row_shr:8
    v = dpp_scan_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v = dpp_scan_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
    return v;
}

// smallest / second smallest screen value and the node holding the smallest
struct Top2 {
    float s1, s2;
    uint32_t i1;   // node index << 16 (the lower bits are whatever the entry's word held)
};
__device__ __forceinline__ void top2_push(Top2& t, float s, uint32_t i) {
    const bool lt = s < t.s1;
    t.i1 = lt ? i : t.i1;
    float m;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(s), "v"(t.s1), "v"(t.s2));   // the second smallest of three with s1 <= s2
    t.s2 = m;
    vmin_f32(t.s1, s);
}
template <int DIM>
__device__ __forceinline__ float cell_s(const cfloat4& nd, const float (&tq)[3]) {
    const float e0 = nd[0] - tq[0];
    float s = e0 * e0;
    const float e1 = nd[1] - tq[1];
    s = __builtin_fmaf(e1, e1, s);
    if (DIM >= 3) {
        const float e2 = nd[2] - tq[2];
        s = __builtin_fmaf(e2, e2, s);
    }
    return s;
}

// one block (16 dwords: count, next, seven entries) against a query: the entries e < nvalid are pushed, two at a time through
// the packed binary32 pipe (v_pk_fma_f32 / v_pk_mul_f32: the same IEEE operations as the scalar forms).  off = (cell + half a
// bin) - query, so that a coordinate decodes with one fused multiply-add.  Top2::i1 holds the entry's upper word as it is
// stored -- the node index in its upper 16 bits -- and is shifted once, when the round's winner is read.
typedef float cells_f32x2 __attribute__((ext_vector_type(2)));
template <int DIM>
__device__ __forceinline__ void block_eval(const cuint4 (&v)[4], uint32_t nvalid, const float (&off)[3], Top2& t2) {
    const cells_f32x2 sc = {0x1p-16f, 0x1p-16f};
#pragma unroll
    for (int e = 0; e < (int)kBlkEntries; e += 2) {
        if (__ballot((uint32_t)e < nvalid) == 0) break;   // (uniform)
        constexpr int kLast = (int)kBlkEntries - 1;
        const int e1 = e + 1 <= kLast ? e + 1 : kLast;
        const uint32_t lo0 = v[(2 + 2 * e) / 4][(2 + 2 * e) % 4], hi0 = v[(3 + 2 * e) / 4][(3 + 2 * e) % 4];
        const uint32_t lo1 = v[(2 + 2 * e1) / 4][(2 + 2 * e1) % 4], hi1 = v[(3 + 2 * e1) / 4][(3 + 2 * e1) % 4];
        const cells_f32x2 ex = __builtin_elementwise_fma(cells_f32x2{(float)(lo0 & 0xFFFFu), (float)(lo1 & 0xFFFFu)}, sc, cells_f32x2{off[0], off[0]});
        const cells_f32x2 ey = __builtin_elementwise_fma(cells_f32x2{(float)(lo0 >> 16), (float)(lo1 >> 16)}, sc, cells_f32x2{off[1], off[1]});
        cells_f32x2 s = ex * ex;
        s = __builtin_elementwise_fma(ey, ey, s);
        if (DIM >= 3) {
            const cells_f32x2 ez = __builtin_elementwise_fma(cells_f32x2{(float)(hi0 & 0xFFFFu), (float)(hi1 & 0xFFFFu)}, sc, cells_f32x2{off[2], off[2]});
            s = __builtin_elementwise_fma(ez, ez, s);
        }
        top2_push(t2, (uint32_t)e < nvalid ? s[0] : __builtin_inff(), hi0);
        if (e + 1 <= kLast) top2_push(t2, (uint32_t)(e + 1) < nvalid ? s[1] : __builtin_inff(), hi1);
    }
}
__device__ __forceinline__ void block_load(const CellBlock* blk, uint32_t b, cuint4 (&v)[4]) {
    const cuint4* src = reinterpret_cast<const cuint4*>(blk + b);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = src[t];
}
// the chained blocks of a cell that holds more than seven nodes (entries 7 .. cnt - 1)
template <int DIM>
__device__ __forceinline__ void chain_follow(const CellBlock* blk, uint32_t next, uint32_t cnt, bool on, const float (&off)[3], Top2& t2) {
    uint32_t base = kBlkEntries;
    bool go = on && cnt > kBlkEntries && next != 0;
    while (__ballot(go) != 0) {
        cuint4 v[4];
        block_load(blk, go ? next : 0u, v);
        const uint32_t left = go ? cnt - base : 0u;
        block_eval<DIM>(v, left < kBlkEntries ? left : kBlkEntries, off, t2);
        base += kBlkEntries;
        next = v[0][1];
        go = go && base < cnt && next != 0;
    }
}
// a whole cell for the lanes with `on`
template <int DIM>
__device__ __forceinline__ void cell_visit(const CellBlock* blk, uint32_t cell, bool on, const uint32_t (&cc)[3], const float (&tq)[3], Top2& t2) {
    float off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) off[k] = ((float)cc[k] + 0x1p-17f) - tq[k];
    cuint4 v[4];
    block_load(blk, on ? cell : 0u, v);
    const uint32_t cnt = on ? v[0][0] : 0u;
    block_eval<DIM>(v, cnt < kBlkEntries ? cnt : kBlkEntries, off, t2);
    if (__ballot(cnt > kBlkEntries) != 0) chain_follow<DIM>(blk, v[0][1], cnt, on, off, t2);
}

// distance (cell units) from tq to the nearest OPEN face of the block of cells [cq - r, cq + r] (+inf: the block is the grid)
template <int DIM>
__device__ __forceinline__ double block_lb(const CellGrid& g, const float (&tq)[3], const uint32_t (&cq)[3], uint32_t r) {
    double lb = __builtin_inf();
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        if (cq[k] > r) lb = fmin(lb, (double)tq[k] - (double)(cq[k] - r));                     // cells below the block exist
        if (cq[k] + r + 1u < g.G[k]) lb = fmin(lb, (double)(cq[k] + r + 1u) - (double)tq[k]);    // cells above
    }
    return lb;
}

// the acceptance test: 0 = proven (the smallest is the unique nearest node), 1 = search the next shell, 2 = ambiguous
__device__ __forceinline__ int cells_verdict(const Top2& t, double lb, double A) {
    if (!(t.s1 < __builtin_inff())) return lb < __builtin_inf() ? 1 : 2;   // nothing found yet
    const double d1 = sqrt((double)t.s1) * (1.0 + 0x1p-20) + A;
    const double d2 = sqrt((double)t.s2) * (1.0 - 0x1p-20) - A;
    if (!(d1 < d2)) return 2;
    if (!(d1 < lb * (1.0 - 0x1p-20))) return 1;
    return 0;
}

// First round of part `part` when a frozen launch of `rounds` rounds is cut into `split` parts.  A part that finds its own
// start (split <= kSelfSkipMax) pays for the rounds in front of it -- about 1 / kSkipCostInv of a round's work each -- so the
// later parts get fewer rounds: equal finishing times need begin_p ~ (1 - (1 - c)^p) / (1 - (1 - c)^split) of the rounds.
#ifndef OXHIP_CELLS_SKIP_COST_INV
#define OXHIP_CELLS_SKIP_COST_INV 16
#endif
__host__ __device__ __forceinline__ uint32_t cells_part_begin(uint32_t rounds, uint32_t part, uint32_t split) {
    if (part >= split) return rounds;
    if (split > kSelfSkipMax) return (uint32_t)(((uint64_t)rounds * part) / split);
    constexpr uint64_t ci = OXHIP_CELLS_SKIP_COST_INV;
    uint64_t full = 1, keep_s = 1, keep_p = 1;   // ci^split, (ci - 1)^split, (ci - 1)^part ci^(split - part)
    for (uint32_t i = 0; i < split; ++i) { full *= ci; keep_s *= ci - 1u; keep_p *= i < part ? ci - 1u : ci; }
    return (uint32_t)(((uint64_t)rounds * (full - keep_p)) / (full - keep_s));
}

#ifndef OXHIP_CELLS_WAVES_PER_EU
#define OXHIP_CELLS_WAVES_PER_EU 2
#endif
#ifndef OXHIP_CELLS_WAVES_PER_EU_FROZEN
#define OXHIP_CELLS_WAVES_PER_EU_FROZEN 3   // the frozen specialisation (no insert, no prefix code) of a steady-state launch: 168 registers
#endif
template <int DIM, bool STAMP, bool FROZEN>
__global__ __launch_bounds__(kCellsWaves * 64, FROZEN ? OXHIP_CELLS_WAVES_PER_EU_FROZEN : OXHIP_CELLS_WAVES_PER_EU) void rrt_cells_kernel(DevParams p) {
    constexpr int D = DIM;
    __shared__ CellsShared<DIM> shared;

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;
    // problems are dealt to the XCDs round-robin (workgroup w runs on XCD w % 8) and all parts of a problem stay on one XCD:
    // they share its L2 copy of the grid
    const uint32_t split = FROZEN ? p.cells_split : 1u;
    const uint32_t xcd = blockIdx.x & 7u, unit = (blockIdx.x >> 3) * (uint32_t)kCellsWaves + wave;
    const uint32_t prob = (unit / split) * 8u + xcd, part = unit % split;

    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const bool extras = nobs > ns64;
    double c0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c0[k] = 0.5 * p.lo[k] + 0.5 * p.hi[k];
    // sphere `lane` in LDS: the lane-parallel checks read any sphere, the wave-wide motion check (whole-tree path, repairs) reads
    // its own lane's back into registers when it runs (not held across the rounds: ten registers)
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) shared.obs[k][lane] = lane < ns64 ? p.sph_c[(size_t)k * p.n_spheres + lane] : 0.0;
        shared.obs[D][lane] = lane < ns64 ? p.sph_thr[lane] : -1.0;
        shared.obs[D + 1][lane] = lane < ns64 ? p.sph_filt[lane] : -1.0;
    }
    const bool live = prob < p.n_problems;
    const ProblemState st0 = p.state[live ? prob : 0u];
    CellsWaveLds<DIM>* sh = &shared.w[wave];
    CellMeta& meta = p.cell_meta[live ? prob : 0u];
    CellGrid grid;
    grid_load<DIM>(meta, grid);
    LMargins mg;
    {
        double h = (double)lbits_f32(uni(meta.mabs_bits)) * (1.0 + 0x1p-23);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            h = fmax(h, fmax(fabs(p.lo[k] - c0[k]), fabs(p.hi[k] - c0[k])));
            h = fmax(h, fabs(p.goal_c[(size_t)(live ? prob : 0u) * DIM + k] - c0[k]));
        }
        h = unid(h) * 1.001;   // interpolation rounding over any chain of inserts
        const double u = 0x1p-24;
        mg.usable = h < 1e15 && fabs(c0[0]) < 1e300;
        mg.e2 = 2.0 * (u * h * h * (double)(D * (3 * D + 9)) * 1.0001 + 1e-290);
    }
    __syncthreads();   // the launch's only barrier: the obstacle tables are in LDS
    if (!live) return;
    if (p.stop_at_goal && st0.goal_node >= 0) return;
    // the midpoint filter's lookup (DevParams::sph_grid): cell = (mid - lo) G / (hi - lo), from fl32(mid - c0)
    float sg_inv[D], sg_bias[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double inv = (double)p.sph_grid_G / (p.hi[k] - p.lo[k]);
        sg_inv[k] = lbits_f32(uni(lf32_bits((float)inv)));
        sg_bias[k] = lbits_f32(uni(lf32_bits((float)((c0[k] - p.lo[k]) * inv))));
    }

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    uint8_t* skip = p.skip + (size_t)prob * cap;
    CellBlock* blk = p.cell_blk + (size_t)prob * p.cell_blocks;
    cfloat4* flat = reinterpret_cast<cfloat4*>(p.cell_flat) + (size_t)prob * kFlatCap;
    cdouble4* xyz = reinterpret_cast<cdouble4*>(p.cell_xyz) + (size_t)prob * cap;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    const double goal_thr = p.goal_thr[prob];
    const double goal_radius = p.goal_r[prob];   // (the disc goal sampler scales by it)

    // this wave's share of the launch: iterations [j_lo, j_hi) of the problem's budget
    const uint32_t budget_all = (uint32_t)p.budget;
    uint32_t j_lo = 0, j_hi = budget_all;
    uint64_t pos_start = st0.draws;
    if (split > 1) {
        const uint32_t rounds = (budget_all + 63u) / 64u;
        j_lo = cells_part_begin(rounds, part, split) * 64u;
        j_hi = cells_part_begin(rounds, part + 1u, split) * 64u;
        if (j_hi > budget_all) j_hi = budget_all;
        if (split > kSelfSkipMax) pos_start = p.cell_part_pos[(size_t)prob * kMaxSplit + part];   // (else: found below)
    }
    const uint32_t budget = j_hi - j_lo;

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

    RngWindow rng;
    rng.init(sh->rng_buf, p.seed, p.first_problem_id + prob, pos_start);
    if (split > 1 && split <= kSelfSkipMax && j_lo > 0) {
        // With few parts a part finds its own start: the sampler's position arithmetic over the queries in front of it (a goal
        // sample draws one word, a uniform one 1 + D; nothing else is evaluated).  It runs in the shadow of the SIMD's other waves'
        // memory stalls -- a separate pass over every problem before the launch (cells_prepare_kernel) cost 6 % of a steady launch.
        for (uint32_t r0 = 0; r0 < j_lo; r0 += 64u) cells_sample_block<DIM, false>(rng, p, goal_c, goal_radius, 64u, lane, sh, r0);
        pos_start = rng.pos;
    }
    ProblemState st = st0;
    if (split > 1) { st.checksum = 0; st.accepted = 0; st.iterations = 0; }   // this part's own sums (combined at the end)
    uint64_t draws_done = pos_start;
    uint32_t n = st.n_nodes;
    uint32_t mabs_bits = uni(meta.mabs_bits);
    uint32_t jr = 0, js = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    uint64_t n_rounds = 0, n_lanes = 0, n_amb = 0, n_expand = 0, n_cut_conflict = 0, n_tie = 0, n_memo = 0, n_forced = 0, n_regrid = 0, n_steps = 0, n_tail = 0, n_tail_pairs = 0;
    uint64_t t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_pm = 0, t_c[4] = {0, 0, 0, 0}, t_cm = 0;
    uint64_t h_lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h_cells[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h_trips[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin = STAMP ? (uint64_t)clock64() : 0;
#define OXHIP_CPHASE(IDX) do { if (STAMP) { const uint64_t now_ = (uint64_t)clock64(); t_ph[IDX] += now_ - t_pm; t_pm = now_; } } while (0)
    // the last whole-tree answer: valid while the tree has not grown
    uint32_t memo_n = 0xFFFFFFFFu, memo_idx = kNoNode;
    double memo_g = 0.0, memo_q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) memo_q[k] = 0.0;
    typedef double ldouble4 __attribute__((ext_vector_type(4)));

    while (true) {
        if (jr >= budget) { stop = 1; break; }
        if (!FROZEN && n >= p.max_nodes) { stop = 2; break; }
        if (!FROZEN && n >= grid.regrid_at) {   // the tree outgrew its grid: the next finer one
            cells_build<DIM>(p, prob, n, lane, c0, grid, mabs_bits);
            if (STAMP) ++n_regrid;
        }
        uint32_t m = budget - jr < 64u ? budget - jr : 64u;
        if (js < jr + m) {   // draw the queries this round still lacks (rrt.rs:177-184): they depend on the stream only
            cells_sample_block<DIM, true>(rng, p, goal_c, goal_radius, jr + m - js, lane, sh, js);
            js = jr + m;
        }
        if (STAMP) { ++n_rounds; n_lanes += m; t_pm = (uint64_t)clock64(); }
        const bool act = lane < m;
        const uint32_t slot = (jr + (act ? lane : 0u)) & 63u;
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = sh->q[k][slot];
        const uint64_t pos_after_l = sh->pos_after[slot];

        // ---- nearest neighbour (rrt.rs:187-196): a binary32 screen over the cells around the query names one candidate
        Top2 t2{__builtin_inff(), __builtin_inff(), kNoNode};
        float tq[3] = {0.0f, 0.0f, 0.0f};
        uint32_t cq[3] = {0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            tq[k] = (float)((q[k] - grid.lo[k]) * grid.inv_h);
            const float fl = floorf(tq[k]);
            cq[k] = fl > 0.0f ? (fl < (float)grid.G[k] ? (uint32_t)fl : grid.G[k] - 1u) : 0u;   // (NaN -> 0)
        }
        const uint32_t gmax = grid.G[0] > grid.G[1] ? (grid.G[0] > grid.G[2] ? grid.G[0] : grid.G[2]) : (grid.G[1] > grid.G[2] ? grid.G[1] : grid.G[2]);
        const double A = sqrt((double)D) * (grid.delta_node + (double)(gmax + 8u) * 0x1p-23) * 1.01 + 1e-30;
        int verdict;   // 0 proven, 2 ambiguous (whole-tree path)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's own earlier stores (tree, cell blocks) have landed: the loads below may read them
        bool from_memo = act && memo_n == n;
#pragma unroll
        for (int k = 0; k < D; ++k) from_memo = from_memo && __double_as_longlong(q[k]) == __double_as_longlong(memo_q[k]);
        if (grid.level == 0) {
            // every node, by index: a coalesced load gives each lane one node of the next 64, then node after node is
            // broadcast from its lane (v_readlane -> scalar operands): no memory latency per node
            cfloat4 mine_nd = flat[lane < n ? lane : 0u];
            // (four independent running pairs: with one wave on the SIMD a single chain of dependent min / med3 / select
            //  instructions would wait out every instruction's latency)
            Top2 ta[4] = {{__builtin_inff(), __builtin_inff(), kNoNode}, {__builtin_inff(), __builtin_inff(), kNoNode},
                          {__builtin_inff(), __builtin_inff(), kNoNode}, {__builtin_inff(), __builtin_inff(), kNoNode}};
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const cfloat4 cur_nd = mine_nd;
                const uint32_t nx = i0 + 64u + lane;
                mine_nd = flat[nx < n ? nx : 0u];   // (the next 64 are on their way while these are processed)
                const uint32_t cnt = n - i0 < 64u ? n - i0 : 64u;
                for (uint32_t t = 0; t < cnt; t += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t tt = t + (uint32_t)u < cnt ? t + (uint32_t)u : cnt - 1u;   // (a repeated node changes nothing but s2 ...
                        cfloat4 nd;
#pragma unroll
                        for (int k = 0; k < 3; ++k) nd[k] = lbits_f32((uint32_t)__builtin_amdgcn_readlane((int)lf32_bits(cur_nd[k]), (int)tt));
                        nd[3] = 0.0f;
                        const float sv = t + (uint32_t)u < cnt ? cell_s<DIM>(nd, tq) : __builtin_inff();   // ... so it is pushed as +inf)
                        top2_push(ta[u], sv, (i0 + tt) << 16);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // fold the four pairs: a pair's second value is >= its first, so only its value matters
                top2_push(t2, ta[u].s1, ta[u].i1);
                top2_push(t2, ta[u].s2, kNoNode);
            }
            if (STAMP) n_steps += n;
            verdict = cells_verdict(t2, __builtin_inf(), A);
            OXHIP_CPHASE(7);   // (the flat scan of a small tree)
        } else {
            // (1) the query's own cell and, in the same trip, the NEARER face cell along every axis: together they almost always
            //     hold the nearest node, so the bound that prunes the other cells is tight from the start
            float gap_lo[3], gap_hi[3];   // distance (cell units) from the query to its cell's low / high face along each axis
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gap_lo[k] = fmaxf(tq[k] - (float)cq[k], 0.0f) * (1.0f - 0x1p-20f);
                gap_hi[k] = fmaxf((float)(cq[k] + 1u) - tq[k], 0.0f) * (1.0f - 0x1p-20f);
            }
            uint32_t done_faces = 0;   // bits of `need` (faces come first there: -x +x -y +y -z +z) visited in this trip
            {
                cuint4 vo[4], vf[DIM][4];
                uint32_t fc[DIM][3];
                bool fon[DIM];
                block_load(blk, act ? (cq[2] * grid.G[1] + cq[1]) * grid.G[0] + cq[0] : 0u, vo);
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    const bool up = gap_hi[a] < gap_lo[a];
#pragma unroll
                    for (int k = 0; k < 3; ++k) fc[a][k] = cq[k] + (k == a ? (up ? 1u : 0xFFFFFFFFu) : 0u);
                    fon[a] = act && (up ? cq[a] + 1u < grid.G[a] : cq[a] > 0u);
                    done_faces |= fon[a] ? (1u << (2 * a + (up ? 1 : 0))) : 0u;
                    block_load(blk, fon[a] ? (fc[a][2] * grid.G[1] + fc[a][1]) * grid.G[0] + fc[a][0] : 0u, vf[a]);
                }
                {
                    float off[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) off[k] = ((float)cq[k] + 0x1p-17f) - tq[k];
                    const uint32_t cnt = act ? vo[0][0] : 0u;
                    block_eval<DIM>(vo, cnt < kBlkEntries ? cnt : kBlkEntries, off, t2);
                    if (__ballot(cnt > kBlkEntries) != 0) chain_follow<DIM>(blk, vo[0][1], cnt, act, off, t2);
                }
#pragma unroll
                for (int a = 0; a < DIM; ++a) {
                    float off[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) off[k] = ((float)fc[a][k] + 0x1p-17f) - tq[k];
                    const uint32_t cnt = fon[a] ? vf[a][0][0] : 0u;
                    block_eval<DIM>(vf[a], cnt < kBlkEntries ? cnt : kBlkEntries, off, t2);
                    if (__ballot(cnt > kBlkEntries) != 0) chain_follow<DIM>(blk, vf[a][0][1], cnt, fon[a], off, t2);
                }
            }
            if (STAMP) ++n_steps;
            // (2) of the other cells only those whose box comes within (d1 + 2A) of the query: every node of a skipped
            //     cell is farther than that, so it can neither beat nor tie with the final winner (whose d1 can only shrink)
            float thr2;
            {
                const double dd = sqrt((double)t2.s1) * (1.0 + 0x1p-20) + 2.0 * A;
                thr2 = t2.s1 < __builtin_inff() ? f32_up(dd * dd * (1.0 + 0x1p-20)) : __builtin_inff();
            }
            // bit b of `need` = the b-th neighbour in the order faces, edges, corners (kOrder: b -> o = (dz + 1) 9 + (dy + 1) 3 +
            // (dx + 1)): the cells most likely to hold the nearest node come first, and the bound shrinks trip by trip
            constexpr int NO = DIM == 3 ? 26 : 8;
            constexpr uint8_t kOrder3[26] = {12, 14, 10, 16, 4, 22,  9, 11, 15, 17, 3, 5, 21, 23, 1, 7, 19, 25,  0, 2, 6, 8, 18, 20, 24, 26};
            constexpr uint8_t kOrder2[8] = {12, 14, 10, 16, 9, 11, 15, 17};
            // a neighbour's box is at least sqrt(sum over its shifted axes of gap^2) away; +inf where the grid ends.  (The gaps
            // are scaled down by 2^-20: the three roundings of a sum of squares cannot lift it above the true bound.)
            float sq_lo[3], sq_hi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                sq_lo[k] = (k < DIM && cq[k] > 0u) ? gap_lo[k] * gap_lo[k] : __builtin_inff();
                sq_hi[k] = (k < DIM && cq[k] + 1u < grid.G[k]) ? gap_hi[k] * gap_hi[k] : __builtin_inff();
            }
            auto needed = [&](float thr) -> uint32_t {
                const float th = fminf(thr, 0x1.fffffep+127f);   // (no bound yet: every cell inside the grid)
                uint32_t bits = 0;   // neighbour b ends up at bit NO - 1 - b
#pragma unroll
                for (int b = 0; b < NO; ++b) {
                    const int o = DIM == 3 ? kOrder3[b] : kOrder2[b];
                    const int d[3] = {o % 3 - 1, (o / 3) % 3 - 1, o / 9 - 1};
                    float l = 0.0f;
                    bool first = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (d[k] == 0) continue;
                        const float g2 = d[k] < 0 ? sq_lo[k] : sq_hi[k];
                        l = first ? g2 : l + g2;
                        first = false;
                    }
                    screen_bit(bits, l, th);   // !(l > th)
                }
                const uint32_t mask = __brev(bits) >> (32 - NO);
                return act ? mask : 0u;
            };
            uint32_t need = (OXHIP_ABLATE & 2) ? 0u : needed(thr2) & ~done_faces;
            // a lane's next needed cells, NB cells in flight per trip
            uint32_t trip_no = 0;
            // b -> (dx + 1) + 3 (dy + 1) + 9 (dz + 1) through a 5-bit-per-entry table in two 64-bit words and change
            auto order_o = [&](uint32_t b) -> uint32_t {
                if (DIM == 3) {
                    constexpr uint64_t w0 = 12ull | 14ull << 5 | 10ull << 10 | 16ull << 15 | 4ull << 20 | 22ull << 25 | 9ull << 30 | 11ull << 35 | 15ull << 40 | 17ull << 45 | 3ull << 50 | 5ull << 55;
                    constexpr uint64_t w1 = 21ull | 23ull << 5 | 1ull << 10 | 7ull << 15 | 19ull << 20 | 25ull << 25 | 0ull << 30 | 2ull << 35 | 6ull << 40 | 8ull << 45 | 18ull << 50 | 20ull << 55;
                    constexpr uint64_t w2 = 24ull | 26ull << 5;
                    const uint64_t w = b < 12u ? w0 : (b < 24u ? w1 : w2);
                    const uint32_t sh5 = (b < 12u ? b : (b < 24u ? b - 12u : b - 24u)) * 5u;
                    return (uint32_t)(w >> sh5) & 31u;
                }
                constexpr uint64_t w0 = 12ull | 14ull << 5 | 10ull << 10 | 16ull << 15 | 9ull << 20 | 11ull << 25 | 15ull << 30 | 17ull << 35;
                return (uint32_t)(w0 >> (b * 5u)) & 31u;
            };
            while (__ballot(need != 0) != 0) {
                if (OXHIP_CELLS_TAIL != 0 && trip_no != 0) {
                    // THE TAIL: after a trip few lanes still ask, each for several cells -- a further trip would run every lane
                    // through four block evaluations for their sake.  When the outstanding (query, cell) pairs fit one per
                    // lane, they are dealt out instead: lane k evaluates pair k, the owners collect.
                    const uint32_t cntl = (uint32_t)__popc(need);
                    const uint32_t incl = wave_scan_u32(cntl);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (total <= (uint32_t)OXHIP_CELLS_TAIL) {
                        if (STAMP) { ++n_tail; n_tail_pairs += total; }
                        const uint32_t first = incl - cntl;
                        {
                            uint32_t nd = need, pos = first;
                            while (__ballot(nd != 0) != 0) {
                                if (nd != 0) {
                                    sh->tail_pair[pos & 63u] = lane | ((uint32_t)(__ffs((int)nd) - 1) << 8);
                                    nd &= nd - 1u;
                                    ++pos;
                                }
                            }
                        }
                        const bool mine = lane < total;
                        const uint32_t pr = sh->tail_pair[mine ? lane : 0u];
                        const uint32_t owner = pr & 63u;
                        float tqo[3];
                        uint32_t cqo[3], cco[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            tqo[k] = k < DIM ? lbits_f32((uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)lf32_bits(tq[k]))) : 0.0f;
                            cqo[k] = k < DIM ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)cq[k]) : 0u;
                        }
                        const uint32_t o = mine ? order_o(pr >> 8) : 13u;
                        const uint32_t o3 = (o * 11u) >> 5, o9 = (o3 * 11u) >> 5;   // o / 3, o / 9 for o < 27
                        cco[0] = cqo[0] + (o - 3u * o3) - 1u;
                        cco[1] = cqo[1] + (o3 - 3u * o9) - 1u;
                        cco[2] = cqo[2] + o9 - 1u;
                        Top2 tp{__builtin_inff(), __builtin_inff(), kNoNode};
                        cell_visit<DIM>(blk, mine ? (cco[2] * grid.G[1] + cco[1]) * grid.G[0] + cco[0] : 0u, mine, cco, tqo, tp);
                        sh->tail_s1[lane] = tp.s1;
                        sh->tail_s2[lane] = tp.s2;
                        sh->tail_i1[lane] = tp.i1;
                        for (uint32_t t = 0; __ballot(t < cntl) != 0; ++t) {
                            const uint32_t at = (first + (t < cntl ? t : 0u)) & 63u;
                            const float a1 = sh->tail_s1[at], a2 = sh->tail_s2[at];
                            const uint32_t ai = sh->tail_i1[at];
                            top2_push(t2, t < cntl ? a1 : __builtin_inff(), ai);
                            top2_push(t2, t < cntl ? a2 : __builtin_inff(), kNoNode);
                        }
                        need = 0;
                        break;
                    }
                }
                if (STAMP) {
                    ++n_steps;
                    const uint32_t tn = trip_no < 7u ? trip_no : 7u;
                    h_lanes[tn] += (uint64_t)__popcll(__ballot(need != 0));
                    h_cells[tn] += wave_sum_u64((uint64_t)__popc(need));
                    ++h_trips[tn];
                }
                ++trip_no;
                constexpr int NB = FROZEN ? OXHIP_CELLS_NB_FROZEN : OXHIP_CELLS_NB;
                bool on[NB];
                uint32_t cc[NB][3], cnt[NB];
                cuint4 v[NB][4];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    on[t] = need != 0;
                    const uint32_t b = on[t] ? (uint32_t)(__ffs((int)need) - 1) : 0u;
                    need &= need - 1u;
                    const uint32_t o = on[t] ? order_o(b) : 13u;
                    const uint32_t o3 = (o * 11u) >> 5, o9 = (o3 * 11u) >> 5;   // o / 3, o / 9 for o < 27
                    cc[t][0] = cq[0] + (o - 3u * o3) - 1u;
                    cc[t][1] = cq[1] + (o3 - 3u * o9) - 1u;
                    cc[t][2] = cq[2] + o9 - 1u;
                    block_load(blk, on[t] ? (cc[t][2] * grid.G[1] + cc[t][1]) * grid.G[0] + cc[t][0] : 0u, v[t]);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    if (__ballot(on[t]) == 0) break;
                    float off[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) off[k] = ((float)cc[t][k] + 0x1p-17f) - tq[k];
                    cnt[t] = on[t] ? v[t][0][0] : 0u;
                    block_eval<DIM>(v[t], cnt[t] < kBlkEntries ? cnt[t] : kBlkEntries, off, t2);
                    if (__ballot(cnt[t] > kBlkEntries) != 0) chain_follow<DIM>(blk, v[t][0][1], cnt[t], on[t], off, t2);
                }
                // what this trip found may rule out cells that were still on the list
                if (__ballot(need != 0) != 0) {
                    const double dd = sqrt((double)t2.s1) * (1.0 + 0x1p-20) + 2.0 * A;
                    const float thr_now = t2.s1 < __builtin_inff() ? f32_up(dd * dd * (1.0 + 0x1p-20)) : __builtin_inff();
                    need &= needed(thr_now);
                }
            }
            verdict = cells_verdict(t2, block_lb<DIM>(grid, tq, cq, 1u), A);
            OXHIP_CPHASE(0);   // own cell + the neighbour cells that matter
            // ---- a lane whose block cannot rule out the cells beyond it: the WAVE searches the next shells for that query
            uint64_t needm = __ballot(act && !from_memo && verdict == 1);
            while (needm != 0) {
                const int jl = __ffsll((unsigned long long)needm) - 1;
                needm &= needm - 1;
                if (STAMP) ++n_expand;
                float tqj[3];
                uint32_t cqj[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    tqj[k] = lbits_f32((uint32_t)__builtin_amdgcn_readlane((int)lf32_bits(tq[k]), jl));
                    cqj[k] = (uint32_t)__builtin_amdgcn_readlane((int)cq[k], jl);
                }
                Top2 tj{lbits_f32((uint32_t)__builtin_amdgcn_readlane((int)lf32_bits(t2.s1), jl)),
                        lbits_f32((uint32_t)__builtin_amdgcn_readlane((int)lf32_bits(t2.s2), jl)),
                        (uint32_t)__builtin_amdgcn_readlane((int)t2.i1, jl)};
                int vj = 1;
                for (uint32_t r = 2; vj == 1; ++r) {
                    if (r > (uint32_t)kMaxShell) { vj = 2; break; }   // far too sparse here: the whole-tree path settles it
                    const uint32_t side = 2u * r + 1u, total = DIM == 3 ? side * side * side : side * side;
                    Top2 tl{__builtin_inff(), __builtin_inff(), kNoNode};
                    for (uint32_t t = lane; t < total; t += 64) {
                        const int dx = (int)(t % side) - (int)r, dy = (int)((t / side) % side) - (int)r;
                        const int dz = DIM == 3 ? (int)(t / (side * side)) - (int)r : 0;
                        const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
                        const int cheb = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
                        const int cx = (int)cqj[0] + dx, cy = (int)cqj[1] + dy, cz = (int)cqj[2] + dz;
                        if (cheb != (int)r || cx < 0 || cy < 0 || cz < 0 || cx >= (int)grid.G[0] || cy >= (int)grid.G[1] || cz >= (int)grid.G[2]) continue;
                        const uint32_t cs[3] = {(uint32_t)cx, (uint32_t)cy, (uint32_t)cz};
                        cell_visit<DIM>(blk, ((uint32_t)cz * grid.G[1] + (uint32_t)cy) * grid.G[0] + (uint32_t)cx, true, cs, tqj, tl);
                    }
                    // the shell's two smallest over the wave (binary32 values >= +0 order like their bit patterns)
                    const uint32_t m1 = wave_min_u32(lf32_bits(tl.s1));
                    const uint64_t winm = __ballot(lf32_bits(tl.s1) == m1);
                    const uint32_t m2 = __popcll(winm) >= 2 ? m1 : wave_min_u32(lf32_bits(tl.s1) == m1 ? lf32_bits(tl.s2) : lf32_bits(tl.s1));
                    const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)tl.i1, __ffsll((unsigned long long)winm) - 1);
                    top2_push(tj, lbits_f32(m1), wi);
                    top2_push(tj, lbits_f32(m2), kNoNode);   // (only its value matters; i1 stays the holder of the smallest)
                    vj = cells_verdict(tj, block_lb<DIM>(grid, tqj, cqj, r), A);
                }
                if (lane == (uint32_t)jl) { t2 = tj; verdict = vj; }
            }
        }
        OXHIP_CPHASE(6);   // shell searches
        // ---- the candidate in binary64, exactly as the reference computes it
        uint32_t nearest = (act && verdict == 0) ? t2.i1 >> 16 : kNoNode;   // (Top2::i1: the node index in the upper 16 bits)
        double q_near[D];
        {
            const uint32_t ni = nearest == kNoNode ? 0u : nearest;
            const cdouble4 c4 = xyz[ni];   // (the node-major copy: one 32-byte access instead of D cache lines)
#pragma unroll
            for (int k = 0; k < D; ++k) q_near[k] = c4[k];
        }
        double g = nearest == kNoNode ? __builtin_inf() : dist2<D>(q_near, q, DIM);
        // the screen's own claim, checked on the binary64 value (turns a corrupted record into an unproven lane)
        bool clear = nearest != kNoNode && fabs(sqrt(g) * grid.inv_h - sqrt((double)t2.s1)) <= A + sqrt((double)t2.s1) * 0x1p-19 + 1e-30;
        if (p.dbg_flags & OXHIP_DEBUG_ALL_WHOLE_TREE) clear = false;   // (tests: no screen verdict is trusted)
        if (from_memo) {   // the whole-tree path's last answer, for this very query on this very tree
            nearest = memo_idx;
            g = memo_g;
            const cdouble4 m4 = xyz[memo_idx];
#pragma unroll
            for (int k = 0; k < D; ++k) q_near[k] = m4[k];
            clear = true;
        }
        if (STAMP) n_memo += (uint64_t)__popcll(__ballot(from_memo));
        const bool amb = act && !clear;
        const uint32_t hb = hi32(g) + 1;
        double qn[D], mid[D];
        steer<DIM>(p, false, g, q_near, q, qn);
        bool dup = g == 0.0;
        OXHIP_CPHASE(1);   // candidate + steer
        // this lane's query as the dot-product pre-screens see it: Q = -2 fl32(q - c0), |b|^2
        float Qf[D];
        double bb = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float bk = (float)(q[k] - c0[k]);
            Qf[k] = -2.0f * bk;
            bb += (double)bk * (double)bk;
        }
        // ---- check_motion (rrt.rs:90-116): the midpoint filter names the spheres the segment can touch at all ...
        bool bad = false;
        if (nobs > 0) {
            lerp<DIM>(q_near, qn, 0.5, mid, DIM);
            // ... looked up: the grid cell of the midpoint holds the spheres whose filter ball reaches the cell (a superset of
            // the spheres the midpoint test keeps; the cells are built 2^-9 wider than they are, the index below is good to
            // 2^-17 cells).  A midpoint outside the bounds (a tree handed in from outside) takes every sphere.
            uint64_t cand;
            {
                const float gs = (float)p.sph_grid_G;
                bool inside = mg.usable && !(OXHIP_ABLATE & 1);
                uint32_t ci[3] = {0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float u = __builtin_fmaf((float)(mid[k] - c0[k]), sg_inv[k], sg_bias[k]);
                    inside = inside && u >= -0x1p-10f && u <= gs + 0x1p-10f;   // (NaN: outside)
                    const float fl = floorf(u);
                    ci[k] = fl > 0.0f ? (fl < gs ? (uint32_t)fl : p.sph_grid_G - 1u) : 0u;
                }
                const uint64_t m = p.sph_grid[D == 3 ? (ci[2] * p.sph_grid_G + ci[1]) * p.sph_grid_G + ci[0] : ci[1] * p.sph_grid_G + ci[0]];
                cand = inside ? m : ~0ull;
            }
            uint32_t maybe_lo = (uint32_t)cand, maybe_hi = (uint32_t)(cand >> 32);
            {
                uint64_t rem = ((uint64_t)maybe_hi << 32) | maybe_lo;
                if (ns64 < 64) rem &= (1ull << ns64) - 1ull;
                uint64_t keep = 0;
                while (__ballot(rem != 0) != 0) {
                    const bool has = rem != 0;
                    const uint32_t o = has ? (uint32_t)(__ffsll((unsigned long long)rem) - 1) : 0u;
                    double c[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = shared.obs[k][o];
                    if (has && sphere_maybe_hit<DIM>(c, shared.obs[D + 1][o], mid)) keep |= 1ull << o;
                    rem &= rem - 1;
                }
                maybe_lo = (uint32_t)keep;
                maybe_hi = (uint32_t)(keep >> 32);
            }
            const uint64_t maybe = ((uint64_t)maybe_hi << 32) | maybe_lo;
            OXHIP_CPHASE(2);   // sphere filter
            const bool need = !(OXHIP_ABLATE & 4) && act && !amb && (maybe != 0 || extras);
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
                        for (int k = 0; k < D; ++k) c[k] = shared.obs[k][o];
                        bad = bad || (has && !(dist2<D>(c, x, DIM) > shared.obs[D][o]));
                        rem &= rem - 1;
                    }
                    for (uint32_t jx = ns64; jx < nobs; ++jx) bad = bad || (on && obstacle_hit<DIM>(p, DIM, x, jx));
                }
            }
        }
        OXHIP_CPHASE(3);   // motion check
        bool ok = act && !bad;
        constexpr bool ins = !FROZEN;
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
        const uint32_t amb_at = cut;   // lanes [0, amb_at) hold a proven nearest node of the tree as it was when the round began
        uint64_t okm = __ballot(ok);
        uint64_t hitm = 0;
        if (ins) {
            // A node accepted earlier in the round that is (nearly) as close to a later query as that query's nearest node changes
            // that query's result.  Such a query is REPAIRED in place: its nearest node is then the old one or one of those few
            // new nodes -- the reference's loop (rrt.rs:187-196: sqrt, strict '<', ascending index) over just them, the steer and
            // the motion check by the whole wave -- and the queries behind it are tested against its new node.  The would-be new
            // nodes are staged by lane; bit i of `confl` = the staged node of lane i < lane matters to this lane's query (binary32
            // first: d2 <= g (1 + 2^-19) is possible; then the binary64 value within a rounding of g).
            bool hit = ok && dist2<D>(qn, goal_c, DIM) <= goal_thr;
            uint64_t newm = __ballot(ok && !dup && lane < amb_at);
            const float thr_c = (mg.usable && g < 1e300) ? f32_up(g * (1.0 + 0x1p-19) - bb + mg.e2) : __builtin_inff();
            if (ok && !dup) {
#pragma unroll
                for (int k = 0; k < D; ++k) { sh->newn32[lane][k] = nf_a[k]; sh->newn[k][lane] = qn[k]; }
                sh->newn32[lane][D] = nf_cc;
            }
            uint64_t confl = 0;
            for (uint32_t t0 = 0; t0 + 1u < amb_at; t0 += 8) {
                const uint32_t grp = (uint32_t)(newm >> t0) & 0xFFu;   // (uniform)
                if (grp == 0) continue;
                const uint32_t ahead = lane > t0 ? (lane - t0 < 8u ? lane - t0 : 8u) : 0u;   // staged slots of this group in front of the lane
                const uint32_t valid = (act && lane < amb_at) ? (grp & ((1u << ahead) - 1u)) : 0u;
                uint32_t bits = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float* nf = sh->newn32[(t0 + (uint32_t)t) & 63u];
                    float sp = nf[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sp = __builtin_fmaf(nf[k], Qf[k], sp);
                    screen_bit(bits, sp, thr_c);
                }
                const uint32_t lookm = rev8(bits) & valid;
                if (__ballot(lookm != 0) != 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (__ballot((lookm >> t) & 1u) != 0) {
                            const uint32_t sl = (t0 + (uint32_t)t) & 63u;
                            double ca[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) ca[k] = sh->newn[k][sl];
                            if (((lookm >> t) & 1u) != 0 && hi32(dist2<D>(ca, q, DIM)) <= hb) confl |= 1ull << sl;
                        }
                    }
                }
            }
            while (true) {
                const uint64_t dirtym = __ballot((confl & newm) != 0);   // (bits at or beyond a lane, or of lanes past amb_at, are never set)
                if (dirtym == 0) break;
                const uint32_t c = (uint32_t)(__ffsll((unsigned long long)dirtym) - 1);
                if (STAMP) ++n_cut_conflict;
                const uint64_t candm = uni64(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((confl & newm) >> 32), (int)c) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(confl & newm), (int)c));
                double qc[D];
#pragma unroll
                for (int k = 0; k < D; ++k) qc[k] = unid(sh->q[k][(jr + c) & 63u]);
                double best_d = sqrt(readlane_f64(g, (int)c));   // the old nearest node: the lowest index, so it keeps every tie
                uint32_t best = 64u;
                for (uint64_t rem = candm; rem != 0; rem &= rem - 1) {
                    const uint32_t i = (uint32_t)(__ffsll((unsigned long long)rem) - 1);
                    double ca[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) ca[k] = unid(sh->newn[k][i]);
                    const double d = sqrt(dist2<D>(ca, qc, DIM));
                    if (d < best_d) { best_d = d; best = i; }
                }
                if (best != 64u) {   // a node of this round is the query's nearest: everything that follows from it, again
                    double q_near1[D], qn1[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) q_near1[k] = unid(sh->newn[k][best]);
                    const uint32_t nearest1 = n + (uint32_t)__popcll(okm & below_mask(best));
                    const double g1 = dist2<D>(q_near1, qc, DIM);
                    steer<DIM>(p, false, g1, q_near1, qc, qn1);
                    bool ok1 = true;
                    if (nobs > 0) {
                        double mid1[D];
                        lerp<DIM>(q_near1, qn1, 0.5, mid1, DIM);
                        double oc[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) oc[k] = shared.obs[k][lane];
                        const double othr = shared.obs[D][lane], ofilt = shared.obs[D + 1][lane];
                        if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid1)) != 0 || extras)
                            ok1 = motion_lanes<DIM>(p, lane, q_near1, qn1, oc, othr, ofilt, ns64);
                    }
                    const bool dup1 = g1 == 0.0;
                    const bool hit1 = ok1 && dist2<D>(qn1, goal_c, DIM) <= goal_thr;
                    float na1[D], ncc1;
                    {
                        double sq = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            na1[k] = (float)(qn1[k] - c0[k]);
                            sq += (double)na1[k] * (double)na1[k];
                        }
                        ncc1 = (float)sq;
                    }
                    if (lane == c) {
                        nearest = nearest1;
                        g = g1;
                        ok = ok1;
                        dup = dup1;
                        hit = hit1;
#pragma unroll
                        for (int k = 0; k < D; ++k) { qn[k] = qn1[k]; q_near[k] = q_near1[k]; nf_a[k] = na1[k]; }
                        nf_cc = ncc1;
                        if (ok1 && !dup1) {
#pragma unroll
                            for (int k = 0; k < D; ++k) { sh->newn32[c][k] = na1[k]; sh->newn[k][c] = qn1[k]; }
                            sh->newn32[c][D] = ncc1;
                        }
                    }
                    const uint64_t bitc = 1ull << c;
                    okm = ok1 ? (okm | bitc) : (okm & ~bitc);
                    newm = (ok1 && !dup1) ? (newm | bitc) : (newm & ~bitc);
                    // the queries behind it against its new node
                    confl &= ~bitc;
                    if (ok1 && !dup1) {
                        float sp = ncc1;
#pragma unroll
                        for (int k = 0; k < D; ++k) sp = __builtin_fmaf(na1[k], Qf[k], sp);
                        const bool look = act && lane > c && lane < amb_at && !(sp > thr_c);
                        if (look && hi32(dist2<D>(qn1, q, DIM)) <= hb) confl |= bitc;
                    }
                }
                if (lane == c) confl = 0;
            }
            hitm = __ballot(hit);
            // node cap: query j is processed only while the tree has room (checked before any draw of the iteration)
            const uint64_t capm = __ballot(act && n + (uint32_t)__popcll(okm & below_mask(lane)) >= p.max_nodes);
            if (capm != 0) {
                const uint32_t c = (uint32_t)(__ffsll((unsigned long long)capm) - 1);
                if (c <= cut) { cut = c; stop_after = 2; }
            }
            if (p.stop_at_goal && hitm != 0) {
                const uint32_t c = (uint32_t)__ffsll((unsigned long long)hitm);   // first hit lane + 1
                if (c <= cut) { cut = c; stop_after = 0; }
            }
        }
        if ((p.dbg_flags & OXHIP_DEBUG_ONE_LANE_ROUNDS) != 0 && cut > 1) { cut = 1; stop_after = -1; if (STAMP) ++n_forced; }
        OXHIP_CPHASE(4);   // prefix: cap, goal, conflicts
        // ---- commit lanes [0, cut) in query order
        if (STAMP) t_cm = (uint64_t)clock64();
        if (cut > 0) {
            const uint64_t cutm = first_n_mask(cut);
            const bool mine = lane < cut;
            uint32_t ins_cell = 0, ins_slot = 0;
            uint64_t ins_entry = 0;
            bool ins_link = false;
            if (ins) {
                const uint32_t idx = n + (uint32_t)__popcll(okm & below_mask(lane));
                if (mine && ok) {
                    // insert (rrt.rs:213-217): the tree, and -- unless it repeats its nearest node's position, which the strict
                    // '<' of rrt.rs:192 can never prefer -- its cell's list
#pragma unroll
                    for (int k = 0; k < D; ++k) tree[(size_t)k * cap + idx] = qn[k];
                    xyz[idx] = cdouble4{qn[0], qn[1], D >= 3 ? qn[D - 1] : 0.0, 0.0};
                    parent[idx] = (int32_t)nearest;
                    skip[idx] = dup ? 1 : 0;
                }
                {
                    float tf[3];
                    double err;
                    uint64_t entry;
                    const uint32_t cell = cell_place<DIM>(grid, qn, idx, entry, tf, err);
                    const bool link = mine && ok && !dup;
                    if (grid.level == 0) {
                        if (mine && ok && idx < kFlatCap)
                            flat[idx] = link ? cfloat4{tf[0], tf[1], tf[2], 0.0f} : cfloat4{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f};
                    } else {
                        // (the slot comes back from the L2 while the reductions and the checksum below run; the entry is stored after them)
                        ins_cell = cell; ins_entry = entry; ins_link = link;
                        ins_slot = cells_insert_begin(blk, cell, link);
                    }
                    const double e = -wave_min_f64(link ? -err : 0.0);
                    if (e > grid.delta_node) grid.delta_node = (double)f32_up(e * (1.0 + 1e-9));
                    uint32_t mab = 0;
#pragma unroll
                    for (int k = 0; k < D; ++k) { const uint32_t ab = lf32_bits(nf_a[k]) & 0x7FFFFFFFu; mab = ab > mab ? ab : mab; }
                    mab = wave_max_u32(mine && ok ? mab : 0u);
                    mabs_bits = mab > mabs_bits ? mab : mabs_bits;
                }
                // goal test (rrt.rs:220-223): the first hit in query order
                const uint64_t hits = hitm & cutm;
                if (hits != 0 && st.goal_node < 0)
                    st.goal_node = (int32_t)__builtin_amdgcn_readlane((int)idx, __ffsll((unsigned long long)hits) - 1);
                n += (uint32_t)__popcll(okm & cutm);
            }
            if (STAMP) { const uint64_t nw = (uint64_t)clock64(); t_c[2] += nw - t_cm; t_cm = nw; }
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
            if (ins && grid.level != 0) {
                if (STAMP) { const uint64_t nw = (uint64_t)clock64(); t_c[0] += nw - t_cm; t_cm = nw; }
                cells_insert_finish(blk, grid.pool_next, ins_cell, ins_entry, ins_link, ins_slot, lane);
                if (STAMP) { const uint64_t nw = (uint64_t)clock64(); t_c[1] += nw - t_cm; t_cm = nw; }
            }
            st.iterations += cut;
            st.accepted += (uint64_t)__popcll(okm & cutm);
            draws_done = uni64((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_l, (int)(cut - 1u)) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_l >> 32), (int)(cut - 1u)) << 32));
            jr += cut;
        }
        if (STAMP) { const uint64_t nw = (uint64_t)clock64(); t_c[3] += nw - t_cm; t_cm = nw; }
        OXHIP_CPHASE(5);   // commit
        if (stop_after >= 0) { stop = stop_after; break; }
        if (cut == m || ambm == 0 || (uint32_t)(__ffsll((unsigned long long)ambm) - 1) != cut) continue;

        // ---- the lane at `cut` is ambiguous: that one query over the WHOLE binary64 tree (the path of rrt_lanes.hip).  First
        //      by squared distances, the wave striding over the nodes: if exactly one node is within a rounding of the
        //      minimum, it is the reference's nearest node (sqrt is monotone).  Only a genuine near-tie -- two d2 that may
        //      share a correctly rounded root -- takes the reference's literal loop (post-sqrt compare, lowest index).
        if (!FROZEN && n >= p.max_nodes) { stop = 2; break; }
        {
            if (STAMP) ++n_amb;
            const uint32_t slot1 = jr & 63u;
            double q1[D];
#pragma unroll
            for (int k = 0; k < D; ++k) q1[k] = unid(sh->q[k][slot1]);
            bool same_q = memo_n == n;
#pragma unroll
            for (int k = 0; k < D; ++k) same_q = same_q && __double_as_longlong(q1[k]) == __double_as_longlong(memo_q[k]);
            double gmin = memo_g;
            uint32_t memo_hit_idx = memo_idx;
            bool tie = false;
            if (!same_q) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores to the tree have landed (same CU)
                Scan ps{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
                constexpr int WT = OXHIP_CELLS_WT_UNROLL;   // chunks of 256 nodes in flight (4 nodes per lane each)
                for (uint32_t i0 = 4u * lane; i0 < n; i0 += 256u * (uint32_t)WT) {
                    uint32_t sk4[WT], il[WT];
                    double d16[WT][4];
#pragma unroll
                    for (int t = 0; t < WT; ++t) {
                        const uint32_t ib = i0 + 256u * (uint32_t)t;
                        il[t] = ib < n ? ib : 0u;   // (rows are padded to cap >= n rounded up to 1024)
                        sk4[t] = *reinterpret_cast<const uint32_t*>(skip + il[t]);
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        ldouble4 ck[WT];
#pragma unroll
                        for (int t = 0; t < WT; ++t) ck[t] = *reinterpret_cast<const ldouble4*>(tree + (size_t)k * cap + il[t]);
#pragma unroll
                        for (int t = 0; t < WT; ++t) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double df = ck[t][r] - q1[k];
                                const double sq = df * df;
                                d16[t][r] = k == 0 ? sq : d16[t][r] + sq;   // the reference's summation order
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < WT; ++t) {
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
                double oc[D];
#pragma unroll
                for (int k = 0; k < D; ++k) oc[k] = shared.obs[k][lane];
                const double othr = shared.obs[D][lane], ofilt = shared.obs[D + 1][lane];
                if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid1)) != 0 || extras)
                    ok1 = motion_lanes<DIM>(p, lane, q_near1, qn1, oc, othr, ofilt, ns64);
            }
            st.checksum = uni64(chk_push(st.checksum, iter_digest<D>(nearest1, qn1, DIM, ok1)));
            st.iterations++;
            draws_done = uni64(sh->pos_after[slot1]);
            bool hit1 = false;
            if (ok1) {
                st.accepted++;
                if (!FROZEN) {
                    const uint32_t i = n;
                    float tf[3];
                    double err;
                    uint64_t entry;
                    const uint32_t cell = cell_place<DIM>(grid, qn1, i, entry, tf, err);
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < D; ++k) tree[(size_t)k * cap + i] = qn1[k];
                        xyz[i] = cdouble4{qn1[0], qn1[1], D >= 3 ? qn1[D - 1] : 0.0, 0.0};
                        parent[i] = (int32_t)nearest1;
                        skip[i] = dup1 ? 1 : 0;
                        if (grid.level == 0 && i < kFlatCap)
                            flat[i] = dup1 ? cfloat4{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.0f} : cfloat4{tf[0], tf[1], tf[2], 0.0f};
                    }
                    if (grid.level != 0) cells_insert(blk, grid.pool_next, cell, entry, lane == 0 && !dup1, lane);
                    if (!dup1 && err > grid.delta_node) grid.delta_node = (double)f32_up(err * (1.0 + 1e-9));
                    {
                        uint32_t mab = 0;
#pragma unroll
                        for (int k = 0; k < D; ++k) { const uint32_t ab = lf32_bits((float)(qn1[k] - c0[k])) & 0x7FFFFFFFu; mab = ab > mab ? ab : mab; }
                        mabs_bits = mab > mabs_bits ? mab : mabs_bits;
                    }
                    ++n;
                    if (dist2<D>(qn1, goal_c, DIM) <= goal_thr) {
                        if (st.goal_node < 0) st.goal_node = (int32_t)i;
                        hit1 = true;
                    }
                }
            }
            jr += 1;
            if (hit1 && p.stop_at_goal) { stop = 0; break; }
        }
    }
#undef OXHIP_CPHASE
    if (!FROZEN) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // (a frozen launch has stored nothing)
    if (split > 1) {
        // this part's sums into the problem's accumulator; the last part to arrive writes the state.
        // H_final = H0 P^N + sum_parts Hpart P^(N - j_hi)
        uint64_t w = 1, base = kFnvPrime;
        for (uint32_t e = budget_all - j_hi; e != 0; e >>= 1) { if (e & 1u) w *= base; base *= base; }
        CellAcc& acc = p.cell_acc[prob];
        uint32_t arrived = 0;
        if (lane == 0) {
            // No cache flush here (a device-scope fence writes back and invalidates the XCD's L2 -- per part, that cost more than
            // the part's work at high splits): the sums travel by atomics, which execute at the coherence point, and the arrival
            // counter is bumped only after their return values are back -- the empty asm makes the increment depend on them.
            const unsigned long long r0 = atomicAdd((unsigned long long*)&acc.chk, (unsigned long long)(st.checksum * w));
            const unsigned long long r1 = atomicAdd((unsigned long long*)&acc.accepted, (unsigned long long)st.accepted);
            const unsigned long long r2 = part == split - 1u ? atomicExch((unsigned long long*)&acc.pos, (unsigned long long)draws_done) : 0ull;
            uint32_t one = 1u;
            asm volatile("" : "+v"(one) : "v"((uint32_t)r0), "v"((uint32_t)r1), "v"((uint32_t)r2));
            arrived = atomicAdd(&acc.done, one);
        }
        arrived = uni(arrived);
        if (arrived == split - 1u && lane == 0) {
            uint64_t pn = 1, b2 = kFnvPrime;
            for (uint32_t e = budget_all; e != 0; e >>= 1) { if (e & 1u) pn *= b2; b2 *= b2; }
            ProblemState out = st0;
            out.checksum = st0.checksum * pn + (uint64_t)atomicAdd((unsigned long long*)&acc.chk, 0ull);
            out.accepted = st0.accepted + (uint64_t)atomicAdd((unsigned long long*)&acc.accepted, 0ull);
            out.iterations = st0.iterations + budget_all;
            out.draws = (uint64_t)atomicAdd((unsigned long long*)&acc.pos, 0ull);
            out.stop_reason = 1;
            p.state[prob] = out;
            acc.chk = 0; acc.accepted = 0; acc.pos = 0; acc.done = 0;
        }
    } else if (lane == 0) {
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
    }
    if (!FROZEN) grid_store<DIM>(meta, grid, n, mabs_bits, lane);
    if (STAMP && p.dbg && lane == 0) {
        atomicAdd((unsigned long long*)&p.dbg[54], (unsigned long long)n_amb);
        atomicAdd((unsigned long long*)&p.dbg[55], (unsigned long long)n_memo);
        atomicAdd((unsigned long long*)&p.dbg[56], (unsigned long long)n_cut_conflict);
        atomicAdd((unsigned long long*)&p.dbg[59], (unsigned long long)n_forced);
        atomicAdd((unsigned long long*)&p.dbg[60], (unsigned long long)n_expand);
        atomicAdd((unsigned long long*)&p.dbg[61], (unsigned long long)n_regrid);
        atomicAdd((unsigned long long*)&p.dbg[45], (unsigned long long)n_amb);
        if (prob == 0) {   // problem 0, summed over the parts of a split launch (and over the launches since enable_stamps)
            const uint64_t vals[8] = {n_amb, n_rounds, n_lanes, st.iterations, n_expand, n_steps, n_regrid, n_memo};
            const int idx[8] = {4, 5, 6, 7, 8, 9, 10, 11};
            for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&p.dbg[idx[i]], (unsigned long long)vals[i]);
            atomicAdd((unsigned long long*)&p.dbg[12], (unsigned long long)n_cut_conflict);
            atomicAdd((unsigned long long*)&p.dbg[15], (unsigned long long)n_tie);
            for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&p.dbg[32 + i], (unsigned long long)t_ph[i]);
            atomicAdd((unsigned long long*)&p.dbg[13], (unsigned long long)((uint64_t)clock64() - t_begin));
            p.dbg[62] = t_c[0]; p.dbg[63] = t_c[1]; p.dbg[49] = t_c[2]; p.dbg[0] = t_c[3];
            atomicAdd((unsigned long long*)&p.dbg[1], (unsigned long long)n_tail);
            atomicAdd((unsigned long long*)&p.dbg[2], (unsigned long long)n_tail_pairs);
            for (int i = 0; i < 8; ++i) {   // the neighbour trips of a round, by trip number: lanes still asking, cells asked for, trips
                atomicAdd((unsigned long long*)&p.dbg[16 + i], (unsigned long long)h_lanes[i]);
                atomicAdd((unsigned long long*)&p.dbg[24 + i], (unsigned long long)h_trips[i]);
            }
            const int ci[8] = {40, 41, 42, 43, 44, 46, 47, 48};
            for (int i = 0; i < 8; ++i) atomicAdd((unsigned long long*)&p.dbg[ci[i]], (unsigned long long)h_cells[i]);
        }
    }
}

// Before every launch of rrt_cells_kernel, one wave per problem: (re)build the grid when it does not cover the tree as it
// is (first launch, after setup / set_tree), and -- for a split frozen launch -- run the sampler's position arithmetic over
// the whole budget to learn the stream position at which each part starts (rrt.rs:177-184: a goal sample takes one word,
// a uniform sample 1 + D; nothing else of the draws is evaluated).
template <int DIM>
__global__ __launch_bounds__(64) void cells_prepare_kernel(DevParams p) {
    __shared__ CellsWaveLds<DIM> shw;
    const uint32_t prob = blockIdx.x, lane = threadIdx.x;
    const ProblemState st0 = p.state[prob];
    if (p.stop_at_goal && st0.goal_node >= 0) return;
    CellMeta& meta = p.cell_meta[prob];
    const uint32_t n = st0.n_nodes;
    double c0[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) c0[k] = 0.5 * p.lo[k] + 0.5 * p.hi[k];
    const bool valid = uni(meta.valid) != 0 && uni(meta.n_grid) == n &&
                       (p.freeze || n < uni(meta.regrid_at)) && uni(meta.level) == cells_level(n, DIM, p.cell_level_max);
    if (!valid) {
        CellGrid g;
        uint32_t mabs_bits;
        cells_build<DIM>(p, prob, n, lane, c0, g, mabs_bits);
        grid_store<DIM>(meta, g, n, mabs_bits, lane);
    }
    const uint32_t split = p.freeze ? p.cells_split : 1u;
    if (split <= kSelfSkipMax) return;   // (few parts find their own starts: rrt_cells_kernel)
    double goal_c[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    RngWindow rng;
    rng.init(shw.rng_buf, p.seed, p.first_problem_id + prob, st0.draws);
    const uint32_t budget = (uint32_t)p.budget, rounds = (budget + 63u) / 64u;
    uint32_t next_part = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        while (next_part < split && cells_part_begin(rounds, next_part, split) == r) {
            if (lane == 0) p.cell_part_pos[(size_t)prob * kMaxSplit + next_part] = rng.pos;
            ++next_part;
        }
        if (next_part >= split) break;
        const uint32_t m = budget - r * 64u < 64u ? budget - r * 64u : 64u;
        cells_sample_block<DIM, false>(rng, p, goal_c, p.goal_r[prob], m, lane, &shw, r * 64u);
    }
    while (next_part < split) {   // (parts without a round start where the stream ends)
        if (lane == 0) p.cell_part_pos[(size_t)prob * kMaxSplit + next_part] = rng.pos;
        ++next_part;
    }
}

// DevParams::sph_grid / star_sph_grid: one thread per cell of the bounds' grid; bit j = sphere j's filter ball (every midpoint within
// sqrt(filt[j]) of the centre) reaches the cell's box, taken 2^-9 of a cell wider on every side.  (NaN / +inf thresholds: always set.)
__global__ __launch_bounds__(256) void sphere_grid_kernel(DevParams p, uint64_t* grid, const double* filt) {
    const uint32_t G = p.sph_grid_G, dim = p.dim;
    const uint32_t cells = dim == 2 ? G * G : G * G * G;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= cells) return;
    const uint32_t ic[3] = {idx % G, (idx / G) % G, idx / (G * G)};
    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
    uint64_t mask = 0;
    for (uint32_t j = 0; j < ns64; ++j) {
        double d2 = 0.0;
        for (uint32_t k = 0; k < dim; ++k) {
            const double w = (p.hi[k] - p.lo[k]) / (double)G;
            const double blo = p.lo[k] + ((double)ic[k] - 0x1p-9) * w, bhi = p.lo[k] + ((double)ic[k] + 1.0 + 0x1p-9) * w;
            const double c = p.sph_c[(size_t)k * p.n_spheres + j];
            const double e = fmax(fmax(blo - c, c - bhi), 0.0);
            d2 += e * e;
        }
        if (!(d2 * (1.0 - 1e-9) > filt[j])) mask |= 1ull << j;
    }
    grid[idx] = mask;
}
uint32_t sphere_grid_side(uint32_t dim) { return dim == 2 ? 128u : 32u; }
void launch_sphere_grid(const DevParams& p, uint64_t* grid, const double* filt, hipStream_t stream) {
    const uint32_t G = p.sph_grid_G, cells = p.dim == 2 ? G * G : G * G * G;
    hipLaunchKernelGGL(sphere_grid_kernel, dim3((cells + 255u) / 256u), dim3(256), 0, stream, p, grid, filt);
}

bool cells_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && cap <= 65535u; }   // (a block entry names its node with 16 bits)

// grid level a tree of `cap` nodes can reach: the block array is sized for it
uint32_t cells_level_max(uint32_t dim, uint32_t cap) {
    const uint32_t lim = dim == 2 ? 14u : 10u;   // at most 128^2 / 32^3 cells
    uint32_t l = 1;
    while (l < lim && cells_level_cap(l, (int)dim) < cap) ++l;
    return l;
}
// head blocks the finest grid of a capacity needs
uint32_t cells_head_blocks(uint32_t dim, uint32_t cap) {
    const uint32_t g = cells_G(cells_level_max(dim, cap));
    return dim == 2 ? g * g : g * g * g;
}

template <int DIM>
static void launch_cells_dim(const DevParams& p, hipStream_t stream) {
    const uint32_t split = p.freeze ? p.cells_split : 1u;
    // the prepare pass builds missing grids and hands many-part launches their stream positions; a launch that needs neither skips it
    if (!(p.cells_meta_ok != 0 && split <= kSelfSkipMax))
        hipLaunchKernelGGL((cells_prepare_kernel<DIM>), dim3(p.n_problems), dim3(64), 0, stream, p);
    const uint32_t per_xcd = (p.n_problems + 7u) / 8u;
    const uint32_t wgs_per_xcd = (per_xcd * split + (uint32_t)kCellsWaves - 1u) / (uint32_t)kCellsWaves;
    dim3 grid(8u * wgs_per_xcd), block(kCellsWaves * 64);
    if (p.freeze) {
        if (p.dbg) hipLaunchKernelGGL((rrt_cells_kernel<DIM, true, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((rrt_cells_kernel<DIM, false, true>), grid, block, 0, stream, p);
    } else {
        if (p.dbg) hipLaunchKernelGGL((rrt_cells_kernel<DIM, true, false>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((rrt_cells_kernel<DIM, false, false>), grid, block, 0, stream, p);
    }
}

void launch_rrt_cells(const DevParams& p, hipStream_t stream) {
    if (p.dim == 2) launch_cells_dim<2>(p, stream);
    else if (p.dim == 3) launch_cells_dim<3>(p, stream);
}

}  // namespace oxhip
