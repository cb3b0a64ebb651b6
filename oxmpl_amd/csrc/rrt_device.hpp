// rrt_device.hpp -- device functions shared by the RRT kernels (gfx950, wave64).
//
// Every function restates one piece of the reference's hot path in IEEE binary64 with the
// reference's evaluation order; the translation unit is built with -ffp-contract=off so
// that `from + (to - from) * t` keeps its three roundings (the reference, Rust, never fuses).
//
// Reference lines (relative to /root/reference):
//   distance      oxmpl/src/base/spaces/real_vector_state_space.rs:137-155
//   interpolate   oxmpl/src/base/spaces/real_vector_state_space.rs:161-186
//   sample        oxmpl/src/base/spaces/real_vector_state_space.rs:233-249, rrt.rs:177-184
//   nearest       oxmpl/src/geometric/planners/rrt.rs:187-196
//   steer         oxmpl/src/geometric/planners/rrt.rs:199-208
//   check_motion  oxmpl/src/geometric/planners/rrt.rs:90-116
//   RNG           rand 0.9.1 random_bool / random_range(f64), rand_chacha 0.9.0 ChaCha12Rng
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/oxmpl_hip.h"   // the enums the kernels share with the boundary (debug flags, goal sampler)
#include "ox_sincos.hpp"

namespace oxhip {

constexpr int kMaxDim = 8;
constexpr uint64_t kFnvPrime = 0x100000001B3ull;
constexpr uint64_t kFnvBasis = 0xCBF29CE484222325ull;

// per-problem planner state that persists in HBM between launches (resume / chunked solve)
struct ProblemState {
    uint64_t iterations;
    uint64_t accepted;
    uint64_t checksum;
    uint64_t draws;       // u64 words consumed from the problem's ChaCha12 stream
    uint32_t n_nodes;     // RRT: tree size; RRTConnect: start-tree size
    int32_t goal_node;    // RRT: first node satisfying the goal; RRTConnect: last start-tree node of the solution; -1 = none
    int32_t stop_reason;  // oxhip_stop_reason of the last launch
    uint32_t n_nodes_b;   // RRTConnect: goal-tree size
    int32_t goal_node_b;  // RRTConnect: last goal-tree node of the solution (-1: the start tree reached the goal itself)
    uint32_t pad;
};

// one neighbour of one node (rrt_star_wire.hip): written as (j, owner node i, d2) by the pair search -- the edge kernel reads the
// owner from `flags` -- and completed by the edge kernel (validity bits, distance)
struct StarEntry {
    uint32_t j;        // the neighbour's index (< the node's own)
    uint32_t flags;    // bit 0: check_motion(neighbour, node) holds; bit 1: check_motion(node, neighbour) holds
    double d;          // distance(node, neighbour)  (the pair search leaves the squared distance here)
};

// 32 neighbours of one node as the counting pass found them (rrt_star_wire.hip): the node's list is its chunks in ordinal order
struct StarChunk {
    uint32_t i, cnt, seq, pad;   // owner node, entries used, ordinal within the node's list
    uint32_t j[32];              // ascending neighbour indices
};

// rrt_cells.hip: a problem's cell grid (HBM, persists between launches) and the accumulator of a split frozen launch
struct CellMeta {
    double lo[3];          // grid origin: the corner of the box around the bounds, the goal centre and the tree
    double inv_h;          // 1 / cell size (cubic cells)
    uint32_t G[3];         // cells per axis
    uint32_t level;        // G = 2^level along the longest side (0: no grid, every node by index)
    uint32_t n_grid;       // tree nodes the grid covers
    uint32_t regrid_at;    // tree size at which the next finer grid takes over
    uint32_t valid;        // 0 after setup / set_tree
    float delta_node;      // bound on |stored - true| of any node coordinate, cell units
    uint32_t mabs_bits;    // bits of the largest |fl32(coordinate - c0)| over the tree
    uint32_t pool_next;    // next free overflow block
};
// one cell of the grid: 64 bytes = one memory sector.  Head blocks are indexed by cell; a cell with more than seven nodes
// chains overflow blocks (allocated after the head blocks).  Entry = x | y << 16 | z << 32 | node << 48: the node's position
// inside the cell in 2^-16 cell units (decoded at the bin centre) and its index in the tree.
struct alignas(64) CellBlock {
    uint32_t count;   // head block: nodes filed in this cell (all its blocks)
    uint32_t next;    // next block of the cell's chain (0: none)
    uint64_t e[7];
};
struct CellAcc {
    uint64_t chk, accepted, pos;
    uint32_t done, pad;
};

// kernel arguments (by value)
struct DevParams {
    uint32_t dim, n_problems, cap, max_nodes;
    double lo[kMaxDim], hi[kMaxDim], scale[kMaxDim];  // scale = hi - lo (random_range)
    double max_distance;
    double res;          // get_longest_valid_segment_length() * 0.1   (rrt.rs:97)
    uint64_t p_int;      // Bernoulli::new(goal_bias)
    uint64_t seed, first_problem_id;
    uint64_t budget;     // iterations this launch may run per problem
    uint32_t stop_at_goal, freeze;
    uint32_t n_spheres, n_boxes;
    const double* sph_c;    // [dim][n_spheres]
    const double* sph_thr;  // [n_spheres]  valid iff d2 > thr  (== sqrt(d2) > radius, exactly)
    const double* box_lo;   // [dim][n_boxes]
    const double* box_hi;   // [dim][n_boxes]
    double* tree;           // [P][dim][cap]  SoA per problem
    int32_t* parent;        // [P][cap]
    double* tree_b;         // RRTConnect goal trees, same layout (null for RRT)
    int32_t* parent_b;
    uint8_t* skip;          // [P][cap] 1 = coordinate duplicate of a lower-index node: never nearest (resident kernel)
    ProblemState* state;    // [P]
    const double* goal_c;   // [P][dim]
    const double* goal_thr; // [P]  satisfied iff d2 <= thr (== sqrt(d2) <= radius, exactly)
    double t_steer;         // largest d2 with sqrt(d2) <= max_distance: steer iff d2 > t_steer (exact)
    const double* sph_filt; // [n_spheres] conservative filter: d2(centre, segment midpoint) > filt => sphere cannot be hit
    uint64_t* dbg;          // optional [16] cycle stamps of workgroup 0 (diagnostic build of the resident kernel)
    // SE(2) only (rrt_connect_se2.hip): segment-soup checker
    const double* segs;     // [n_segs][4] (ax, ay, bx, by)
    uint32_t n_segs, space; // space: oxhip_space_kind
    double seg_thr;         // valid iff d2(point, segment) > seg_thr  (== distance > clearance, exactly)
    const uint16_t* seg_grid;   // [seg_grid_G^2][8] the segments that can reach each cell of the (x, y) bounds' grid (null: every state meets every segment)
    uint32_t seg_grid_G, seg_grid_pad;
    double seg_grid_inv[2]; // cells per unit length along x, y
    uint32_t adv_steps, adv_pad;   // RRTConnect: ceil(max_distance / res), check_motion's step count for a motion of length max_distance (0: not usable)
    double adv_slack;   // how far a distance may be from max_distance without changing that count (half the gap to the nearer integer, times res)
    // RRT* only (rrt_star.hip)
    double* cost;           // [P][cap] cost-to-come of every node (Node::cost, rrt_star.rs:26)
    uint32_t* nb_idx;       // [P][cap] scratch: find_neighbours' result of the current iteration
    double* nb_dist;        // [P][cap] scratch: distance(q_new, neighbour)
    double thr_search;      // largest d2 with sqrt(d2) < search_radius (strict, rrt_star.rs:125)
    uint64_t* wire_chk;     // [P] W, the wiring polynomial of RRT*'s checksum (ProblemState::checksum holds H; reported: H + W)
    // RRT*, decoupled design (rrt_star_wire.hip): geometry by the RRT kernel, then neighbour lists, edge validity, wiring
    uint32_t* wired;        // [P] nodes whose parent and cost are final (the wiring cursor; node 0 is wired by setup)
    uint32_t* nbr_cnt;      // [P][cap] |find_neighbours(node)| among the nodes before it
    uint32_t* nbr_off;      // [P][cap] where the node's list starts in the problem's pool segment (this round)
    uint32_t* nbr_take;     // [P] nodes wired this round: the longest prefix of the pending nodes whose lists fit the segment
    uint32_t* nbr_total;    // [P] neighbour pairs of those nodes (the used part of the pool segment)
    double* d_near;         // [P][cap] distance(node, its nearest node)
    struct StarEntry* pool; // [P][pool_share] neighbour lists, ascending index within a list
    uint32_t pool_share;
    struct StarChunk* chunks; // [P][chunk_share]
    uint32_t* chunk_cursor;   // [P] chunks handed out this round (may run past chunk_share: then the round takes the two-pass path)
    uint32_t chunk_share;
    uint32_t seg_index, seg_count;   // star_edges / star_wire: this launch's segment of the round (entries resp. the nodes whose lists end in them)
    // PRM only (prm_kernels.hip): the midpoint filter's inputs for motions of any length
    const double* sph_r;    // [n_spheres] radii as given
    double filt_abs;        // 1e-9 * largest coordinate magnitude in play (absolute rounding margin)
    // binary32 screen of the streaming kernels (rrt_stream.hip, rrt_star.hip): fl32 shadow of the tree
    float* tree32;          // [P][dim][cap] fl32(tree), same layout; maintained by the kernels that insert
    uint32_t* shadow_state; // [P][2]: nodes whose shadow is valid; bits of the largest |fl32(coordinate)| among them
    uint32_t dbg_flags;     // oxhip_rrt_config.debug_flags (oxhip_debug_flag bits): test-only switches, results identical
    uint32_t goal_sampler;  // oxhip_goal_sampler
    const double* goal_r;   // [P] goal radii as given (the disc sampler scales by them)
    // rrt_cells.hip
    CellBlock* cell_blk;     // [P][cell_blocks]: head blocks (one per cell of the finest grid), then the overflow blocks
    float* cell_flat;        // [P][4096][4]: small trees (up to 1,024 nodes) as a flat list (tx, ty, tz, -): positions in cell units
    double* cell_xyz;        // [P][cap][4]: the tree's binary64 coordinates once more, node-major (one 32-byte access per winner)
    CellMeta* cell_meta;     // [P]
    CellAcc* cell_acc;       // [P] zero between launches
    uint64_t* cell_part_pos; // [P][8] stream position at which each part of a split frozen launch starts
    uint32_t cell_blocks, cell_level_max, cells_split;
    uint32_t cells_meta_ok;  // host: every problem's grid record was written by a launch since the last setup / set_tree (no prepare pass needed)
    // the spheres a motion's midpoint can meet, looked up instead of screened: bit j of sph_grid[cell] is set when sphere j's filter
    // ball (sph_filt) reaches the cell's box (sph_grid_G cells along every axis of the bounds, dimensions 2 / 3)
    const uint64_t* sph_grid;
    uint32_t sph_grid_G;
    const uint64_t* star_sph_grid;   // the same for RRT*'s edge checks: filter balls of motions up to search_radius long (null: none)
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double unid(double v) { return __longlong_as_double((long long)uni64((uint64_t)__double_as_longlong(v))); }

// ------------------------------------------------------------------ ChaCha12 (rand_chacha)
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define OXHIP_QR(a, b, c, d)               \
    a += b; d ^= a; d = rotl32(d, 16);     \
    c += d; b ^= c; b = rotl32(b, 12);     \
    a += b; d ^= a; d = rotl32(d, 8);      \
    c += d; b ^= c; b = rotl32(b, 7);

// one 16-word block; key = LE(seed) || 0^24, 64-bit block counter, 64-bit stream id
__device__ inline void chacha12_block(uint64_t seed, uint64_t counter, uint64_t stream, uint32_t out[16]) {
    const uint32_t s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32);
    const uint32_t t0 = (uint32_t)stream, t1 = (uint32_t)(stream >> 32);
    uint32_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, x4 = k0, x5 = k1, x6 = 0, x7 = 0;
    uint32_t x8 = 0, x9 = 0, x10 = 0, x11 = 0, x12 = c0, x13 = c1, x14 = t0, x15 = t1;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        OXHIP_QR(x0, x4, x8, x12) OXHIP_QR(x1, x5, x9, x13) OXHIP_QR(x2, x6, x10, x14) OXHIP_QR(x3, x7, x11, x15)
        OXHIP_QR(x0, x5, x10, x15) OXHIP_QR(x1, x6, x11, x12) OXHIP_QR(x2, x7, x8, x13) OXHIP_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + s0; out[1] = x1 + s1; out[2] = x2 + s2; out[3] = x3 + s3;
    out[4] = x4 + k0; out[5] = x5 + k1; out[6] = x6; out[7] = x7;
    out[8] = x8; out[9] = x9; out[10] = x10; out[11] = x11;
    out[12] = x12 + c0; out[13] = x13 + c1; out[14] = x14 + t0; out[15] = x15 + t1;
}

// Workgroup-shared window of 64 consecutive blocks (512 u64 words) of one problem's stream,
// stored word-major ([word][block]) so the refill's 64 lanes write conflict-free.
// All threads of the workgroup call next() in the same (uniform) sequence.
struct RngWindow {
    uint32_t (*buf)[64];   // LDS, [16][64]
    uint64_t seed, stream;
    uint64_t pos;          // absolute u64-word position in the stream (uniform)
    uint64_t base_blk;     // first block held in buf (uniform)

    __device__ __forceinline__ void init(uint32_t (*lds)[64], uint64_t seed_, uint64_t stream_, uint64_t pos_) {
        buf = lds; seed = seed_; stream = stream_; pos = pos_;
        base_blk = (pos_ >> 3) - 64;  // forces a refill on first use (modular arithmetic)
    }
    // WG_SYNC = true: every thread of the workgroup calls next() in the same sequence (barriers
    // around the refill).  WG_SYNC = false: the window is private to wave 0 (LDS is in-order per wave).
    template <bool WG_SYNC = true>
    __device__ __forceinline__ uint64_t next() {
        uint64_t blk = uni64(pos >> 3);
        if (blk - base_blk >= 64) {  // workgroup-uniform
            if (WG_SYNC) __syncthreads();  // everybody is done reading the old window
            base_blk = blk;
            if (!WG_SYNC || threadIdx.x < 64) {  // WG_SYNC: wave 0 refills; otherwise the owning wave does
                const uint32_t bl = threadIdx.x & 63;
                uint32_t o[16];
                chacha12_block(seed, blk + bl, stream, o);
#pragma unroll
                for (int w = 0; w < 16; ++w) buf[w][bl] = o[w];
            }
            if (WG_SYNC) __syncthreads();
        }
        uint32_t l = uni((uint32_t)(blk - base_blk));
        uint32_t w = uni((uint32_t)(pos & 7) * 2);
        uint64_t lo = buf[w][l], hi = buf[w + 1][l];
        pos = uni64(pos + 1);
        return uni64((hi << 32) | lo);
    }
};

// GoalSampleableRegion::sample_goal of the disc fixture (oxmpl/tests/rrt_rvss_tests.rs:55-66) from its two words:
//   angle = rng.random_range(0.0..2.0 * PI)  (52-bit transform, res = v01 * scale + lo);  radius = self.radius * rng.random::<f64>().sqrt()
//   (StandardUniform f64: (u64 >> 11) * 2^-53);  (x, y) = (cx + radius * cos, cy + radius * sin) through ox_sincos.
// Returns false when the range draw has to be repeated (res >= hi, probability ~2^-53).
__device__ __forceinline__ bool goal_disc_sample(uint64_t w_angle, uint64_t w_radius, const double* goal_c, double goal_radius, double& x, double& y) {
    const double two_pi = 2.0 * 3.14159265358979323846;
    const double v01 = __longlong_as_double((long long)((w_angle >> 12) | 0x3FF0000000000000ull)) - 1.0;
    double angle = v01 * two_pi;
    angle = angle + 0.0;
    const double u01 = (double)(w_radius >> 11) * 0x1p-53;
    const double radius = goal_radius * sqrt(u01);
    double sn, cs;
    ox_sincos(angle < two_pi ? angle : 0.0, sn, cs);
    const double rx = radius * cs, ry = radius * sn;
    x = goal_c[0] + rx;
    y = goal_c[1] + ry;
    return angle < two_pi;
}

// rrt.rs:177-184 + real_vector_state_space.rs:233-249 with rand 0.9's transforms.
// Returns true when the goal was sampled (oxhip_goal_sampler: q = the goal centre and no further draw, or the disc sampler).
template <int D, bool WG_SYNC = true>
__device__ __forceinline__ bool sample_state(RngWindow& rng, const DevParams& p, int dim, const double* goal_c,
                                             double q[D], double goal_radius = 0.0) {
    bool goal;
    if (p.p_int == ~0ull) goal = true;            // Bernoulli ALWAYS_TRUE: no draw
    else goal = rng.next<WG_SYNC>() < p.p_int;    // one u64
    if (goal) {
        if (D >= 2 && p.goal_sampler == OXHIP_GOAL_SAMPLE_UNIFORM_DISC) {
            for (;;) {
                const uint64_t wa = rng.next<WG_SYNC>();
                // (random_range redraws only the angle's word; the radius word is drawn after the angle was accepted)
                const double v01 = __longlong_as_double((long long)((wa >> 12) | 0x3FF0000000000000ull)) - 1.0;
                double angle = v01 * (2.0 * 3.14159265358979323846);
                angle = angle + 0.0;
                if (!(angle < 2.0 * 3.14159265358979323846)) continue;
                const uint64_t wr = rng.next<WG_SYNC>();
                double x, y;
                goal_disc_sample(wa, wr, goal_c, goal_radius, x, y);
                q[0] = x;
                q[D >= 2 ? 1 : 0] = y;
                break;
            }
            return true;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q[k] = goal_c[k];
        return true;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < dim) {
            double res;
            for (;;) {
                uint64_t bits = (rng.next<WG_SYNC>() >> 12) | 0x3FF0000000000000ull;
                double v01 = __longlong_as_double((long long)bits) - 1.0;
                res = v01 * p.scale[k];
                res = res + p.lo[k];
                if (res < p.hi[k]) break;     // else draw again (rand's sample_single loop)
            }
            q[k] = res;
        }
    }
    return false;
}

// squared distance with the reference's summation order (sum starts at 0.0: 0.0 + x*x == x*x)
template <int D>
__device__ __forceinline__ double dist2(const double a[D], const double b[D], int dim) {
    double d0 = a[0] - b[0];
    double acc = d0 * d0;
#pragma unroll
    for (int k = 1; k < D; ++k) {
        if (k < dim) {
            double d = a[k] - b[k];
            double sq = d * d;
            acc = acc + sq;
        }
    }
    return acc;
}

template <int D>
__device__ __forceinline__ void lerp(const double from[D], const double to[D], double t, double out[D], int dim) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < dim) {
            double diff = to[k] - from[k];
            double scaled = diff * t;
            out[k] = from[k] + scaled;
        }
    }
}

// (dist / res).ceil() as usize, clamped to u32 (create() bounds the reachable step count)
__device__ __forceinline__ uint32_t num_steps_u32(double dist, double res) {
    double c = ceil(dist / res);
    if (!(c > 0.0)) return 0;          // NaN / negative / zero: Rust's saturating cast gives 0
    if (c >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)c;
}

// is obstacle j violated by state s?  j < n_spheres: sphere, else box j - n_spheres
template <int D>
__device__ __forceinline__ bool obstacle_hit(const DevParams& p, int dim, const double s[D], uint32_t j) {
    if (j < p.n_spheres) {
        double c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) c[k] = p.sph_c[(size_t)k * p.n_spheres + j];
        double d2 = dist2<D>(c, s, dim);
        return !(d2 > p.sph_thr[j]);
    }
    uint32_t b = j - p.n_spheres;
    bool inside = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < dim) {
            double v = s[k];
            inside = inside && (v >= p.box_lo[(size_t)k * p.n_boxes + b]) && (v <= p.box_hi[(size_t)k * p.n_boxes + b]);
        }
    }
    return inside;
}

// rrt.rs:90-116 evaluated data-parallel: the (step, obstacle) pairs are striped over the
// `nthreads` callers; returns this thread's "found an invalid state" flag (caller ORs them).
// is_valid is pure, so testing all steps equals the reference's first-invalid early exit.
template <int D>
__device__ __forceinline__ bool motion_invalid_partial(const DevParams& p, int dim, const double from[D],
                                                       const double to[D], uint32_t tid, uint32_t nthreads) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    if (nobs == 0) return false;
    double dist = sqrt(dist2<D>(from, to, dim));
    uint32_t nsteps = num_steps_u32(dist, p.res);
    bool bad = false;
    if (nsteps <= 1) {
        for (uint32_t j = tid; j < nobs; j += nthreads) bad = bad || obstacle_hit<D>(p, dim, to, j);
        return bad;
    }
    const uint64_t total = (uint64_t)nsteps * nobs;
    const double dn = (double)nsteps;
    if (total <= 0xFFFFFFFFull) {   // the usual case: 32-bit index arithmetic (a 64-bit divide is ~10x dearer)
        const uint32_t total32 = (uint32_t)total;
        for (uint32_t w = tid; w < total32; w += nthreads) {
            const uint32_t q = w / nobs;
            const uint32_t j = w - q * nobs;
            const double t = (double)(q + 1) / dn;
            double s[D];
            lerp<D>(from, to, t, s, dim);
            bad = bad || obstacle_hit<D>(p, dim, s, j);
            if (w + nthreads < w) break;   // index wrap (total32 near 2^32)
        }
        return bad;
    }
    for (uint64_t w = tid; w < total; w += nthreads) {
        uint32_t step = (uint32_t)(w / nobs) + 1;
        uint32_t j = (uint32_t)(w % nobs);
        double t = (double)step / dn;
        double s[D];
        lerp<D>(from, to, t, s, dim);
        bad = bad || obstacle_hit<D>(p, dim, s, j);
    }
    return bad;
}

// ---------------------------------------------------------------- obstacle table in LDS
// The sphere / box table is read by every motion check; small tables (the usual case: 64 spheres in R^3 = 2 KB)
// are copied into LDS once per launch and the kernel's private copy of DevParams points there, so obstacle_hit
// reads them at LDS latency instead of an L2 round trip per (step, obstacle) pair.
constexpr int kObsLdsDoubles = 1024;
struct ObsLds { double data[kObsLdsDoubles]; };

// Returns the parameters to use from now on; the caller issues a workgroup barrier before the first motion check.
__device__ __forceinline__ DevParams stage_obstacles(const DevParams& p, ObsLds& buf, uint32_t tid, uint32_t nthreads) {
    const uint32_t ns = p.n_spheres, nb = p.n_boxes, dim = p.dim;
    const uint32_t need = (dim + 1) * ns + 2 * dim * nb;
    if (need == 0 || need > (uint32_t)kObsLdsDoubles) return p;
    double* c = buf.data;                 // [dim][ns]
    double* thr = c + dim * ns;           // [ns]
    double* lo = thr + ns;                // [dim][nb]
    double* hi = lo + dim * nb;           // [dim][nb]
    for (uint32_t i = tid; i < dim * ns; i += nthreads) c[i] = p.sph_c[i];
    for (uint32_t i = tid; i < ns; i += nthreads) thr[i] = p.sph_thr[i];
    for (uint32_t i = tid; i < dim * nb; i += nthreads) { lo[i] = p.box_lo[i]; hi[i] = p.box_hi[i]; }
    DevParams q = p;
    q.sph_c = c; q.sph_thr = thr; q.box_lo = lo; q.box_hi = hi;
    return q;
}

// ---------------------------------------------------------------- wave-wide minima on the DPP crossbar
// Six data-parallel-primitive steps (quad perms, row mirrors, row broadcasts) instead of six ds_bpermute
// round trips: ~20 VALU instructions, no LDS traffic.  A shuffle-based (__shfl_xor) reduction of a
// (double, double, index) triple costs ~3,000 cycles per call on this chip -- measured: it was 45 % of an
// RRTConnect iteration -- because every step is a dependent LDS-crossbar access.
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

// rrt.rs:90-116 for a 256-thread workgroup.  With many obstacles the steps are dealt to the waves and the
// obstacles to the lanes: the interpolation parameter and the interpolated state are computed once per step and
// are wave-uniform, no index division is needed, and a lane keeps meeting the same obstacles.  With few obstacles
// and many steps that would idle most lanes, so the (step, obstacle) pairs are striped over all threads instead.
template <int D>
__device__ __forceinline__ bool motion_invalid_wg(const DevParams& p, int dim, const double from[D], const double to[D],
                                                  uint32_t tid, uint32_t nthreads) {
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    if (nobs < 32) return motion_invalid_partial<D>(p, dim, from, to, tid, nthreads);
    const double dist = sqrt(dist2<D>(from, to, dim));
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    const uint32_t wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
    bool bad = false;
    if (nsteps <= 1) {
        for (uint32_t j = tid; j < nobs; j += nthreads) bad = bad || obstacle_hit<D>(p, dim, to, j);
        return bad;
    }
    const double dn = (double)nsteps;
    for (uint32_t step = wave + 1; step <= nsteps && step > wave; step += nwaves) {   // `step > wave`: no wrap at 2^32
        const double t = (double)step / dn;
        double s[D];
        lerp<D>(from, to, t, s, dim);
        for (uint32_t j = lane; j < nobs; j += 64) bad = bad || obstacle_hit<D>(p, dim, s, j);
    }
    return bad;
}

// ---------------------------------------------------------------- binary32 screen of a squared distance
// (stream kernel, RRT*, RRTConnect: the scan of rrt.rs:187-196 run over an fl32 shadow of the tree; it only names a candidate.)
// Error model (u = 2^-24; M = largest coordinate magnitude among the bounds, the goal centre and the tree: every query is a
// sample inside the bounds or the goal centre, every new node a convex combination of two of those):
//   e_k = fl32(fl32(q_k) - fl32(c_k))             |e_k - (q_k - c_k)| <= u|q_k| + u|c_k| + u|e_k|  <= 4.1 u M
//   s   = fma(e_2,e_2, fma(e_1,e_1, e_0*e_0))     s = |e|^2 (1 + eta) + zeta, |eta| <= D u, |zeta| <= D 2^-126
// (+ a relative 2^-18 of slack kept from the round-1 kernel that stole the low 5 bits of s for a slot number), so a node whose
// binary32 squared distance is v lies at true distance
//   sqrt(v)(1 - R) - A <= d <= sqrt(v)(1 + R) + A,  A = sqrt(D) 4.1 u M + 1e-18,  R = 2^-19 + (D + 2) u.
// The kernels use 2A and 2R: a screen winner is accepted only when the runner-up's lower bound exceeds its upper bound -- every
// other screened node is then farther by at least ~1e-7 M, eleven orders of magnitude above the rounding of the binary64
// post-sqrt compare of rrt.rs:192, so strict-'<' / lowest-index semantics cannot be involved; the result is then computed in
// binary64 from the binary64 node.
struct ScreenMargins {
    double a2, r_lo, r_hi;
    bool usable;   // M small enough for binary32 squares
};
__device__ __forceinline__ ScreenMargins screen_margins(double m_all, int dim) {
    ScreenMargins mg;
    const double m = m_all * 1.001;   // interpolation rounding over any chain of inserts
    const double u = 0x1p-24;
    mg.usable = m < 1e15;             // also false for NaN / inf
    mg.a2 = 2.0 * (sqrt((double)dim) * 4.1 * u * m + 1e-18);
    const double r2 = 2.0 * (0x1p-19 + (double)(dim + 2) * u);
    mg.r_lo = 1.0 - r2;
    mg.r_hi = 1.0 + r2;
    return mg;
}
// v1 <= v2: smallest and second smallest screened squared distance (binary32 values held in doubles)
__device__ __forceinline__ bool screen_clear(const ScreenMargins& mg, double v1, double v2) {
    return mg.usable && (sqrt(v2) * mg.r_lo - mg.a2 > sqrt(v1) * mg.r_hi + mg.a2);   // +inf / NaN first: false
}
// largest binary32 squared distance a node within true distance r can show (for threshold screens); +inf when unusable
__device__ __forceinline__ float screen_threshold(const ScreenMargins& mg, double r) {
    if (!mg.usable || !(r < 1e18)) return __builtin_inff();
    const double d = (r * (1.0 + 1e-12) + mg.a2) * mg.r_hi * mg.r_hi;
    const float t = (float)(d * d * (1.0 + 0x1p-20));
    return t;
}

// One thread's share of a screen over the shadow: four consecutive nodes per 16-byte load and coordinate (the SoA rows
// are 16-byte aligned: cap is a multiple of 1024), `visit(i, s)` for every node i < n with its binary32 squared
// distance s to qf.  Loads may run up to 3 nodes past n (inside the row: cap >= n rounded up); those are not visited.
typedef float oxhip_f32x4 __attribute__((ext_vector_type(4)));
template <int D, class F>
__device__ __forceinline__ void screen_scan(const float* tree32, size_t cap, uint32_t n, int dim, const float qf[D],
                                            uint32_t tid, uint32_t nthreads, F&& visit) {
    for (uint32_t i0 = 4u * tid; i0 < n; i0 += 4u * nthreads) {
        oxhip_f32x4 e = *reinterpret_cast<const oxhip_f32x4*>(tree32 + i0) - qf[0];
        oxhip_f32x4 s = e * e;
#pragma unroll
        for (int k = 1; k < D; ++k) {
            if (k < dim) {
                e = *reinterpret_cast<const oxhip_f32x4*>(tree32 + (size_t)k * cap + i0) - qf[k];
                s = __builtin_elementwise_fma(e, e, s);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (i0 + (uint32_t)t < n) visit(i0 + (uint32_t)t, s[t]);
    }
}

// The same for a radius screen: `visit(i)` for every node i < n whose binary32 squared distance is <= thr (one call
// site of `visit`, reached through a loop over the hits of a load: the candidates are few and their handling is heavy).
template <int D, class F>
__device__ __forceinline__ void screen_scan_below(const float* tree32, size_t cap, uint32_t n, int dim, const float qf[D],
                                                  float thr, uint32_t tid, uint32_t nthreads, F&& visit) {
    for (uint32_t i0 = 4u * tid; i0 < n; i0 += 4u * nthreads) {
        oxhip_f32x4 e = *reinterpret_cast<const oxhip_f32x4*>(tree32 + i0) - qf[0];
        oxhip_f32x4 s = e * e;
#pragma unroll
        for (int k = 1; k < D; ++k) {
            if (k < dim) {
                e = *reinterpret_cast<const oxhip_f32x4*>(tree32 + (size_t)k * cap + i0) - qf[k];
                s = __builtin_elementwise_fma(e, e, s);
            }
        }
        uint32_t hits = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) hits |= (i0 + (uint32_t)t < n && s[t] <= thr) ? (1u << t) : 0u;
        for (; hits != 0; hits &= hits - 1) visit(i0 + (uint32_t)(__ffs((int)hits) - 1));
    }
}

// Brings the fl32 shadow of one problem's tree up to date (nodes [valid, n)) and returns the magnitude bound M for this
// launch: the shadowed nodes, the bounds and the goal centre.  Whole workgroup; contains barriers.
template <int D>
__device__ __forceinline__ double shadow_sync(const DevParams& p, int dim, uint32_t prob, const double* tree, float* tree32,
                                              size_t cap, uint32_t n, const double* goal_c, uint32_t* lds_word, uint32_t tid,
                                              uint32_t nthreads) {
    uint32_t* sst = p.shadow_state + 2 * (size_t)prob;
    const uint32_t valid = sst[0] <= n ? sst[0] : 0u;
    if (tid == 0) *lds_word = valid ? sst[1] : 0u;
    __syncthreads();
    uint32_t mab = 0;
    for (uint32_t i = valid + tid; i < n; i += nthreads) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < dim) {
                const float f = (float)tree[(size_t)k * cap + i];
                tree32[(size_t)k * cap + i] = f;
                const uint32_t ab = __builtin_bit_cast(uint32_t, f) & 0x7FFFFFFFu;
                mab = ab > mab ? ab : mab;
            }
        }
    }
    __hip_atomic_fetch_max(lds_word, mab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();   // shadow writes of this workgroup are visible to it; the maximum is final
    double m = (double)__builtin_bit_cast(float, *lds_word) * (1.0 + 0x1p-23);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < dim) {
            m = fmax(m, fmax(fabs(p.lo[k]), fabs(p.hi[k])));
            m = fmax(m, fabs(goal_c[k]));
        }
    }
    return m;
}

// ---------------------------------------------------------------- nearest-neighbour reduction
// The reference takes argmin over (sqrt(d2_i), i) with strict '<' (lowest index among ties).
// The scan compares d2 (no sqrt per node) and also tracks the second-smallest d2: if it lies
// within 3 ulps of the minimum, two different d2 may share one sqrt and the caller re-scans
// exactly (post-sqrt compare).  At most 3 consecutive doubles share a correctly rounded sqrt.
struct Best {
    double b1;    // smallest d2
    double b2;    // second smallest d2 (with multiplicity)
    uint32_t i1;  // lowest index attaining b1
};

__device__ __forceinline__ Best best_init() { return Best{__builtin_inf(), __builtin_inf(), 0xFFFFFFFFu}; }

__device__ __forceinline__ void best_push(Best& b, double d, uint32_t i) {
    // indices arrive in increasing order within a lane, so strict '<' keeps the lowest index
    bool lt = d < b.b1;
    b.b2 = fmin(b.b2, fmax(b.b1, d));
    b.b1 = lt ? d : b.b1;
    b.i1 = lt ? i : b.i1;
}

__device__ __forceinline__ Best best_combine(const Best& a, const Best& b) {
    bool bw = (b.b1 < a.b1) || (b.b1 == a.b1 && b.i1 < a.i1);
    Best r;
    r.b1 = bw ? b.b1 : a.b1;
    r.i1 = bw ? b.i1 : a.i1;
    r.b2 = fmin(fmin(a.b2, b.b2), bw ? a.b1 : b.b1);
    return r;
}

// Wave-wide combine of the per-lane (b1, b2, i1): b1 = min; i1 = lowest index attaining it; b2 = second
// smallest d2 with multiplicity (= b1 when two lanes attain it; else the smaller of the other lanes' minima
// and the winning lane's own second).  Every lane returns the same value.
__device__ __forceinline__ Best best_wave_reduce(Best v) {
    const double m = wave_min_f64(v.b1);
    const bool win = v.b1 == m;
    Best r;
    r.b1 = m;
    r.i1 = wave_min_u32(win ? v.i1 : 0xFFFFFFFFu);
    const double second = wave_min_f64(win ? v.b2 : v.b1);
    r.b2 = __popcll(__ballot(win)) >= 2 ? m : second;
    return r;
}

// near-tie test: could another node's sqrt(d2) equal sqrt(b1)?
__device__ __forceinline__ bool best_ambiguous(const Best& b) {
    uint64_t bound = (uint64_t)__double_as_longlong(b.b1) + 3;  // b1 >= +0: bit pattern is monotone
    return b.b2 <= __longlong_as_double((long long)bound);
}

// exact (post-sqrt) pair for the rare re-scan: lexicographic (dist, index)
struct Exact {
    double dist;
    uint32_t idx;
};
__device__ __forceinline__ Exact exact_combine(const Exact& a, const Exact& b) {
    bool bw = (b.dist < a.dist) || (b.dist == a.dist && b.idx < a.idx);
    return bw ? b : a;
}
// lexicographic (dist, idx) minimum over the wave; every lane returns it
__device__ __forceinline__ Exact exact_wave_reduce(Exact v) {
    Exact r;
    r.dist = wave_min_f64(v.dist);
    r.idx = wave_min_u32(v.dist == r.dist ? v.idx : 0xFFFFFFFFu);
    return r;
}

__device__ __forceinline__ uint64_t fnv_mix(uint64_t h, uint64_t v) { return (h ^ v) * kFnvPrime; }

// RRT's running checksum (build-defined; the CPU checker and the golden generator restate it): every iteration
// has a digest g = FNV-1a fold, from the FNV basis, of (nearest index, bits of q_new[0..dim), verdict), and the run's
// checksum is the polynomial H <- H * P + g (mod 2^64, P the FNV prime, odd: multiplication by P is a bijection), i.e.
// H_T = B P^T + sum_t g_t P^(T-1-t): order-sensitive like a chained hash, but a batch of iterations folds in as
// H * P^m + sum_j g_j * P^(m-1-j) -- one step for a whole wave of queries instead of 5 dependent 64-bit multiplies each.
__device__ __forceinline__ uint64_t chk_push(uint64_t h, uint64_t g) { return h * kFnvPrime + g; }
template <int D>
__device__ __forceinline__ uint64_t iter_digest(uint32_t nearest, const double q_new[D], int dim, bool ok) {
    uint64_t g = fnv_mix(kFnvBasis, (uint64_t)nearest);
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) g = fnv_mix(g, (uint64_t)__double_as_longlong(q_new[k]));
    return fnv_mix(g, ok ? 1ull : 0ull);
}

}  // namespace oxhip
