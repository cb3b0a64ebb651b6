// rrt_connect.hip -- RRTConnect (oxmpl/src/geometric/planners/rrt_connect.rs:121-159,227-309) on the GPU: ONE WAVE PER PROBLEM
// (a 64-thread workgroup, four per CU = one per SIMD).  Iteration k sees the trees left by iterations < k; per iteration up to
// two extends, each a nearest-neighbour scan + steer + motion check + insert.
//
// RRTConnect trees are small (tens to hundreds of nodes) and an iteration is a chain of dependent steps, so what counts is the
// length of that chain, not throughput: nothing here waits at a barrier (LDS is in order per wave), the first
// kConnLdsBytes / (2 * dim * 8) nodes of BOTH trees live in LDS next to their SoA arrays in HBM (which receive every store -- for
// get_tree / resume -- and serve the nodes beyond that capacity), the obstacle table is staged in LDS, 64 iterations are sampled at
// a time, a motion check deals the obstacles to the lanes (or, with few obstacles, the states), an Advanced extend's step count is
// a constant where that is safe, and the checksum's folds run on the scalar unit.  Rounds 1-2 ran a 256-thread workgroup per
// problem: four barriers per extend and a wave-serial sampler made an iteration 5.8-6.7 us; this kernel's is a third of that.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

constexpr int kConnLdsBytes = 23040;   // LDS for the two trees of one problem (the whole workgroup: <= 40 KB, four per CU)

// cycle stamps of problem 0 (tools/build_variant.sh with -DOXHIP_CONN_STAMPS + oxhip_rrt_batch_enable_stamps; not in the product build):
// dbg[0..8] = cycles sampling, in the nearest-neighbour search, steering, in the motion check, -, in checksum + insert + goal test, in the
// whole loop; iterations; extends
#ifdef OXHIP_CONN_STAMPS
#define CONN_T(v) const uint64_t v = (uint64_t)__builtin_readcyclecounter()
#define CONN_ACC(i, d) acc[i] += (d)
#else
#define CONN_T(v)
#define CONN_ACC(i, d)
#endif

template <int D>
struct ConnShared {
    static constexpr int N = kConnLdsBytes / (2 * D * 8);   // nodes of each tree mirrored in LDS
    uint32_t rng_buf[16][64];
    double q[D][64];              // the samples of the current block of iterations ...
    uint64_t pos_after[64];       // ... and the stream position after each of them
    double tree_a[D][N], tree_b[D][N];
    ObsLds obs;
    double tdiv[8][8];            // (s + 1) / n, n = 1 .. 8: the interpolation parameters of motions of up to eight states
};

// One tree: its first N nodes mirrored in LDS ([k][i], conflict-free for the strided scan), all of it in HBM.
template <int D>
struct ConnTree {
    static constexpr int N = ConnShared<D>::N;
    double* g;            // SoA [dim][cap] in HBM
    double (*l)[N];       // LDS [D][N]
    size_t cap;
};

// what the constant step count of an Advanced extend depends on (wave-uniform, kept current by every insert)
struct ConnRange { float mag; };   // largest |coordinate| of any node of either tree

__device__ __forceinline__ double conn_readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Lane-parallel sampling of m <= 64 consecutive iterations (rrt_connect.rs:258-262 + rvss.rs:233-249; the scheme of rrt_cells.hip's
// cells_sample): lane j draws iteration j.  Where its words start depends on how many of the iterations before it sampled the goal
// (one word instead of 1 + dim), so the goal mask is iterated to its fixed point.  Returns false, nothing written, when a range draw
// was rejected or the window is too short.
template <int D>
__device__ __forceinline__ bool conn_sample64(RngWindow& rng, const DevParams& p, int dim, const double* goal_c, uint32_t m, uint32_t lane,
                                              ConnShared<D>& sh) {
    const uint32_t per = 1u + (uint32_t)dim;
    const uint64_t win_lo = rng.base_blk * 8;
    const uint64_t pos0 = rng.pos;
    if (pos0 < win_lo || pos0 + (uint64_t)m * per > win_lo + 512) return false;
    const uint32_t rel0 = (uint32_t)(pos0 - win_lo);
    const bool act = lane < m;
    const bool always_goal = p.p_int == ~0ull;   // Bernoulli ALWAYS_TRUE: no draw at all
    auto word = [&](uint32_t rel) -> uint64_t {
        const uint32_t a = rel0 + rel, bl = a >> 3, w = (a & 7u) * 2u;
        return ((uint64_t)rng.buf[w + 1][bl] << 32) | rng.buf[w][bl];
    };
    uint64_t goal_mask = always_goal ? ~0ull : 0ull;
    uint32_t off = 0u;
    if (!always_goal) {
        const uint64_t below = (1ull << lane) - 1ull;
        for (uint32_t round = 0; round <= m; ++round) {
            off = act ? per * lane - (uint32_t)dim * (uint32_t)__popcll(goal_mask & below) : 0u;
            const uint64_t now = __ballot(act && word(off) < p.p_int);
            if (now == goal_mask) break;
            goal_mask = now;
        }
    }
    const bool goal = (goal_mask >> lane) & 1ull;
    double q[D];
    bool redraw = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        if (k < dim) {
            const uint64_t bits = (word(act && !goal ? off + 1u + (uint32_t)k : 0u) >> 12) | 0x3FF0000000000000ull;
            const double v01 = __longlong_as_double((long long)bits) - 1.0;
            double res = v01 * p.scale[k];
            res = res + p.lo[k];
            redraw = redraw || !(res < p.hi[k]);
            q[k] = goal ? goal_c[k] : res;
        }
    }
    if (__ballot(act && redraw && !goal) != 0) return false;
    const uint32_t cnt = always_goal ? 0u : (goal ? 1u : per);
    if (act) {
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) sh.q[k][lane] = q[k];
        sh.pos_after[lane] = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}
template <int D>
__device__ __forceinline__ void conn_sample_block(RngWindow& rng, const DevParams& p, int dim, const double* goal_c, uint32_t m, uint32_t lane,
                                                  ConnShared<D>& sh) {
    const uint64_t need_hi = rng.pos + (uint64_t)m * (1u + (uint32_t)dim);
    if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
        rng.base_blk = uni64(rng.pos >> 3);
        uint32_t o[16];
        chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
    }
    if (!conn_sample64<D>(rng, p, dim, goal_c, m, lane, sh)) {
        for (uint32_t b = 0; b < m; ++b) {   // (never expected) a redraw: one by one
            double qn[D];
            sample_state<D, false>(rng, p, dim, goal_c, qn);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) sh.q[k][b] = qn[k];
                sh.pos_after[b] = rng.pos;
            }
        }
    }
}

// nearest node of rrt_connect.rs:128-136: d2 compare with second-smallest tracking, exact post-sqrt fallback on a near-tie.  A lane
// keeps the coordinates of its own best node, and q_near is read out of the winning lane's registers (no second trip to the tree; the
// LDS range and the HBM range of a tree are walked by separate loops: one loop over a select of the two addresses compiles to flat loads).
template <int D>
__device__ __forceinline__ void conn_nearest(const ConnTree<D>& tree, int dim, uint32_t n, const double q[D], uint32_t lane, uint32_t& nearest,
                                             double& min_dist, double q_near[D]) {
    constexpr uint32_t N = (uint32_t)ConnTree<D>::N;
    const uint32_t nl = n < N ? n : N;
    Best best = best_init();
    double bc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) bc[k] = 0.0;
    for (uint32_t i = lane; i < nl; i += 64u) {   // indices ascend within a lane: strict < keeps the lowest
        double c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.l[k][i];
        const double d2 = dist2<D>(c, q, dim);
        const bool lt = d2 < best.b1;
        best_push(best, d2, i);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) bc[k] = lt ? c[k] : bc[k];
    }
    if (n > N) {   // beyond the mirror: this wave's own stores, drained first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t i = N + lane; i < n; i += 64u) {
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.g[(size_t)k * tree.cap + i];
            const double d2 = dist2<D>(c, q, dim);
            const bool lt = d2 < best.b1;
            best_push(best, d2, i);
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) bc[k] = lt ? c[k] : bc[k];
        }
    }
    const uint32_t mine = best.i1;
    best = best_wave_reduce(best);
    uint32_t mine_x = mine;
    if (!best_ambiguous(best)) {
        nearest = uni(best.i1);
        min_dist = sqrt(best.b1);
    } else {   // the reference's own comparison: post-sqrt distances, ascending index
        Exact e{__builtin_inf(), 0xFFFFFFFFu};
        for (uint32_t i = lane; i < nl; i += 64u) {
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.l[k][i];
            const double d = sqrt(dist2<D>(c, q, dim));
            const bool lt = d < e.dist;
            e.dist = lt ? d : e.dist;
            e.idx = lt ? i : e.idx;
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) bc[k] = lt ? c[k] : bc[k];
        }
        if (n > N) {
            for (uint32_t i = N + lane; i < n; i += 64u) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.g[(size_t)k * tree.cap + i];
                const double d = sqrt(dist2<D>(c, q, dim));
                const bool lt = d < e.dist;
                e.dist = lt ? d : e.dist;
                e.idx = lt ? i : e.idx;
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) bc[k] = lt ? c[k] : bc[k];
            }
        }
        mine_x = e.idx;
        e = exact_wave_reduce(e);
        nearest = uni(e.idx);
        min_dist = e.dist;
    }
    const int L = __builtin_ctzll(__ballot(mine_x == nearest) | (1ull << 63));   // the lane that holds the winner (a node belongs to one lane)
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) q_near[k] = conn_readlane_f64(bc[k], L);
}

// The obstacle table as the motion check reads it: staged in LDS (OBS_LDS: a typed LDS pointer -- a DevParams field that was
// re-pointed at LDS is a generic pointer, and a flat load costs several times a ds_read) or, when it does not fit, in HBM / L2.
struct ConnObs {
    const double* c;     // [dim][ns] sphere centres
    const double* thr;   // [ns]
    const double* lo;    // [dim][nb]
    const double* hi;    // [dim][nb]
    uint32_t ns, nb;
    const uint64_t* grid;   // R^2 / R^3, up to 64 spheres: bit j of a cell's word = sphere j reaches the cell (DevParams::sph_grid; null: none)
    uint32_t G;
    double ginv[3];         // cells per unit length
    const double (*tdiv)[8];
};

// rrt_connect.rs:166-189 for one wave; returns the wave-uniform verdict "some tested state is invalid".  Up to 64 states: the
// obstacles are dealt to the lanes -- a lane reads its obstacle ONCE per motion -- and the states are walked (the interpolation
// parameters s / n, one division per lane, are handed round with v_readlane).  More states: the states are dealt to the lanes and
// the obstacles walked.  is_valid is pure, so testing every state equals the reference's first-invalid early exit.
template <int D>
__device__ __forceinline__ bool conn_motion_invalid(const ConnObs& ob, const DevParams& p, int dim, const double from[D], const double to[D],
                                                    uint32_t nsteps, uint32_t lane) {
    const uint32_t nobs = ob.ns + ob.nb;
    if (nobs == 0) return false;
    const uint32_t S = nsteps <= 1u ? 1u : nsteps;   // states tested: `to` alone, or steps 1 ..= nsteps
    const double dn = (double)nsteps;
    if ((D == 2 || D == 3) && ob.grid && ob.nb == 0u && S <= 8u) {
        // the usual case in R^2 / R^3: eight lanes per state, the state's spheres looked up in the grid (a state outside the bounds:
        // every sphere); lane k of a state takes the listed spheres j = k mod 8 -- almost always none or one
        const uint32_t sidx = lane >> 3, k = lane & 7u;
        bool bad = false;
        if (sidx < S) {
            double st[D];
            if (nsteps > 1u) lerp<D>(from, to, ob.tdiv[nsteps - 1u][sidx], st, dim);
            else {
#pragma unroll
                for (int c = 0; c < D; ++c) st[c] = to[c];
            }
            bool inside = true;
            uint32_t ci[3] = {0u, 0u, 0u};
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double f = (st[c] - p.lo[c]) * ob.ginv[c];
                inside = inside && f >= 0.0 && f < (double)ob.G;   // (NaN: outside)
                ci[c] = inside ? (uint32_t)f : 0u;
            }
            uint64_t m = ob.grid[D == 3 ? (ci[2] * ob.G + ci[1]) * ob.G + ci[0] : ci[1] * ob.G + ci[0]];
            if (!inside) m = ~0ull;
            if (ob.ns < 64u) m &= (1ull << ob.ns) - 1ull;
            m &= 0x0101010101010101ull << k;
            while (m != 0) {
                const uint32_t j = (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                double cj[D];
#pragma unroll
                for (int c = 0; c < D; ++c) cj[c] = ob.c[(size_t)c * ob.ns + j];
                bad = bad || !(dist2<D>(cj, st, dim) > ob.thr[j]);   // (obstacle_hit's operand order: centre - state)
            }
        }
        return __ballot(bad) != 0;
    }
    if (S <= 64u) {
        const double tl = (double)(lane + 1u) / dn;
        for (uint32_t j0 = 0; j0 < nobs; j0 += 64u) {
            const uint32_t j = j0 + lane;
            const bool act = j < nobs, sphere = j < ob.ns;
            const uint32_t js = sphere ? j : 0u, jb = (act && !sphere) ? j - ob.ns : 0u;
            double a[D], b[D];   // sphere: centre (and the threshold in thr); box: lo, hi
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k < dim) {
                    a[k] = sphere ? (ob.ns ? ob.c[(size_t)k * ob.ns + js] : 0.0) : (ob.nb ? ob.lo[(size_t)k * ob.nb + jb] : 0.0);
                    b[k] = (!sphere && ob.nb) ? ob.hi[(size_t)k * ob.nb + jb] : 0.0;
                }
            }
            const double thr = (sphere && ob.ns) ? ob.thr[js] : 0.0;
            // (no short-circuits in the loops below: a straight line per state, the verdicts folded with bit operations)
            uint32_t bad = 0u;
            const bool boxes = j0 + 64u > ob.ns && ob.nb != 0u;   // (uniform) does this batch of 64 obstacles hold a box at all?
            if (nsteps <= 1u) {
                const uint32_t hit_s = !(dist2<D>(a, to, dim) > thr) ? 1u : 0u;   // (a sphere's test in obstacle_hit's operand order: centre - state)
                uint32_t inside = 1u;
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) inside &= ((to[k] >= a[k]) ? 1u : 0u) & ((to[k] <= b[k]) ? 1u : 0u);
                bad = sphere ? hit_s : inside;
            } else if (S <= 8u && !boxes) {   // the usual case, unrolled: eight independent chains (states past the last one are masked)
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    double st[D];
                    lerp<D>(from, to, conn_readlane_f64(tl, s), st, dim);
                    bad |= ((uint32_t)s < S && !(dist2<D>(a, st, dim) > thr)) ? 1u : 0u;
                }
            } else if (S <= 8u) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    double st[D];
                    lerp<D>(from, to, conn_readlane_f64(tl, s), st, dim);
                    const uint32_t hit_s = !(dist2<D>(a, st, dim) > thr) ? 1u : 0u;
                    uint32_t inside = 1u;
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < dim) inside &= ((st[k] >= a[k]) ? 1u : 0u) & ((st[k] <= b[k]) ? 1u : 0u);
                    bad |= (uint32_t)s < S ? (sphere ? hit_s : inside) : 0u;
                }
            } else if (!boxes) {
                for (uint32_t s = 0; s < S; ++s) {
                    double st[D];
                    lerp<D>(from, to, conn_readlane_f64(tl, (int)s), st, dim);
                    bad |= !(dist2<D>(a, st, dim) > thr) ? 1u : 0u;
                }
            } else {
                for (uint32_t s = 0; s < S; ++s) {
                    double st[D];
                    lerp<D>(from, to, conn_readlane_f64(tl, (int)s), st, dim);
                    const uint32_t hit_s = !(dist2<D>(a, st, dim) > thr) ? 1u : 0u;
                    uint32_t inside = 1u;
#pragma unroll
                    for (int k = 0; k < D; ++k) if (k < dim) inside &= ((st[k] >= a[k]) ? 1u : 0u) & ((st[k] <= b[k]) ? 1u : 0u);
                    bad |= sphere ? hit_s : inside;
                }
            }
            if (__ballot(act && bad != 0u) != 0) return true;
        }
        return false;
    }
    for (uint32_t s0 = 0; s0 < S && s0 + 64u > s0; s0 += 64u) {
        const uint32_t s = s0 + lane;
        bool bad = false;
        if (s < S) {
            const double t = (double)(s + 1u) / dn;
            double st[D];
            lerp<D>(from, to, t, st, dim);
            for (uint32_t j = 0; j < nobs; ++j) bad = bad || obstacle_hit<D>(p, dim, st, j);
        }
        if (__ballot(bad) != 0) return true;
    }
    return false;
}

// extend() of rrt_connect.rs:121-159 for the wave.  Returns 0 = motion invalid (None), 1 = Advanced, 2 = Reached; `nearest` and
// `q_new` are filled in every case.  On success the new node is appended at index n and n is incremented.
template <int D>
__device__ __forceinline__ int conn_extend(const DevParams& p, const ConnObs& ob, int dim, const ConnTree<D>& tree, int32_t* parent, uint32_t& n,
                                           ConnRange& rg, const double q[D], uint32_t& nearest, double q_new[D], uint64_t* acc) {
    constexpr uint32_t N = (uint32_t)ConnTree<D>::N;
    const uint32_t lane = threadIdx.x & 63u;
    double min_dist, q_near[D];
    CONN_T(t0);
    conn_nearest<D>(tree, dim, n, q, lane, nearest, min_dist, q_near);
    CONN_T(t1);
    int result;
    uint32_t nsteps;
    if (min_dist > p.max_distance) {  // rrt_connect.rs:140-147
        const double t = p.max_distance / min_dist;
        lerp<D>(q_near, q, t, q_new, dim);
        // check_motion's step count is ceil(distance(q_near, q_new) / res), and q_new lies max_distance along the segment from q_near:
        // the computed distance is max_distance up to a few roundings of quantities no larger than the coordinates in play
        // (< 2^-45 (mag + 1 + max_distance) by a wide margin).  When max_distance / res is farther than that from an integer (adv_slack, in distance
        // units, from the host) the count is the constant adv_steps: no square root, no division, no ceil.
        float qm = 0.0f;
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) qm = fmaxf(qm, fabsf((float)q[k]));
        const bool known = p.adv_steps != 0u && ((double)(rg.mag + qm + 1.0f) + p.max_distance) * 0x1p-45 < p.adv_slack;   // (the distance itself is one of those quantities)
        nsteps = known ? p.adv_steps : num_steps_u32(sqrt(dist2<D>(q_near, q_new, dim)), p.res);
        result = 1;
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q_new[k] = q[k];
        nsteps = num_steps_u32(min_dist, p.res);   // sqrt(dist2(q_near, q)): the value the scan produced for this very pair
        result = 2;
    }
    CONN_T(t2);
    const bool invalid = conn_motion_invalid<D>(ob, p, dim, q_near, q_new, nsteps, lane);
    CONN_T(t3);
    CONN_ACC(1, t1 - t0); CONN_ACC(2, t2 - t1); CONN_ACC(3, t3 - t2); CONN_ACC(8, 1);
    if (invalid) return 0;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < dim) {
                if (n < N) tree.l[k][n < N ? n : 0u] = q_new[k];
                tree.g[(size_t)k * tree.cap + n] = q_new[k];
            }
        }
        parent[n] = (int32_t)nearest;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) rg.mag = fmaxf(rg.mag, fabsf((float)q_new[k]));
    // the next scan is this wave's own and LDS is in order per wave (the fence keeps the compiler from moving the store)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    ++n;
    return result;
}

// mirror of the first nodes of a tree an earlier launch (or setup) left in HBM; folds every node into the range
template <int D>
__device__ __forceinline__ void conn_mirror_load(const ConnTree<D>& tree, int dim, uint32_t n, uint32_t lane, ConnRange& rg) {
    constexpr uint32_t N = (uint32_t)ConnTree<D>::N;
    float mag = 0.0f;
    for (uint32_t i = lane; i < n; i += 64u) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (k < dim) {
                const double v = tree.g[(size_t)k * tree.cap + i];
                if (i < N) tree.l[k][i < N ? i : 0u] = v;
                mag = fmaxf(mag, fabsf((float)v));
            }
        }
    }
    mag = __uint_as_float(~wave_min_u32(~__float_as_uint(mag)));   // (maximum of non-negative values through their bit patterns)
    rg.mag = fmaxf(rg.mag, mag);
}

template <int DIM, bool OBS_LDS>
__global__ __launch_bounds__(64) void rrt_connect_kernel(DevParams p) {
    constexpr int D = DIM ? DIM : kMaxDim;
    const int dim = DIM ? DIM : (int)p.dim;
    const uint32_t prob = blockIdx.x, lane = threadIdx.x;
    __shared__ ConnShared<D> sh;
    static_assert(sizeof(ConnShared<D>) <= 40960, "four problems per CU");
    ConnObs ob{p.sph_c, p.sph_thr, p.box_lo, p.box_hi, p.n_spheres, p.n_boxes, (D == 2 || D == 3) ? p.sph_grid : nullptr, p.sph_grid_G, {0.0, 0.0, 0.0}, sh.tdiv};
    if (ob.grid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) if (k < dim) ob.ginv[k] = (double)ob.G / (p.hi[k] - p.lo[k]);
    }
    sh.tdiv[lane >> 3][lane & 7u] = (double)((lane & 7u) + 1u) / (double)((lane >> 3) + 1u);
    if (OBS_LDS) {   // the table at LDS latency, through typed pointers (stage_obstacles' layout; the barrier below covers it)
        double* c = sh.obs.data;
        double* thr = c + (size_t)dim * ob.ns;
        double* lo = thr + ob.ns;
        double* hi = lo + (size_t)dim * ob.nb;
        for (uint32_t i = lane; i < (uint32_t)dim * ob.ns; i += 64u) c[i] = p.sph_c[i];
        for (uint32_t i = lane; i < ob.ns; i += 64u) thr[i] = p.sph_thr[i];
        for (uint32_t i = lane; i < (uint32_t)dim * ob.nb; i += 64u) { lo[i] = p.box_lo[i]; hi[i] = p.box_hi[i]; }
        ob.c = c; ob.thr = thr; ob.lo = lo; ob.hi = hi;
    }

    ProblemState st = p.state[prob];
    if (st.goal_node >= 0) return;  // solved: RRTConnect::solve returned Ok

    const size_t cap = p.cap;
    const ConnTree<D> tree_a{p.tree + (size_t)prob * p.dim * cap, sh.tree_a, cap};
    const ConnTree<D> tree_b{p.tree_b + (size_t)prob * p.dim * cap, sh.tree_b, cap};
    int32_t* par_a = p.parent + (size_t)prob * cap;
    int32_t* par_b = p.parent_b + (size_t)prob * cap;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) goal_c[k] = p.goal_c[(size_t)prob * p.dim + k];
    const double goal_thr = p.goal_thr[prob];

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint32_t na = st.n_nodes, nb = st.n_nodes_b;
    ConnRange rg{0.0f};   // a solve call continues the trees an earlier one left in HBM
    conn_mirror_load<D>(tree_a, dim, na, lane, rg);
    conn_mirror_load<D>(tree_b, dim, nb, lane, rg);
    __syncthreads();
    // iterations sampled at a time: a block's words must fit the 512-word window whatever the stream position's offset in its block
    const uint32_t blk = (1u + (uint32_t)dim) * 64u <= 504u ? 64u : 32u;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    uint64_t draws = st.draws;   // stream position after the last iteration that ran (the block sampler runs ahead of it)
    uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)acc;
    CONN_T(t_begin);
    for (uint64_t it = 0; it < p.budget; ++it) {
        if (na >= p.max_nodes || nb >= p.max_nodes) { stop = 2; break; }
        const bool grow_start = na <= nb;  // rrt_connect.rs:249-254
        CONN_T(ts0);
        const uint32_t slot = (uint32_t)it & (blk - 1u);   // (blk is 64 or 32)
        if (slot == 0u) conn_sample_block<D>(rng, p, dim, goal_c, p.budget - it < blk ? (uint32_t)(p.budget - it) : blk, lane, sh);   // rrt_connect.rs:258-262
        double q_rand[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q_rand[k] = sh.q[k][slot];
        draws = sh.pos_after[slot];
        CONN_T(ts1);
        CONN_ACC(0, ts1 - ts0);
        const uint64_t ext0 = acc[1] + acc[2] + acc[3];
        (void)ext0;
        uint32_t near_a = 0, near_b = 0;
        double q_new_a[D], q_new_b[D];
        // the tree that grows this iteration and the other one (one copy of extend()'s code serves either role)
        const ConnTree<D> t1{grow_start ? tree_a.g : tree_b.g, grow_start ? tree_a.l : tree_b.l, cap};
        const ConnTree<D> t2{grow_start ? tree_b.g : tree_a.g, grow_start ? tree_b.l : tree_a.l, cap};
        uint32_t n1 = grow_start ? na : nb, n2 = grow_start ? nb : na;
        const int ra = conn_extend<D>(p, ob, dim, t1, grow_start ? par_a : par_b, n1, rg, q_rand, near_a, q_new_a, acc);
        uint64_t h = fnv_mix(uni64(st.checksum), grow_start ? 1ull : 0ull);   // (wave-uniform by construction: the folds run on the scalar unit)
        h = fnv_mix(h, (uint64_t)near_a);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) h = fnv_mix(h, uni64((uint64_t)__double_as_longlong(q_new_a[k])));
        h = fnv_mix(h, (uint64_t)ra);
        st.iterations++;
        bool done = false;
        if (ra) {
            const uint32_t idx_a = n1 - 1;
            if (grow_start && dist2<D>(q_new_a, goal_c, dim) <= goal_thr) {  // rrt_connect.rs:271-274
                st.goal_node = (int32_t)idx_a;
                st.goal_node_b = -1;
                done = true;
            } else {
                const int rb = conn_extend<D>(p, ob, dim, t2, grow_start ? par_b : par_a, n2, rg, q_new_a, near_b, q_new_b, acc);
                h = fnv_mix(h, (uint64_t)near_b);
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) h = fnv_mix(h, uni64((uint64_t)__double_as_longlong(q_new_b[k])));
                h = fnv_mix(h, (uint64_t)rb);
                if (rb == 2) {  // Reached: rrt_connect.rs:281-305
                    const uint32_t idx_b = n2 - 1;
                    st.goal_node = (int32_t)(grow_start ? idx_a : idx_b);
                    st.goal_node_b = (int32_t)(grow_start ? idx_b : idx_a);
                    done = true;
                }
            }
        }
        na = grow_start ? n1 : n2;
        nb = grow_start ? n2 : n1;
        st.checksum = h;
        CONN_T(ts2);
        CONN_ACC(5, ts2 - ts1 - (acc[1] + acc[2] + acc[3] - ext0)); CONN_ACC(7, 1);
        if (done) { stop = 0; break; }
    }
#ifdef OXHIP_CONN_STAMPS
    if (prob == 0 && lane == 0 && p.dbg) {
        acc[6] = (uint64_t)__builtin_readcyclecounter() - t_begin;
        for (int k = 0; k < 9; ++k) p.dbg[k] = acc[k];
    }
#endif
    if (lane == 0) {
        st.n_nodes = na;
        st.n_nodes_b = nb;
        st.draws = draws;
        st.stop_reason = stop;
        p.state[prob] = st;
    }
}

template <int DIM>
static void launch_connect_dim(const DevParams& p, hipStream_t stream) {
    const dim3 grid(p.n_problems), block(64);
    const uint32_t need = (p.dim + 1) * p.n_spheres + 2 * p.dim * p.n_boxes;
    if (need <= (uint32_t)kObsLdsDoubles) hipLaunchKernelGGL((rrt_connect_kernel<DIM, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((rrt_connect_kernel<DIM, false>), grid, block, 0, stream, p);
}

void launch_rrt_connect(const DevParams& p, hipStream_t stream) {
    switch (p.dim) {
        case 2: launch_connect_dim<2>(p, stream); break;
        case 3: launch_connect_dim<3>(p, stream); break;
        case 4: launch_connect_dim<4>(p, stream); break;
        case 5: launch_connect_dim<5>(p, stream); break;
        case 6: launch_connect_dim<6>(p, stream); break;
        case 7: launch_connect_dim<7>(p, stream); break;
        default: launch_connect_dim<0>(p, stream); break;
    }
}

}  // namespace oxhip
