// rrt_star_wire.hip -- the wiring stages of the decoupled RRT* (oxmpl/src/geometric/planners/rrt_star.rs:170-289).
//
// What makes RRT* expensive on a GPU is find_neighbours (:121-131): a second scan of the whole tree per accepted iteration,
// issued only after that iteration's nearest-neighbour query has been answered -- a chain of dependent memory round trips.
// But the GEOMETRY of an RRT* run does not depend on any cost:
//   * q_rand comes from the RNG stream, nearest from the node positions, q_new from both (:178-209);
//   * the node is inserted iff check_motion(q_near, q_new) holds (:212-214) -- the nearest node is always an admissible
//     parent, choose-parent (:225-241) can only replace it -- and it is inserted AT q_new whatever its parent (:244-250);
//   * the loop ends on goal.is_satisfied(q_new) (:285-288).
// So the sequence of node positions (and of nearest indices, verdicts, the goal node, the RNG position) is exactly RRT's
// (rrt.rs:170-225) on the same stream, and rrt_lanes.hip produces it at RRT speed.  What is left -- parents and costs --
// is a function of the positions alone:
//   1. star_pairs   N(i) = { j < i : distance(x_i, x_j) < search_radius }, ascending j: the tree as node i found it.
//                   All pairs of all nodes at once (count, prefix sum, fill): throughput, not latency.
//   2. star_edges   per neighbour pair: distance, check_motion(x_j, x_i) and check_motion(x_i, x_j) -- is_valid is pure, so
//                   evaluating a motion the reference skipped (its cost test failed) changes nothing.
//   3. star_wire    one wave per problem walks the nodes in insertion order: choose parent = the lexicographic minimum of
//                   (cost via neighbour, index) among the neighbours with a valid motion and a cost below the nearest
//                   node's (what the reference's ascending walk with its running minimum computes, :231-240); rewire: every
//                   neighbour but the parent whose cost drops and whose motion is valid (:253-282; each decision depends on
//                   that neighbour's own cost only).  Costs are read and written in order, so rewires of earlier nodes are
//                   seen by later ones exactly as in the reference.
// Stages 2 and 3 run segment by segment (eighths of the round's pairs): while star_edges checks segment s + 1 on the batch's
// stream, star_wire wires the nodes whose lists end inside segment s on a second stream -- one latency-bound wave per problem
// next to a throughput-bound kernel (oxhip_api.hip, wire_new_nodes).
// Everything that enters a result is binary64 in the reference's evaluation order.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "motion_seq.hpp"

namespace oxhip {

constexpr int kPairThreads = 256;
constexpr int kChunk = 32;   // neighbour indices per chunk (StarChunk)

// ---- 0. binary32 copy of the node positions (what the pair search screens with) and the largest magnitude among them
template <int DIM>
__global__ __launch_bounds__(256) void star_shadow_kernel(DevParams p) {
    const uint32_t prob = blockIdx.y, i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n = p.state[prob].n_nodes;
    if (i >= n) return;
    const size_t cap = p.cap;
    uint32_t mab = 0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const float f = (float)p.tree[((size_t)prob * DIM + k) * cap + i];
        p.tree32[((size_t)prob * DIM + k) * cap + i] = f;
        const uint32_t ab = __builtin_bit_cast(uint32_t, f) & 0x7FFFFFFFu;
        mab = ab > mab ? ab : mab;
    }
    atomicMax(&p.shadow_state[2 * (size_t)prob + 1], mab);   // (non-negative floats order like their bit patterns; NaN sorts last: unusable)
}

typedef float sw_f32x2 __attribute__((ext_vector_type(2)));

// ---- 1. neighbour lists.  A thread owns node i and walks j = 0 .. i-1; the 64 nodes of a wave are consecutive, j is
// wave-uniform (x_j comes through the scalar cache).  32 nodes j per trip are SCREENED in packed binary32 (two j per
// instruction: difference, square, fused multiply-add -- the screen of rrt_stream.hip / rrt_star.hip, error model and
// threshold of rrt_device.hpp: a pair inside the radius cannot show a binary32 squared distance above screen_threshold);
// the trip's hits are kept as a bit mask and only they -- 0.7 % of the pairs at radius 1 in configs[1]'s world -- get the
// reference's binary64 test d2 <= T(search_radius) (rrt_star.rs:125), in ascending j.  FILL = false counts, FILL = true
// writes (j, i, d2) at the node's offset (the owner i rides in the flags word until the edge kernel replaces it) and the
// node's distance to its nearest node.
template <int DIM, bool FILL>
__global__ __launch_bounds__(kPairThreads, 8) void star_pairs_kernel(DevParams p) {   // (8 waves per SIMD: the loop lives on scalar-load latency hiding)
    // the counting pass also keeps what it finds: a lane collects its hits in LDS and writes them out 32 at a time as a chunk
    // (owner node, ordinal, 32 neighbour indices) wherever the problem's chunk cursor points; when all of a round's pending
    // nodes fit (the usual case) the edge kernel builds the lists from the chunks and the second search (FILL) is not run
    __shared__ uint16_t hitbuf[FILL ? 1 : kChunk][FILL ? 1 : kPairThreads];   // (node indices fit 16 bits: the lane-per-query kernel holds at most 20,480 nodes)
    const uint32_t prob = blockIdx.y;
    const uint32_t n = p.state[prob].n_nodes;
    const uint32_t w0 = p.wired[prob];
    const uint32_t hi = FILL ? w0 + p.nbr_take[prob] : n;   // nodes [w0, hi)
    const uint32_t wave_first = uni(w0 + blockIdx.x * kPairThreads + (threadIdx.x & ~63u));
    if (wave_first >= hi) return;
    const uint32_t i = w0 + blockIdx.x * kPairThreads + threadIdx.x;
    const bool act = i < hi;
    const size_t cap = p.cap;
    const double* __restrict__ tree = p.tree + (size_t)prob * DIM * cap;
    // the binary32 rows are read through the CONSTANT address space: nothing writes them during this kernel, and only then
    // does the compiler keep the wave-uniform loads scalar (s_load_dwordx16) now that the kernel also stores chunks and bumps a
    // cursor -- with ordinary global loads the counting pass took 26.7 ms instead of 12
    typedef const __attribute__((address_space(4))) float* cfloat_ptr;
    typedef const __attribute__((address_space(4))) sw_f32x2* cfloat2_ptr;
    const cfloat_ptr t32 = (cfloat_ptr)(uintptr_t)(p.tree32 + (size_t)prob * DIM * cap);
    double x[DIM];
    float xf[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        x[k] = tree[(size_t)k * cap + (act ? i : w0)];
        xf[k] = t32[(size_t)k * cap + (act ? i : w0)];
    }
    const uint32_t last = hi - 1u < wave_first + 63u ? hi - 1u : wave_first + 63u;   // the wave's largest node: j < last
    const double thr = p.thr_search;
    float thr32;
    {
        const double m_all = (double)__builtin_bit_cast(float, p.shadow_state[2 * (size_t)prob + 1]) * (1.0 + 0x1p-23);
        thr32 = screen_threshold(screen_margins(m_all, DIM), sqrt(thr));   // +inf when the screen cannot be used
    }
    const bool screen = thr32 < __builtin_inff();
    uint32_t cnt = 0, nbuf = 0, seq = 0;
    StarEntry* __restrict__ out = nullptr;
    if (FILL && act) out = p.pool + (size_t)prob * p.pool_share + p.nbr_off[(size_t)prob * cap + i];
    auto flush = [&](uint32_t c) {   // (divergent, rare: one per 32 hits of a lane)
        const uint32_t slot = atomicAdd(&p.chunk_cursor[prob], 1u);
        if (slot < p.chunk_share) {   // (beyond it: the host sees the cursor and falls back to the two-pass path)
            StarChunk* ch = p.chunks + (size_t)prob * p.chunk_share + slot;
            ch->i = i; ch->cnt = c; ch->seq = seq; ch->pad = 0;
#pragma unroll 1
            for (uint32_t t = 0; t < c; ++t) ch->j[t] = hitbuf[t][threadIdx.x];
        }
        ++seq;
    };
    for (uint32_t j0 = 0; j0 < last; j0 += 32) {   // (rows are padded to cap: a trip may read up to 31 floats past `last`, masked below)
        uint32_t bits = screen ? 0u : 0xFFFFFFFFu;   // (no usable screen -- coordinates beyond binary32's range --: every pair gets the exact test)
        if (screen)
#pragma unroll
        for (int t = 0; t < 32; t += 2) {
            sw_f32x2 s;
#pragma unroll
            for (int k = 0; k < DIM; ++k) {
                const sw_f32x2 cj = *(cfloat2_ptr)(t32 + (size_t)k * cap + j0 + t);   // wave-uniform address
                const sw_f32x2 e = cj - xf[k];
                s = k == 0 ? e * e : __builtin_elementwise_fma(e, e, s);
            }
            // two verdicts shifted into the lane's bit string: compare into vcc, add-with-carry of the string to itself
            asm("v_cmp_le_f32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_cmp_le_f32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(bits) : "v"(s[0]), "v"(s[1]), "v"(thr32) : "vcc");
        }
        if (screen) bits = __brev(bits);   // verdict t back at bit t
        // only nodes before this lane's own count (j < i)
        const uint32_t lim = (act && i > j0) ? (i - j0 >= 32u ? 0xFFFFFFFFu : ((1u << (i - j0)) - 1u)) : 0u;
        bits &= lim;
        if (__ballot(bits != 0) == 0) continue;
        uint32_t ebits = 0;   // the trip's neighbours (the exact test's verdicts): kept apart from the loads, appended below
        for (; bits != 0; bits &= bits - 1) {   // ascending j
            const uint32_t b = (uint32_t)(__ffs((int)bits) - 1), j = j0 + b;
            double c[DIM];
#pragma unroll
            for (int k = 0; k < DIM; ++k) c[k] = tree[(size_t)k * cap + j];
            const double d2 = dist2<DIM>(x, c, DIM);   // distance(node.state, tree[j].state), rrt_star.rs:125
            if (d2 <= thr) {
                if (FILL) out[cnt++] = StarEntry{j, i, d2};
                else ebits |= 1u << b;
            }
        }
        if (!FILL) {
            cnt += (uint32_t)__popc(ebits);
            for (; ebits != 0; ebits &= ebits - 1) {
                hitbuf[nbuf][threadIdx.x] = (uint16_t)(j0 + (uint32_t)(__ffs((int)ebits) - 1));
                if (++nbuf == (uint32_t)kChunk) { flush(nbuf); nbuf = 0; }
            }
        }
    }
    if (!FILL) {   // the lanes' last, partly filled chunks: one cursor bump for the wave
        const bool part = act && nbuf != 0;
        const uint64_t pm = __ballot(part);
        if (pm != 0) {
            uint32_t base0 = 0;
            if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((unsigned long long)pm) - 1)) base0 = atomicAdd(&p.chunk_cursor[prob], (uint32_t)__popcll(pm));
            base0 = (uint32_t)__shfl((int)base0, __ffsll((unsigned long long)pm) - 1, 64);
            const uint32_t slot = base0 + (uint32_t)__popcll(pm & ((1ull << (threadIdx.x & 63u)) - 1ull));
            if (part && slot < p.chunk_share) {
                StarChunk* ch = p.chunks + (size_t)prob * p.chunk_share + slot;
                ch->i = i; ch->cnt = nbuf; ch->seq = seq; ch->pad = 0;
#pragma unroll 1
                for (uint32_t t = 0; t < nbuf; ++t) ch->j[t] = hitbuf[t][threadIdx.x];
            }
        }
        if (act) p.nbr_cnt[(size_t)prob * cap + i] = cnt;
    }
    if (act) {   // distance(q_new, q_near), rrt_star.rs:228: the nearest node is the parent the RRT kernel recorded
        const uint32_t nearest = (uint32_t)p.parent[(size_t)prob * cap + i];
        double c[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) c[k] = tree[(size_t)k * cap + nearest];
        p.d_near[(size_t)prob * cap + i] = sqrt(dist2<DIM>(x, c, DIM));
    }
}

// ---- exclusive prefix of the pending nodes' counts, and how many of them fit the problem's pool segment this round
__global__ __launch_bounds__(256) void star_scan_kernel(DevParams p) {
    const uint32_t prob = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = p.state[prob].n_nodes, w0 = p.wired[prob];
    const uint32_t pending = n > w0 ? n - w0 : 0u;
    __shared__ unsigned long long part[256];
    __shared__ uint32_t first_over;
    if (tid == 0) first_over = pending;
    const size_t base = (size_t)prob * p.cap;
    const uint32_t seg = (pending + 255u) / 256u;
    const uint32_t a = tid * seg < pending ? tid * seg : pending;
    const uint32_t b = a + seg < pending ? a + seg : pending;
    unsigned long long s = 0;
    for (uint32_t t = a; t < b; ++t) s += p.nbr_cnt[base + w0 + t];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {   // 256 partial sums: a serial pass is cheaper than it looks next to the kernels around it
        unsigned long long run = 0;
        for (int t = 0; t < 256; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    unsigned long long run = part[tid];
    for (uint32_t t = a; t < b; ++t) {
        const uint32_t c = p.nbr_cnt[base + w0 + t];
        if (run + c > (unsigned long long)p.pool_share) { atomicMin(&first_over, t); break; }   // (a list is at most cap <= pool_share long)
        p.nbr_off[base + w0 + t] = (uint32_t)run;
        run += c;
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t take = first_over;
        p.nbr_take[prob] = take;
        // entries of the nodes taken this round (the barrier above makes the block's offsets visible)
        p.nbr_total[prob] = take == 0 ? 0u : p.nbr_off[base + w0 + take - 1] + p.nbr_cnt[base + w0 + take - 1];
    }
}

// check_motion(a, b) and check_motion(b, a) at once (rrt_star.rs:73-104 twice): the two motions share their length, hence
// their step count, and -- every interpolated state of either lies within dist/2 (+ rounding) of the midpoint -- ONE pass of
// the conservative midpoint filter of motion_seq.hpp (margins: 1e-6 relative, 1e-9 |coordinates| absolute, against a few ulps
// between the two directions' midpoints); only the spheres it leaves are stepped, direction by direction, with the
// reference's own interpolation a + (b - a) s/n  resp.  b + (a - b) s/n.
// ---- 1b. the counting pass's chunks -> the nodes' lists: entry t of a chunk goes to offset(node) + 32 * ordinal + t as (j, i, -)
__global__ __launch_bounds__(256) void star_compact_kernel(DevParams p) {
    const uint32_t prob = blockIdx.y;
    const uint32_t nch = p.chunk_cursor[prob] < p.chunk_share ? p.chunk_cursor[prob] : p.chunk_share;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= nch * (uint32_t)kChunk) return;
    const StarChunk* ch = p.chunks + (size_t)prob * p.chunk_share + e / (uint32_t)kChunk;
    const uint32_t t = e % (uint32_t)kChunk;
    if (t >= ch->cnt) return;
    const uint32_t i = ch->i;
    p.pool[(size_t)prob * p.pool_share + p.nbr_off[(size_t)prob * p.cap + i] + ch->seq * (uint32_t)kChunk + t] = StarEntry{ch->j[t], i, 0.0};
}

// ---- 2. a lane per neighbour pair, the round's pool segment taken linearly (full waves whatever the list lengths): the
// distance and the validity of both motions, motion_both() with the sphere table in LDS (every lane reads the same word).
template <int DIM>
__global__ __launch_bounds__(256) void star_edges_kernel(DevParams p) {
    __shared__ double sc[DIM][64], srad[64], sthr[64];
    const uint32_t prob = blockIdx.y, tid = threadIdx.x;
    // the launch covers segment seg_index of seg_count of the problem's entries (the wiring of a segment runs on a second
    // stream while the next segment's pairs are checked here)
    const uint32_t total = p.nbr_total[prob];
    const uint32_t e_lo = (uint32_t)((uint64_t)total * p.seg_index / p.seg_count);
    const uint32_t e_hi = (uint32_t)((uint64_t)total * (p.seg_index + 1u) / p.seg_count);
    if (e_lo + blockIdx.x * 256u >= e_hi) return;
    const uint32_t e = e_lo + blockIdx.x * 256u + tid;
    const bool act = e < e_hi;
    const size_t cap = p.cap;
    const double* __restrict__ tree = p.tree + (size_t)prob * DIM * cap;
    StarEntry* ent = p.pool + (size_t)prob * p.pool_share + (act ? e : 0u);
    StarEntry en = *ent;   // (j, i, -)
    double a[DIM], b[DIM];   // a = the neighbour x_j, b = the node x_i
#pragma unroll
    for (int k = 0; k < DIM; ++k) { a[k] = tree[(size_t)k * cap + en.j]; b[k] = tree[(size_t)k * cap + en.flags]; }
    en.d = dist2<DIM>(b, a, DIM);   // distance(node.state, tree[j].state), rrt_star.rs:125: the search's own expression
    const double dist = sqrt(en.d);   // distance(x_i, x_j): symmetric in its arguments, bit for bit
    bool ab = true, ba = true;        // check_motion(neighbour, q_new) :235 / check_motion(new, neighbour) :271
    const uint32_t ns = p.n_spheres, nb = p.n_boxes;
    if (ns + nb != 0) {
        const uint32_t nsteps = num_steps_u32(dist, p.res);
        const bool single = nsteps <= 1;   // is_valid(to) only
        const double dn = (double)nsteps;
        double mid[DIM];
        lerp<DIM>(a, b, 0.5, mid, DIM);
        const double h = 0.5 * dist * (1.0 + 1e-6) + p.filt_abs;
        for (uint32_t w0 = 0; w0 < ns; w0 += 64) {
            const uint32_t cnt = ns - w0 < 64u ? ns - w0 : 64u;
            __syncthreads();   // the previous chunk has been consumed
            if (tid < cnt) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) sc[k][tid] = p.sph_c[(size_t)k * ns + w0 + tid];
                srad[tid] = p.sph_r[w0 + tid];
                sthr[tid] = p.sph_thr[w0 + tid];
            }
            __syncthreads();
            if (!act) continue;
            if (single) {
                for (uint32_t jj = 0; jj < cnt; ++jj) {
                    double c[DIM];
#pragma unroll
                    for (int k = 0; k < DIM; ++k) c[k] = sc[k][jj];
                    if (!(dist2<DIM>(c, b, DIM) > sthr[jj])) ab = false;
                    if (!(dist2<DIM>(c, a, DIM) > sthr[jj])) ba = false;
                }
                continue;
            }
            uint64_t mask = 0;
            bool looked_up = false;
            if (DIM <= 3 && w0 == 0 && p.star_sph_grid != nullptr) {
                // the first 64 spheres in R^2 / R^3: the midpoint's cell of the mask grid (rrt_cells.hip, sphere_grid_kernel; built
                // for the longest motion a neighbour pair can have) names a superset of what the loop below would keep
                const uint32_t G = p.sph_grid_G;
                bool inside = true;
                uint32_t ci[3] = {0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < DIM; ++k) {
                    const double u = (mid[k] - p.lo[k]) * ((double)G / (p.hi[k] - p.lo[k]));
                    inside = inside && u >= -0x1p-10 && u <= (double)G + 0x1p-10;   // (NaN: outside)
                    const double fl = floor(u);
                    ci[k] = fl > 0.0 ? (fl < (double)G ? (uint32_t)fl : G - 1u) : 0u;
                }
                if (inside) {
                    mask = p.star_sph_grid[DIM == 3 ? (ci[2] * G + ci[1]) * G + ci[0] : ci[1] * G + ci[0]];
                    if (cnt < 64u) mask &= (1ull << cnt) - 1ull;
                    looked_up = true;
                }
            }
            if (!looked_up) {
                for (uint32_t jj = 0; jj < cnt; ++jj) {
                    double c[DIM];
#pragma unroll
                    for (int k = 0; k < DIM; ++k) c[k] = sc[k][jj];
                    const double rr = srad[jj] + h;
                    const double lim = rr * rr * (1.0 + 1e-9);
                    if (!(dist2<DIM>(c, mid, DIM) > lim)) mask |= 1ull << jj;   // NaN / inf radii stay in
                }
            }
            if (mask == 0) continue;
            // second, tighter filter for the few spheres the midpoint test left: every interpolated state of either motion is a
            // point of the segment [a, b] up to a few ulps, so a sphere whose distance to the SEGMENT exceeds r (+ the same kind
            // of margin) cannot be hit either.  (A computed closest point is still a point near the segment: its distance can
            // only overestimate the minimum by the square of the parameter's rounding error -- far below the margin.)
            {
                const double inv_len2 = 1.0 / en.d;   // (nsteps > 1: a != b)
                const double h2 = 1e-6 * dist + p.filt_abs;
                for (uint64_t m = mask; m != 0; m &= m - 1) {
                    const uint32_t o = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                    double dot = 0.0;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) dot += (sc[k][o] - a[k]) * (b[k] - a[k]);
                    double ts = dot * inv_len2;
                    ts = ts < 0.0 ? 0.0 : (ts > 1.0 ? 1.0 : ts);   // (NaN stays NaN: the sphere stays in)
                    double d2s = 0.0;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) {
                        const double e = sc[k][o] - (a[k] + ts * (b[k] - a[k]));
                        d2s += e * e;
                    }
                    const double rr = srad[o] + h2;
                    if (d2s > rr * rr * (1.0 + 1e-9)) mask &= ~(1ull << o);
                }
            }
            if (mask == 0) continue;
            for (uint32_t step = 1; step <= nsteps && (ab || ba); ++step) {
                const double t = (double)step / dn;
                double s1[DIM], s2[DIM];
                lerp<DIM>(a, b, t, s1, DIM);
                lerp<DIM>(b, a, t, s2, DIM);
                for (uint64_t m = mask; m != 0; m &= m - 1) {
                    const uint32_t o = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                    double c[DIM];
#pragma unroll
                    for (int k = 0; k < DIM; ++k) c[k] = sc[k][o];
                    if (!(dist2<DIM>(c, s1, DIM) > sthr[o])) ab = false;
                    if (!(dist2<DIM>(c, s2, DIM) > sthr[o])) ba = false;
                }
            }
        }
        if (nb != 0 && act) {   // boxes are never filtered
            if (single) {
                for (uint32_t bx = 0; bx < nb; ++bx) {
                    if (obstacle_hit<DIM>(p, DIM, b, ns + bx)) ab = false;
                    if (obstacle_hit<DIM>(p, DIM, a, ns + bx)) ba = false;
                }
            } else {
                for (uint32_t step = 1; step <= nsteps && (ab || ba); ++step) {
                    const double t = (double)step / dn;
                    double s1[DIM], s2[DIM];
                    lerp<DIM>(a, b, t, s1, DIM);
                    lerp<DIM>(b, a, t, s2, DIM);
                    for (uint32_t bx = 0; bx < nb; ++bx) {
                        if (ab && obstacle_hit<DIM>(p, DIM, s1, ns + bx)) ab = false;
                        if (ba && obstacle_hit<DIM>(p, DIM, s2, ns + bx)) ba = false;
                    }
                }
            }
        }
    }
    if (act) {
        en.flags = (ab ? 1u : 0u) | (ba ? 2u : 0u);
        en.d = dist;
        *ent = en;
    }
}

// ---- 3. one wave per problem: rrt_star.rs:225-282 for the nodes [wired, wired + take) in insertion order.  A node costs one
// memory round trip: its metadata comes 64 nodes at a time (a lane per node, read back with v_readlane), its first 64
// neighbour entries were requested while the node before it was being wired (entries never change), so only the costs --
// which the node before it may just have rewired -- are fetched on the spot.  Loads after stores of the same wave see
// them (vector memory operations of a wave execute in order; the stores write through the CU's own L1).
__global__ __launch_bounds__(64) void star_wire_kernel(DevParams p) {
    const uint32_t prob = blockIdx.x, lane = threadIdx.x;
    const uint32_t wired0 = p.wired[prob], take = p.nbr_take[prob];
    if (take == 0) return;
    const size_t cap = p.cap, base = (size_t)prob * cap;
    // segment seg_index of seg_count: the nodes whose lists END inside the entries checked so far -- lists are contiguous
    // and ascending, so that is a prefix of the round's nodes, found by bisection on the lists' ends
    const uint32_t total = p.nbr_total[prob];
    auto nodes_within = [&](uint32_t e_end) {   // how many of the round's nodes have their whole list below entry e_end
        uint32_t lo = 0, hi = take;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (p.nbr_off[base + wired0 + mid] + p.nbr_cnt[base + wired0 + mid] <= e_end) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const bool last_seg = p.seg_index + 1u == p.seg_count;
    const uint32_t n_lo = p.seg_index == 0 ? 0u : nodes_within((uint32_t)((uint64_t)total * p.seg_index / p.seg_count));
    const uint32_t n_hi = last_seg ? take : nodes_within((uint32_t)((uint64_t)total * (p.seg_index + 1u) / p.seg_count));
    const uint32_t w0 = wired0 + n_lo, end = wired0 + n_hi;
    double* cost = p.cost + base;
    int32_t* parent = p.parent + base;
    const StarEntry* pool = p.pool + (size_t)prob * p.pool_share;
    uint64_t W = uni64(p.wire_chk[prob]);   // (wave-uniform, and said so: the checksum's 64-bit multiplies then run on the scalar unit)
    // an entry as four dwords in registers (a struct copied through a select of two addresses ends up in memory, and the
    // "prefetch" then waits for its own load: measured, 2.2 us per node).  Lanes past the list read a valid address and are masked.
    auto fetch = [](const StarEntry* list, uint32_t idx) { return *reinterpret_cast<const uint4*>(list + idx); };
    for (uint32_t i0 = w0; i0 < end; i0 += 64) {
        const uint32_t t = i0 + lane;
        const bool mine = t < end;
        const uint32_t m_cnt = mine ? p.nbr_cnt[base + t] : 0u, m_off = mine ? p.nbr_off[base + t] : 0u;
        const uint32_t m_near = mine ? (uint32_t)parent[t] : 0u;   // still the nearest node: the RRT kernel's parent
        const double m_dn = mine ? p.d_near[base + t] : 0.0;
        const uint32_t nb = end - i0 < 64u ? end - i0 : 64u;
        uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)m_cnt, 0);
        const StarEntry* list = pool + (uint32_t)__builtin_amdgcn_readlane((int)m_off, 0);
        uint4 nxt = fetch(list, lane < cnt ? lane : 0u);
        for (uint32_t u = 0; u < nb; ++u) {
            const uint32_t i = i0 + u;
            const uint4 cur = nxt;   // entries [0, 64) of node i
            const uint32_t nearest = (uint32_t)__builtin_amdgcn_readlane((int)m_near, (int)u);
            const double dn = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m_dn), (int)u), __builtin_amdgcn_readlane(__double2loint(m_dn), (int)u));
            const StarEntry* list_i = list;
            const uint32_t cnt_i = cnt;
            if (u + 1 < nb) {   // the next node's first entries: requested now, used in the next trip
                cnt = (uint32_t)__builtin_amdgcn_readlane((int)m_cnt, (int)(u + 1));
                list = pool + (uint32_t)__builtin_amdgcn_readlane((int)m_off, (int)(u + 1));
                nxt = fetch(list, lane < cnt ? lane : 0u);
            }
            // 6. choose parent: cost(temp_node, q_near_node) first (:228), then the neighbours in ascending index with the running minimum
            const bool has0 = lane < cnt_i;
            const double cj0 = cost[has0 ? cur.x : 0u];
            const double c0 = unid(cost[nearest] + dn);
            double best_c = c0;
            uint32_t best_j = nearest;
            for (uint32_t e0 = 0; e0 < cnt_i; e0 += 64) {
                const bool has = e0 + lane < cnt_i;
                const uint4 en = e0 == 0 ? cur : fetch(list_i, has ? e0 + lane : 0u);
                const double cj = e0 == 0 ? cj0 : cost[has ? en.x : 0u];
                const double c = cj + __hiloint2double((int)en.w, (int)en.z);   // cost(temp_node, neighbour), :104-113
                const bool cand = has && (en.y & 1u) != 0 && c < best_c;       // strict: an equal cost keeps the earlier choice
                uint64_t m = __ballot(cand);
                if (m == 0) continue;                        // (usual for a node whose nearest node is also its best parent)
                if ((m & (m - 1)) != 0) {                    // several candidates: the cheapest, lowest index among equals (the list is ascending)
                    const double cm = wave_min_f64(cand ? c : __builtin_inf());
                    m = __ballot(cand && c == cm);
                }
                const int l = __ffsll((unsigned long long)m) - 1;
                best_c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), l), __builtin_amdgcn_readlane(__double2loint(c), l));
                best_j = (uint32_t)__builtin_amdgcn_readlane((int)en.x, l);
            }
            // 7. push: parent and cost of the new node (:244-250)
            if (lane == 0) { parent[i] = (int32_t)best_j; cost[i] = best_c; }
            // 8. rewire (:253-282)
            uint64_t rew_cnt = 0, rew_sum = 0;
            for (uint32_t e0 = 0; e0 < cnt_i; e0 += 64) {
                const bool has = e0 + lane < cnt_i;
                const uint4 en = e0 == 0 ? cur : fetch(list_i, has ? e0 + lane : 0u);
                const double cj = e0 == 0 ? cj0 : cost[has ? en.x : 0u];
                const double c2 = best_c + __hiloint2double((int)en.w, (int)en.z);   // cost(neighbour, new_node), :265
                const bool rw = has && en.x != best_j && c2 < cj && (en.y & 2u) != 0;
                if (rw) { parent[en.x] = (int32_t)i; cost[en.x] = c2; }
                uint64_t rm = __ballot(rw);
                rew_cnt += (uint64_t)__popcll(rm);
                for (; rm != 0; rm &= rm - 1)   // (a few lanes at most)
                    rew_sum += (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)en.x, __ffsll((unsigned long long)rm) - 1);
            }
            uint64_t w = fnv_mix(kFnvBasis, (uint64_t)uni(best_j));
            w = fnv_mix(w, uni64((uint64_t)__double_as_longlong(best_c)));
            w = fnv_mix(w, rew_cnt);
            w = fnv_mix(w, rew_sum);
            W = W * kFnvPrime + w;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // program order for the compiler; the hardware keeps a wave's memory operations in order
        }
    }
    if (lane == 0) {
        p.wire_chk[prob] = W;
        if (last_seg) p.wired[prob] = end;   // (the cursor moves once the whole round is wired: the edge launches read it)
    }
}

bool star_wire_supported(uint32_t dim) { return dim >= 2 && dim <= 6; }

template <bool FILL>
static void launch_pairs(const DevParams& p, uint32_t max_nodes, hipStream_t stream) {
    if (max_nodes == 0) return;
    dim3 grid((max_nodes + kPairThreads - 1) / kPairThreads, p.n_problems), block(kPairThreads);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL((star_pairs_kernel<2, FILL>), grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL((star_pairs_kernel<3, FILL>), grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL((star_pairs_kernel<4, FILL>), grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL((star_pairs_kernel<5, FILL>), grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL((star_pairs_kernel<6, FILL>), grid, block, 0, stream, p); break;
        default: break;
    }
}
void launch_star_shadow(const DevParams& p, uint32_t max_nodes, hipStream_t stream) {
    if (max_nodes == 0) return;
    dim3 grid((max_nodes + 255) / 256, p.n_problems), block(256);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL(star_shadow_kernel<2>, grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL(star_shadow_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(star_shadow_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(star_shadow_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(star_shadow_kernel<6>, grid, block, 0, stream, p); break;
        default: break;
    }
}
void launch_star_count(const DevParams& p, uint32_t max_pending, hipStream_t stream) { launch_pairs<false>(p, max_pending, stream); }
void launch_star_fill(const DevParams& p, uint32_t max_take, hipStream_t stream) { launch_pairs<true>(p, max_take, stream); }
void launch_star_scan(const DevParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(star_scan_kernel, dim3(p.n_problems), dim3(256), 0, stream, p);
}
void launch_star_edges(const DevParams& p, uint32_t max_total, hipStream_t stream) {
    if (max_total == 0) return;
    dim3 grid((max_total + 255) / 256, p.n_problems), block(256);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL(star_edges_kernel<2>, grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL(star_edges_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(star_edges_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(star_edges_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(star_edges_kernel<6>, grid, block, 0, stream, p); break;
        default: break;
    }
}
void launch_star_compact(const DevParams& p, uint32_t max_chunks, hipStream_t stream) {
    if (max_chunks == 0) return;
    hipLaunchKernelGGL(star_compact_kernel, dim3((max_chunks * (uint32_t)kChunk + 255) / 256, p.n_problems), dim3(256), 0, stream, p);
}
void launch_star_wire(const DevParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(star_wire_kernel, dim3(p.n_problems), dim3(64), 0, stream, p);
}

}  // namespace oxhip
