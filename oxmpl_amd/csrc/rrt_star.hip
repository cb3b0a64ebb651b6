// rrt_star.hip -- RRT* (oxmpl/src/geometric/planners/rrt_star.rs:170-289) on the GPU: one 256-thread
// workgroup per problem, tree + cost-to-come as SoA arrays in HBM / L2.
//
// One iteration of the reference:  sample -> nearest -> steer -> check_motion(q_near, q_new) [continue if
// invalid] -> find_neighbours (every node with distance < search_radius, :121-131) -> choose parent
// (:225-241) -> push (:244-250) -> rewire (:253-282) -> goal test (:285-288).
//
// What is sequential in the reference and how it is evaluated here (is_valid is pure, so evaluating a
// motion the reference skipped, or skipping one whose outcome cannot matter, changes nothing):
//  * choose parent walks the neighbours in index order and takes one when `cost < min_cost && motion ok`;
//    its result is the lexicographic minimum of (cost, index) over the neighbours whose motion is valid
//    and whose cost is below the nearest node's -- found by trying candidates in increasing (cost, index)
//    order until one motion is valid;
//  * every rewire decision depends only on that neighbour's own cost and the new node's (rrt_star.rs:267-281;
//    nothing is propagated to descendants, :278-280), so the neighbours are rewired independently, one wave each.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

constexpr int kStarThreads = 256;
constexpr int kStarWaves = kStarThreads / 64;

struct StarShared {
    uint32_t rng_buf[16][64];
    Best wave_best[kStarWaves];
    Exact wave_exact[kStarWaves];
    uint32_t nb_count;
    uint32_t shadow_word;
    uint32_t rew_cnt;
    unsigned long long rew_sum;
};

template <int DIM>
__global__ __launch_bounds__(kStarThreads, DIM ? 4 : 1) void rrt_star_kernel(DevParams p_in) {
    constexpr int D = DIM ? DIM : kMaxDim;
    const int dim = DIM ? DIM : (int)p_in.dim;
    const uint32_t prob = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ StarShared sh;
    __shared__ ObsLds obs;
    const DevParams p = stage_obstacles(p_in, obs, tid, kStarThreads);   // visible after the first barrier of iteration 0

    ProblemState st = p.state[prob];
    if (p.stop_at_goal && st.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * p.dim * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    double* cost = p.cost + (size_t)prob * cap;
    uint32_t* nb_idx = p.nb_idx + (size_t)prob * cap;     // find_neighbours' result for this iteration (unordered)
    double* nb_dist = p.nb_dist + (size_t)prob * cap;     // distance(q_new, neighbour)
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) goal_c[k] = p.goal_c[(size_t)prob * p.dim + k];
    const double goal_thr = p.goal_thr[prob];

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);

    uint32_t n = st.n_nodes;
    uint64_t wire = p.wire_chk[prob];
    // binary32 shadow of the tree: both scans of an iteration screen over it (rrt_device.hpp) and decide in binary64
    float* tree32 = p.tree32 + (size_t)prob * p.dim * cap;
    const ScreenMargins mg = screen_margins(shadow_sync<D>(p, dim, prob, tree, tree32, cap, n, goal_c, &sh.shadow_word, tid, kStarThreads), dim);
    // find_neighbours' radius test d2 <= thr_search, screened: a node whose binary32 d2 exceeds this cannot pass it
    const float nb_screen = screen_threshold(mg, sqrt(p.thr_search));

    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    for (uint64_t it = 0; it < p.budget; ++it) {
        if (n >= p.max_nodes) { stop = 2; break; }

        // 2. sample (rrt_star.rs:178-186)
        double q[D];
        sample_state<D>(rng, p, dim, goal_c, q, p.goal_r[prob]);

        // 3. nearest (rrt_star.rs:189-199): binary32 screen first (see rrt_stream.hip), else d2 compare with the
        //    exact post-sqrt fallback on near-ties
        Best best = best_init();
        bool screened = false;
        if (mg.usable) {
            float qf[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) qf[k] = (float)q[k];
            float s1 = __builtin_inff(), s2 = __builtin_inff();
            uint32_t si = 0xFFFFFFFFu;
            screen_scan<D>(tree32, cap, n, dim, qf, tid, kStarThreads, [&](uint32_t i, float s) {
                s2 = __builtin_amdgcn_fmed3f(s, s1, s2);
                const bool lt = s < s1;
                s1 = lt ? s : s1;
                si = lt ? i : si;
            });
            best = best_wave_reduce(Best{(double)s1, (double)s2, si});
            if (lane == 0) sh.wave_best[wave] = best;
            if (tid == 0) { sh.nb_count = 0; sh.rew_cnt = 0; sh.rew_sum = 0; }
            __syncthreads();
            best = sh.wave_best[0];
#pragma unroll
            for (int w = 1; w < kStarWaves; ++w) best = best_combine(best, sh.wave_best[w]);
            screened = screen_clear(mg, best.b1, best.b2);
            if (!screened) __syncthreads();   // everyone has read wave_best before the binary64 scan rewrites it
        }
        if (!screened) {
            best = best_init();
            for (uint32_t i = tid; i < n; i += kStarThreads) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + i];
                best_push(best, dist2<D>(c, q, dim), i);
            }
            best = best_wave_reduce(best);
            if (lane == 0) sh.wave_best[wave] = best;
            if (tid == 0) { sh.nb_count = 0; sh.rew_cnt = 0; sh.rew_sum = 0; }
            __syncthreads();
            best = sh.wave_best[0];
#pragma unroll
            for (int w = 1; w < kStarWaves; ++w) best = best_combine(best, sh.wave_best[w]);
        }
        uint32_t nearest;
        double min_dist;
        if (screened) {
            nearest = best.i1;
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + nearest];
            min_dist = sqrt(dist2<D>(c, q, dim));
        } else if (best_ambiguous(best)) {
            Exact e{__builtin_inf(), 0xFFFFFFFFu};
            for (uint32_t i = tid; i < n; i += kStarThreads) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + i];
                const double d = sqrt(dist2<D>(c, q, dim));
                if (d < e.dist) { e.dist = d; e.idx = i; }
            }
            e = exact_wave_reduce(e);
            if (lane == 0) sh.wave_exact[wave] = e;
            __syncthreads();
            e = sh.wave_exact[0];
#pragma unroll
            for (int w = 1; w < kStarWaves; ++w) e = exact_combine(e, sh.wave_exact[w]);
            nearest = e.idx;
            min_dist = e.dist;
            __syncthreads();  // wave_exact is reused by the choose-parent reductions below
        } else {
            nearest = best.i1;
            min_dist = sqrt(best.b1);
        }
        nearest = uni(nearest);
        double q_near[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q_near[k] = tree[(size_t)k * cap + nearest];

        // 4. steer (rrt_star.rs:203-209)
        double q_new[D];
        if (min_dist > p.max_distance) {
            const double t = p.max_distance / min_dist;
            lerp<D>(q_near, q, t, q_new, dim);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) q_new[k] = q[k];
        }

        // 5. check_motion(q_near, q_new) (rrt_star.rs:212-214)
        const bool bad = motion_invalid_wg<D>(p, dim, q_near, q_new, tid, kStarThreads);
        const bool ok = !__syncthreads_or(bad ? 1 : 0);
        // checksum = H + W: H, the iteration polynomial of RRT's digests, lives in the problem state;
        // W, the wiring polynomial, in wire_chk
        st.checksum = chk_push(st.checksum, iter_digest<D>(nearest, q_new, dim, ok));
        st.iterations++;
        if (!ok) continue;
        st.accepted++;

        // 6a. cost via the nearest node: cost(temp_node, q_near_node) = q_near.cost + distance(q_new, q_near) (:104-113, :228)
        uint32_t best_parent = nearest;
        double min_cost = cost[nearest] + sqrt(dist2<D>(q_new, q_near, dim));
        const double init_cost = min_cost;

        // find_neighbours (rrt_star.rs:121-131): distance(q_new, tree[i]) < search_radius  <=>  d2 <= thr_search.
        // The same pass already finds the first choose-parent candidate: the lexicographic minimum of
        // (cost via neighbour, index) among the neighbours cheaper than the nearest node.
        Exact m{__builtin_inf(), 0xFFFFFFFFu};
        auto neighbour = [&](uint32_t i) {   // the reference's test and bookkeeping for node i, in binary64
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + i];
            const double d2 = dist2<D>(q_new, c, dim);
            if (d2 <= p.thr_search) {
                const uint32_t slot = atomicAdd(&sh.nb_count, 1u);
                const double d = sqrt(d2);
                nb_idx[slot] = i;
                nb_dist[slot] = d;
                const double cv = cost[i] + d;
                if (cv < init_cost && (cv < m.dist || (cv == m.dist && i < m.idx))) { m.dist = cv; m.idx = i; }
            }
        };
        if (mg.usable && nb_screen < __builtin_inff()) {
            float qf[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) qf[k] = (float)q_new[k];
            screen_scan_below<D>(tree32, cap, n, dim, qf, nb_screen, tid, kStarThreads, neighbour);
        } else {
            for (uint32_t i = tid; i < n; i += kStarThreads) neighbour(i);
        }
        m = exact_wave_reduce(m);
        if (lane == 0) sh.wave_exact[wave] = m;
        __syncthreads();
        const uint32_t nbc = sh.nb_count;
        m = sh.wave_exact[0];
#pragma unroll
        for (int w = 1; w < kStarWaves; ++w) m = exact_combine(m, sh.wave_exact[w]);

        // 6b. choose parent (rrt_star.rs:225-241): candidates in increasing (cost, index) order until a motion is valid
        while (m.idx != 0xFFFFFFFFu) {
            double from[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) from[k] = tree[(size_t)k * cap + m.idx];
            const bool bad2 = motion_invalid_wg<D>(p, dim, from, q_new, tid, kStarThreads);
            if (!__syncthreads_or(bad2 ? 1 : 0)) {   // check_motion(neighbour, q_new) holds: this is the parent
                best_parent = m.idx;
                min_cost = m.dist;
                break;
            }
            // rare: that motion is blocked; the next candidate is the smallest (cost, index) above it
            const double last_c = m.dist;
            const uint32_t last_i = m.idx;
            m = Exact{__builtin_inf(), 0xFFFFFFFFu};
            for (uint32_t e = tid; e < nbc; e += kStarThreads) {
                const uint32_t idx = nb_idx[e];
                const double cv = cost[idx] + nb_dist[e];
                const bool after = cv > last_c || (cv == last_c && idx > last_i);
                if (cv < init_cost && after && (cv < m.dist || (cv == m.dist && idx < m.idx))) { m.dist = cv; m.idx = idx; }
            }
            m = exact_wave_reduce(m);
            if (lane == 0) sh.wave_exact[wave] = m;
            __syncthreads();
            m = sh.wave_exact[0];
#pragma unroll
            for (int w = 1; w < kStarWaves; ++w) m = exact_combine(m, sh.wave_exact[w]);
        }

        // 7. push (rrt_star.rs:244-250)
        const uint32_t new_idx = n;
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                if (k < dim) {
                    tree[(size_t)k * cap + n] = q_new[k];
                    tree32[(size_t)k * cap + n] = (float)q_new[k];
                }
            }
            parent[n] = (int32_t)best_parent;
            cost[n] = min_cost;
        }
        ++n;

        // 8. rewire (rrt_star.rs:253-282), one wave per neighbour
        uint32_t my_cnt = 0;
        unsigned long long my_sum = 0;
        for (uint32_t e = wave; e < nbc; e += kStarWaves) {
            const uint32_t idx = nb_idx[e];
            if (idx == best_parent) continue;                        // :258-260
            const double c2 = min_cost + nb_dist[e];                 // cost(neighbour, new_node), :265
            if (!(c2 < cost[idx])) continue;
            double to[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) to[k] = tree[(size_t)k * cap + idx];
            const bool bad3 = motion_invalid_partial<D>(p, dim, q_new, to, lane, 64);   // check_motion(new, neighbour)
            if (__ballot(bad3) != 0) continue;
            if (lane == 0) {
                parent[idx] = (int32_t)new_idx;
                cost[idx] = c2;
            }
            ++my_cnt;
            my_sum += idx;
        }
        if (lane == 0 && my_cnt) {
            atomicAdd(&sh.rew_cnt, my_cnt);
            atomicAdd(&sh.rew_sum, my_sum);
        }
        __syncthreads();   // the new node, the rewired parents / costs and the counters are visible
        {
            uint64_t w = fnv_mix(kFnvBasis, (uint64_t)best_parent);
            w = fnv_mix(w, (uint64_t)__double_as_longlong(min_cost));
            w = fnv_mix(w, (uint64_t)sh.rew_cnt);
            w = fnv_mix(w, (uint64_t)sh.rew_sum);
            wire = wire * kFnvPrime + w;
        }

        // 9. goal test (rrt_star.rs:285-288)
        if (dist2<D>(q_new, goal_c, dim) <= goal_thr) {
            if (st.goal_node < 0) st.goal_node = (int32_t)new_idx;
            if (p.stop_at_goal) { stop = 0; break; }
        }
        __syncthreads();   // everyone has read the counters before the next iteration resets them
    }

    if (tid == 0) {
        st.n_nodes = n;
        st.draws = rng.pos;
        st.stop_reason = stop;
        p.state[prob] = st;
        p.wire_chk[prob] = wire;
        p.shadow_state[2 * (size_t)prob] = n;
        p.shadow_state[2 * (size_t)prob + 1] = sh.shadow_word;
    }
}

void launch_rrt_star(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kStarThreads);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL(rrt_star_kernel<2>, grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL(rrt_star_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(rrt_star_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(rrt_star_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(rrt_star_kernel<6>, grid, block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL(rrt_star_kernel<7>, grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL(rrt_star_kernel<0>, grid, block, 0, stream, p); break;
    }
}

}  // namespace oxhip
