// rrt_connect.hip -- RRTConnect (oxmpl/src/geometric/planners/rrt_connect.rs:121-159,227-309) on the GPU:
// one 256-thread workgroup per problem, every extend() a nearest-neighbour scan + steer + striped motion
// check, exactly the primitives of rrt_stream.hip.  Iteration k sees the trees left by iterations < k; per
// iteration up to two scans.
//
// RRTConnect trees are small (tens to hundreds of nodes) and an iteration is a chain of dependent reads, so
// it is latency, not bandwidth: the first kLdsBytes / (2 * dim * 8) nodes of BOTH trees live in LDS (768 per
// tree in R^3) next to their SoA arrays in HBM, which only receive the stores (for get_tree / resume) and
// serve the nodes beyond that capacity.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

#ifndef OXHIP_CONN_THREADS
#define OXHIP_CONN_THREADS 256
#endif
constexpr int kConnThreads = OXHIP_CONN_THREADS;
constexpr int kConnWaves = kConnThreads / 64;

constexpr int kConnLdsBytes = 36864;   // LDS for the two trees of one problem; 4 workgroups per CU fit in 160 KB

struct ConnShared {
    uint32_t rng_buf[16][64];
    Best wave_best[kConnWaves];
    Exact wave_exact[kConnWaves];
};

// One tree: its first N nodes mirrored in LDS ([k][i], conflict-free for the strided scan), all of it in HBM.
template <int D, int N>
struct ConnTree {
    double* g;            // SoA [dim][cap] in HBM
    double (*l)[N];       // LDS [D][N]
    size_t cap;
    __device__ __forceinline__ double get(int k, uint32_t i) const { return i < (uint32_t)N ? l[k][i] : g[(size_t)k * cap + i]; }
    __device__ __forceinline__ void put(int k, uint32_t i, double v) const {
        if (i < (uint32_t)N) l[k][i] = v;
        g[(size_t)k * cap + i] = v;
    }
};

// hand-off after an insert: the LDS copy only needs the LDS counter; the HBM copy is read back by this workgroup
// only once the tree has outgrown its LDS mirror, and only then is the (slow) wait for the store worth paying
template <int N>
__device__ __forceinline__ void conn_publish(uint32_t n_after) {
    if (n_after > (uint32_t)N) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}

// extend() of rrt_connect.rs:121-159 for the whole workgroup.  Returns 0 = motion invalid (None),
// 1 = Advanced, 2 = Reached; `nearest` and `q_new` are filled in every case.  On success the new
// node is appended at index n (thread 0 writes, a barrier makes it visible) and n is incremented.
template <int D, int N>
__device__ __forceinline__ int wg_extend(const DevParams& p, int dim, ConnShared& sh, const ConnTree<D, N>& tree, int32_t* parent,
                                         uint32_t& n, const double q[D], uint32_t& nearest, double q_new[D]) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // nearest node: d2 compare, second-smallest tracking, exact post-sqrt fallback (rrt_connect.rs:128-136)
    Best best = best_init();
    for (uint32_t i = tid; i < n; i += kConnThreads) {
        double c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.get(k, i);
        best_push(best, dist2<D>(c, q, dim), i);
    }
    best = best_wave_reduce(best);
    if (lane == 0) sh.wave_best[wave] = best;
    __syncthreads();
    best = sh.wave_best[0];
#pragma unroll
    for (int w = 1; w < kConnWaves; ++w) best = best_combine(best, sh.wave_best[w]);
    double min_dist;
    if (best_ambiguous(best)) {
        Exact e{__builtin_inf(), 0xFFFFFFFFu};
        for (uint32_t i = tid; i < n; i += kConnThreads) {
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree.get(k, i);
            double d = sqrt(dist2<D>(c, q, dim));
            if (d < e.dist) { e.dist = d; e.idx = i; }
        }
        e = exact_wave_reduce(e);
        if (lane == 0) sh.wave_exact[wave] = e;
        __syncthreads();
        e = sh.wave_exact[0];
#pragma unroll
        for (int w = 1; w < kConnWaves; ++w) e = exact_combine(e, sh.wave_exact[w]);
        nearest = e.idx;
        min_dist = e.dist;
    } else {
        nearest = best.i1;
        min_dist = sqrt(best.b1);
    }
    nearest = uni(nearest);
    double q_near[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) q_near[k] = tree.get(k, nearest);
    int result;
    if (min_dist > p.max_distance) {  // rrt_connect.rs:140-147
        double t = p.max_distance / min_dist;
        lerp<D>(q_near, q, t, q_new, dim);
        result = 1;
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q_new[k] = q[k];
        result = 2;
    }
    const bool bad = motion_invalid_wg<D>(p, dim, q_near, q_new, tid, kConnThreads);  // rrt_connect.rs:166-189
    if (__syncthreads_or(bad ? 1 : 0)) return 0;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) tree.put(k, n, q_new[k]);
        parent[n] = (int32_t)nearest;
    }
    ++n;
    conn_publish<N>(n);
    return result;
}

template <int DIM>
__global__ __launch_bounds__(kConnThreads) void rrt_connect_kernel(DevParams p_in) {
    constexpr int D = DIM ? DIM : kMaxDim;
    const int dim = DIM ? DIM : (int)p_in.dim;
    constexpr int N = kConnLdsBytes / (2 * D * 8);   // nodes of each tree mirrored in LDS
    const uint32_t prob = blockIdx.x, tid = threadIdx.x;
    __shared__ ConnShared sh;
    __shared__ double lds_a[D][N], lds_b[D][N];
    __shared__ ObsLds obs;
    const DevParams p = stage_obstacles(p_in, obs, threadIdx.x, kConnThreads);   // the barrier below covers it

    ProblemState st = p.state[prob];
    if (st.goal_node >= 0) return;  // solved: RRTConnect::solve returned Ok

    const size_t cap = p.cap;
    const ConnTree<D, N> tree_a{p.tree + (size_t)prob * p.dim * cap, lds_a, cap};
    const ConnTree<D, N> tree_b{p.tree_b + (size_t)prob * p.dim * cap, lds_b, cap};
    int32_t* par_a = p.parent + (size_t)prob * cap;
    int32_t* par_b = p.parent_b + (size_t)prob * cap;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) goal_c[k] = p.goal_c[(size_t)prob * p.dim + k];
    const double goal_thr = p.goal_thr[prob];

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint32_t na = st.n_nodes, nb = st.n_nodes_b;
    // (re)load the LDS mirrors: a solve call continues the trees an earlier one left in HBM
    for (uint32_t i = tid; i < (na < (uint32_t)N ? na : (uint32_t)N); i += kConnThreads)
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) lds_a[k][i] = tree_a.g[(size_t)k * cap + i];
    for (uint32_t i = tid; i < (nb < (uint32_t)N ? nb : (uint32_t)N); i += kConnThreads)
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) lds_b[k][i] = tree_b.g[(size_t)k * cap + i];
    __syncthreads();
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    for (uint64_t it = 0; it < p.budget; ++it) {
        if (na >= p.max_nodes || nb >= p.max_nodes) { stop = 2; break; }
        const bool grow_start = na <= nb;  // rrt_connect.rs:249-254
        double q_rand[D];
        sample_state<D>(rng, p, dim, goal_c, q_rand);  // rrt_connect.rs:258-262
        uint32_t near_a = 0, near_b = 0;
        double q_new_a[D], q_new_b[D];
        const int ra = grow_start ? wg_extend<D, N>(p, dim, sh, tree_a, par_a, na, q_rand, near_a, q_new_a)
                                  : wg_extend<D, N>(p, dim, sh, tree_b, par_b, nb, q_rand, near_a, q_new_a);
        uint64_t h = fnv_mix(st.checksum, grow_start ? 1ull : 0ull);
        h = fnv_mix(h, (uint64_t)near_a);
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) h = fnv_mix(h, (uint64_t)__double_as_longlong(q_new_a[k]));
        h = fnv_mix(h, (uint64_t)ra);
        st.iterations++;
        bool done = false;
        if (ra) {
            const uint32_t idx_a = (grow_start ? na : nb) - 1;
            if (grow_start && dist2<D>(q_new_a, goal_c, dim) <= goal_thr) {  // rrt_connect.rs:271-274
                st.goal_node = (int32_t)idx_a;
                st.goal_node_b = -1;
                done = true;
            } else {
                const int rb = grow_start ? wg_extend<D, N>(p, dim, sh, tree_b, par_b, nb, q_new_a, near_b, q_new_b)
                                          : wg_extend<D, N>(p, dim, sh, tree_a, par_a, na, q_new_a, near_b, q_new_b);
                h = fnv_mix(h, (uint64_t)near_b);
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) h = fnv_mix(h, (uint64_t)__double_as_longlong(q_new_b[k]));
                h = fnv_mix(h, (uint64_t)rb);
                if (rb == 2) {  // Reached: rrt_connect.rs:281-305
                    const uint32_t idx_b = (grow_start ? nb : na) - 1;
                    st.goal_node = (int32_t)(grow_start ? idx_a : idx_b);
                    st.goal_node_b = (int32_t)(grow_start ? idx_b : idx_a);
                    done = true;
                }
            }
        }
        st.checksum = h;
        if (done) { stop = 0; break; }
    }
    if (tid == 0) {
        st.n_nodes = na;
        st.n_nodes_b = nb;
        st.draws = rng.pos;
        st.stop_reason = stop;
        p.state[prob] = st;
    }
}

void launch_rrt_connect(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kConnThreads);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL(rrt_connect_kernel<2>, grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL(rrt_connect_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(rrt_connect_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(rrt_connect_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(rrt_connect_kernel<6>, grid, block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL(rrt_connect_kernel<7>, grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL(rrt_connect_kernel<0>, grid, block, 0, stream, p); break;
    }
}

}  // namespace oxhip
