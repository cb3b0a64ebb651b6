// rrt_stream.hip -- general RRT grow kernel: one 256-thread workgroup per problem, the tree
// streamed from its SoA arrays in HBM / L2 every iteration.  Handles any dim <= 8 and any tree
// size; the register-resident kernel (rrt_resident.hip) is the fast path when the tree fits.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

constexpr int kStreamThreads = 256;
constexpr int kStreamWaves = kStreamThreads / 64;

// D = compile-time array extent; DIM = 0 selects the runtime-dim variant (D = 8)
template <int DIM>
__global__ __launch_bounds__(kStreamThreads) void rrt_stream_kernel(DevParams p_in) {
    constexpr int D = DIM ? DIM : kMaxDim;
    const int dim = DIM ? DIM : (int)p_in.dim;
    __shared__ ObsLds obs;
    const DevParams p = stage_obstacles(p_in, obs, threadIdx.x, kStreamThreads);   // visible after the first barrier
    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63;

    __shared__ uint32_t rng_buf[16][64];
    __shared__ Best wave_best[kStreamWaves];
    __shared__ Exact wave_exact[kStreamWaves];
    __shared__ uint32_t shadow_word;

    ProblemState st = p.state[prob];
    if (p.stop_at_goal && st.goal_node >= 0) return;  // already solved: solve() is idempotent

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * p.dim * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < dim) goal_c[k] = p.goal_c[(size_t)prob * p.dim + k];
    const double goal_thr = p.goal_thr[prob];

    RngWindow rng;
    rng.init(rng_buf, p.seed, p.first_problem_id + prob, st.draws);

    uint32_t n = st.n_nodes;
    // binary32 shadow of the tree (half the bytes per scan) and the error bounds of screening with it
    float* tree32 = p.tree32 + (size_t)prob * p.dim * cap;
    const ScreenMargins mg = screen_margins(shadow_sync<D>(p, dim, prob, tree, tree32, cap, n, goal_c, &shadow_word, tid, kStreamThreads), dim);

    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    for (uint64_t it = 0; it < p.budget; ++it) {
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }

        // 2. sample (rrt.rs:177-184)
        double q[D];
        sample_state<D>(rng, p, dim, goal_c, q, p.goal_r[prob]);

        // 3. nearest neighbour (rrt.rs:187-196).  First a binary32 SCREEN over the shadow (coalesced SoA scan, 4 bytes
        //    per coordinate): smallest and second smallest squared distance.  If the runner-up is provably farther than
        //    the winner (screen_clear), the winner is the reference's nearest node and its distance is recomputed in
        //    binary64 from the binary64 node; otherwise the binary64 scan below decides as before.
        Best best = best_init();
        bool screened = false;
        if (mg.usable) {
            float qf[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) qf[k] = (float)q[k];
            float s1 = __builtin_inff(), s2 = __builtin_inff();
            uint32_t si = 0xFFFFFFFFu;
            screen_scan<D>(tree32, cap, n, dim, qf, tid, kStreamThreads, [&](uint32_t i, float s) {
                s2 = __builtin_amdgcn_fmed3f(s, s1, s2);
                const bool lt = s < s1;
                s1 = lt ? s : s1;
                si = lt ? i : si;
            });
            best = best_wave_reduce(Best{(double)s1, (double)s2, si});
            if (lane == 0) wave_best[wave] = best;
            __syncthreads();
            best = wave_best[0];
#pragma unroll
            for (int w = 1; w < kStreamWaves; ++w) best = best_combine(best, wave_best[w]);
            screened = screen_clear(mg, best.b1, best.b2);
            if (!screened) __syncthreads();   // wave_best is reused by the binary64 scan
        }

        uint32_t nearest;
        double min_dist;
        if (screened) {
            nearest = best.i1;
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + nearest];
            min_dist = sqrt(dist2<D>(c, q, dim));
        } else {
        best = best_init();
        for (uint32_t i = tid; i < n; i += kStreamThreads) {
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + i];
            best_push(best, dist2<D>(c, q, dim), i);
        }
        best = best_wave_reduce(best);
        if (lane == 0) wave_best[wave] = best;
        __syncthreads();
        best = wave_best[0];
#pragma unroll
        for (int w = 1; w < kStreamWaves; ++w) best = best_combine(best, wave_best[w]);

        if (best_ambiguous(best)) {
            // rare: two d2 within 3 ulps -> exact post-sqrt compare with lowest-index ties
            Exact e{__builtin_inf(), 0xFFFFFFFFu};
            for (uint32_t i = tid; i < n; i += kStreamThreads) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) if (k < dim) c[k] = tree[(size_t)k * cap + i];
                double d = sqrt(dist2<D>(c, q, dim));
                if (d < e.dist) { e.dist = d; e.idx = i; }
            }
            e = exact_wave_reduce(e);
            if (lane == 0) wave_exact[wave] = e;
            __syncthreads();
            e = wave_exact[0];
#pragma unroll
            for (int w = 1; w < kStreamWaves; ++w) e = exact_combine(e, wave_exact[w]);
            nearest = e.idx;
            min_dist = e.dist;
        } else {
            nearest = best.i1;
            min_dist = sqrt(best.b1);
        }
        }
        nearest = uni(nearest);

        double q_near[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < dim) q_near[k] = tree[(size_t)k * cap + nearest];

        // 4. steer (rrt.rs:199-208)
        double q_new[D];
        if (min_dist > p.max_distance) {
            double t = p.max_distance / min_dist;
            lerp<D>(q_near, q, t, q_new, dim);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < dim) q_new[k] = q[k];
        }

        // 5. check_motion (rrt.rs:211 -> :90-116), (step, obstacle) pairs over the workgroup
        bool bad = motion_invalid_wg<D>(p, dim, q_near, q_new, tid, kStreamThreads);
        const bool ok = !__syncthreads_or(bad ? 1 : 0);

        st.checksum = chk_push(st.checksum, iter_digest<D>(nearest, q_new, dim, ok));
        st.iterations++;

        bool hit = false;
        if (ok) {
            st.accepted++;
            if (!p.freeze) {
                // 6. insert (rrt.rs:213-217)
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        if (k < dim) {
                            tree[(size_t)k * cap + n] = q_new[k];
                            tree32[(size_t)k * cap + n] = (float)q_new[k];
                        }
                    }
                    parent[n] = (int32_t)nearest;
                }
                ++n;
                // 7. goal test (rrt.rs:220-223)
                if (dist2<D>(q_new, goal_c, dim) <= goal_thr) {
                    if (st.goal_node < 0) st.goal_node = (int32_t)(n - 1);
                    hit = true;
                }
                __syncthreads();  // node n-1 visible to the whole workgroup before the next scan
            }
        }
        if (hit && p.stop_at_goal) { stop = 0; break; }
    }

    if (tid == 0) {
        st.n_nodes = n;
        st.draws = rng.pos;
        st.stop_reason = stop;
        p.state[prob] = st;
        p.shadow_state[2 * (size_t)prob] = n;                  // every node up to n has its shadow
        p.shadow_state[2 * (size_t)prob + 1] = shadow_word;    // (inserts stay inside the hull the bound M covers)
    }
}

void launch_rrt_stream(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kStreamThreads);
    switch (p.dim) {
        case 2: hipLaunchKernelGGL(rrt_stream_kernel<2>, grid, block, 0, stream, p); break;
        case 3: hipLaunchKernelGGL(rrt_stream_kernel<3>, grid, block, 0, stream, p); break;
        case 4: hipLaunchKernelGGL(rrt_stream_kernel<4>, grid, block, 0, stream, p); break;
        case 5: hipLaunchKernelGGL(rrt_stream_kernel<5>, grid, block, 0, stream, p); break;
        case 6: hipLaunchKernelGGL(rrt_stream_kernel<6>, grid, block, 0, stream, p); break;
        case 7: hipLaunchKernelGGL(rrt_stream_kernel<7>, grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL(rrt_stream_kernel<0>, grid, block, 0, stream, p); break;
    }
}

// ------------------------------------------------------------------ stand-alone primitives

// rrt.rs:187-196 for Q independent (tree, query) pairs; nodes AoS here (API layout)
__global__ __launch_bounds__(256) void nn_argmin_kernel(uint32_t dim, const double* nodes, const uint64_t* offsets,
                                                        const uint32_t* n_nodes, const double* queries,
                                                        uint32_t* out_index, double* out_min_dist) {
    constexpr int D = kMaxDim;
    const uint32_t qid = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ Best wave_best[4];
    __shared__ Exact wave_exact[4];
    const double* base = nodes + offsets[qid] * dim;
    const uint32_t n = n_nodes[qid];
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) if (k < (int)dim) q[k] = queries[(size_t)qid * dim + k];
    Best best = best_init();
    for (uint32_t i = tid; i < n; i += 256) {
        double c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) if (k < (int)dim) c[k] = base[(size_t)i * dim + k];
        best_push(best, dist2<D>(c, q, dim), i);
    }
    best = best_wave_reduce(best);
    if (lane == 0) wave_best[wave] = best;
    __syncthreads();
    best = wave_best[0];
    for (int w = 1; w < 4; ++w) best = best_combine(best, wave_best[w]);
    uint32_t nearest;
    double min_dist;
    if (best_ambiguous(best)) {
        Exact e{__builtin_inf(), 0xFFFFFFFFu};
        for (uint32_t i = tid; i < n; i += 256) {
            double c[D];
#pragma unroll
            for (int k = 0; k < D; ++k) if (k < (int)dim) c[k] = base[(size_t)i * dim + k];
            double d = sqrt(dist2<D>(c, q, dim));
            if (d < e.dist) { e.dist = d; e.idx = i; }
        }
        e = exact_wave_reduce(e);
        if (lane == 0) wave_exact[wave] = e;
        __syncthreads();
        e = wave_exact[0];
        for (int w = 1; w < 4; ++w) e = exact_combine(e, wave_exact[w]);
        nearest = e.idx;
        min_dist = e.dist;
    } else {
        nearest = best.i1;
        min_dist = sqrt(best.b1);
    }
    if (tid == 0) {
        out_index[qid] = nearest;
        out_min_dist[qid] = min_dist;
    }
}

void launch_nn_argmin(uint32_t dim, const double* nodes, const uint64_t* offsets, const uint32_t* n_nodes,
                      uint32_t n_queries, const double* queries, uint32_t* out_index, double* out_min_dist,
                      hipStream_t stream) {
    hipLaunchKernelGGL(nn_argmin_kernel, dim3(n_queries), dim3(256), 0, stream, dim, nodes, offsets, n_nodes,
                       queries, out_index, out_min_dist);
}

__global__ void distance_kernel(uint32_t dim, const double* a, const double* b, uint32_t n, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[kMaxDim], y[kMaxDim];
    for (int k = 0; k < kMaxDim; ++k) if (k < (int)dim) { x[k] = a[(size_t)i * dim + k]; y[k] = b[(size_t)i * dim + k]; }
    out[i] = sqrt(dist2<kMaxDim>(x, y, dim));
}

__global__ void interpolate_kernel(uint32_t dim, const double* from, const double* to, const double* t, uint32_t n,
                                   double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[kMaxDim], y[kMaxDim], o[kMaxDim];
    for (int k = 0; k < kMaxDim; ++k) if (k < (int)dim) { x[k] = from[(size_t)i * dim + k]; y[k] = to[(size_t)i * dim + k]; }
    lerp<kMaxDim>(x, y, t[i], o, dim);
    for (int k = 0; k < kMaxDim; ++k) if (k < (int)dim) out[(size_t)i * dim + k] = o[k];
}

// one wave per state / motion: lanes stripe the (step, obstacle) pairs, ballot reduces
__global__ __launch_bounds__(64) void is_valid_kernel(DevParams p, const double* states, uint32_t n, uint8_t* out) {
    uint32_t i = blockIdx.x;
    if (i >= n) return;
    double s[kMaxDim];
    for (int k = 0; k < kMaxDim; ++k) if (k < (int)p.dim) s[k] = states[(size_t)i * p.dim + k];
    bool bad = false;
    for (uint32_t j = threadIdx.x; j < p.n_spheres + p.n_boxes; j += 64) bad = bad || obstacle_hit<kMaxDim>(p, p.dim, s, j);
    bool any_bad = __any(bad ? 1 : 0);
    if (threadIdx.x == 0) out[i] = any_bad ? 0 : 1;
}

__global__ __launch_bounds__(64) void check_motion_kernel(DevParams p, const double* from, const double* to, uint32_t n,
                                                          uint8_t* out) {
    uint32_t i = blockIdx.x;
    if (i >= n) return;
    double a[kMaxDim], b[kMaxDim];
    for (int k = 0; k < kMaxDim; ++k) if (k < (int)p.dim) { a[k] = from[(size_t)i * p.dim + k]; b[k] = to[(size_t)i * p.dim + k]; }
    bool bad = motion_invalid_partial<kMaxDim>(p, p.dim, a, b, threadIdx.x, 64);
    bool any_bad = __any(bad ? 1 : 0);
    if (threadIdx.x == 0) out[i] = any_bad ? 0 : 1;
}

__global__ void f64_op_kernel(uint32_t op, const double* a, const double* b, const double* c, uint32_t n, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b ? b[i] : 0.0, z = c ? c[i] : 0.0, r;
    switch (op) {
        case 0: r = sqrt(x); break;
        case 1: r = x / y; break;
        case 2: r = ceil(x); break;
        case 3: { double diff = y - x; double sc = diff * z; r = x + sc; } break;
        case 5: { double sn, cs; ox_sincos(x, sn, cs); r = sn; } break;   // the disc goal sampler's sin / cos (ox_sincos.hpp)
        case 6: { double sn, cs; ox_sincos(x, sn, cs); r = cs; } break;
        default: { double d = x - y; r = d * d; } break;
    }
    out[i] = r;
}

__global__ void rng_u64_kernel(uint64_t seed, uint64_t stream, uint32_t n, uint64_t* out) {
    uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)blk * 8 >= n) return;
    uint32_t o[16];
    chacha12_block(seed, blk, stream, o);
    for (int w = 0; w < 8; ++w) {
        uint64_t idx = (uint64_t)blk * 8 + w;
        if (idx < n) out[idx] = ((uint64_t)o[2 * w + 1] << 32) | o[2 * w];
    }
}

void launch_distance(uint32_t dim, const double* a, const double* b, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(distance_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dim, a, b, n, out);
}
void launch_interpolate(uint32_t dim, const double* f, const double* t, const double* tt, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(interpolate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dim, f, t, tt, n, out);
}
void launch_is_valid(const DevParams& p, const double* states, uint32_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(is_valid_kernel, dim3(n), dim3(64), 0, s, p, states, n, out);
}
void launch_check_motion(const DevParams& p, const double* from, const double* to, uint32_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(check_motion_kernel, dim3(n), dim3(64), 0, s, p, from, to, n, out);
}
void launch_f64_op(uint32_t op, const double* a, const double* b, const double* c, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(f64_op_kernel, dim3((n + 255) / 256), dim3(256), 0, s, op, a, b, c, n, out);
}
void launch_rng_u64(uint64_t seed, uint64_t stream, uint32_t n, uint64_t* out, hipStream_t s) {
    uint32_t blocks = (n + 7) / 8;
    hipLaunchKernelGGL(rng_u64_kernel, dim3((blocks + 63) / 64), dim3(64), 0, s, seed, stream, n, out);
}

}  // namespace oxhip
