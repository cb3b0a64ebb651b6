// rrt_connect_se2.hip -- RRTConnect (oxmpl/src/geometric/planners/rrt_connect.rs:121-159,166-189,227-309) over
// SE(2) = R^2 x SO(2) with a segment-soup validity checker: BASELINE.json configs[3].
//
// The reference has no SE(2) space (docs/BACKLOG.md:12-14); it is assembled from the reference's components:
//   (x, y)  RealVectorStateSpace  distance rvss.rs:137-155, interpolate :161-186, sample :233-249
//   theta   SO2StateSpace         distance so2_state_space.rs:97-101, interpolate :107-122, sample :164-169,
//                                 normalise so2_state.rs:33-37, maximum extent PI :78-80
//   distance = 1.0 * d_xy + 0.5 * d_theta;  extent = extent_xy + 0.5 * PI        (OMPL's SE2 weights)
// The compound distance is not a Euclidean norm, so the d^2 shortcuts of the R^n kernels do not apply: every
// node's distance is evaluated exactly as the reference would (sqrt + fmod per node) and the argmin is the
// lexicographic (distance, index) minimum.  Validity: a disc of radius `clearance` among line segments, decided
// as d2(point, segment) > T(clearance) with the host-computed exact threshold (no sqrt, no transcendental).
// One 256-thread workgroup per problem, both trees as SoA [3][cap] arrays in HBM / L2.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

constexpr int kSe2Threads = 256;
constexpr int kSe2Waves = kSe2Threads / 64;
#define OXHIP_PI 3.14159265358979323846   // std::f64::consts::PI = 0x400921FB54442D18

// f64::rem_euclid(a, 2*PI): r = a % (2*PI); if r < 0.0 { r + 2*PI } else { r }.  fmod is exact, and for
// -2*PI < a < 4*PI it needs no division: a in [2*PI, 4*PI) gives a - 2*PI (exact by Sterbenz' lemma, which is
// fmod's value), a in [0, 2*PI) gives a, a in (-2*PI, 0) gives a, to which the reference then adds 2*PI -- the
// same rounded addition.  Everything a planner run produces lies in that window; the rest takes fmod.
__device__ __forceinline__ double rem_euclid_2pi(double a) {
    const double b = 2.0 * OXHIP_PI;
    double r;
    if (a >= 0.0 && a < b) r = a;
    else if (a >= b && a < 2.0 * b) r = a - b;
    else if (a < 0.0 && a > -b) r = a;
    else r = fmod(a, b);
    return r < 0.0 ? r + b : r;
}
__device__ __forceinline__ double so2_normalise(double v) { return rem_euclid_2pi(v + OXHIP_PI) - OXHIP_PI; }
__device__ __forceinline__ double so2_distance(double a, double b) {
    double diff = a - b;
    diff = rem_euclid_2pi(diff + OXHIP_PI) - OXHIP_PI;
    return fabs(diff);
}
__device__ __forceinline__ double so2_interpolate(double from, double to, double t) {
    double d = so2_normalise(to) - so2_normalise(from);
    if (d > OXHIP_PI) d -= 2.0 * OXHIP_PI;
    else if (d < -OXHIP_PI) d += 2.0 * OXHIP_PI;
    const double out = from + d * t;
    return so2_normalise(out);
}
__device__ __forceinline__ double se2_distance(const double a[3], const double b[3]) {
    const double dr = sqrt(dist2<2>(a, b, 2));
    const double ws = 0.5 * so2_distance(a[2], b[2]);
    return dr + ws;
}
__device__ __forceinline__ void se2_interpolate(const double from[3], const double to[3], double t, double out[3]) {
    lerp<2>(from, to, t, out, 2);
    out[2] = so2_interpolate(from[2], to[2], t);
}

// squared distance from (px, py) to segment j: project onto the segment (t clamped to [0,1], a degenerate
// segment or NaN gives t = 0), every operation rounded separately in exactly this order (it defines the checker)
__device__ __forceinline__ double point_segment_d2(const DevParams& p, double px, double py, uint32_t j) {
    const double ax = p.segs[4 * (size_t)j], ay = p.segs[4 * (size_t)j + 1];
    const double bx = p.segs[4 * (size_t)j + 2], by = p.segs[4 * (size_t)j + 3];
    const double abx = bx - ax, aby = by - ay;
    const double apx = px - ax, apy = py - ay;
    const double l1 = abx * abx, l2 = aby * aby;
    const double len2 = l1 + l2;
    double t = 0.0;
    if (len2 > 0.0) {
        const double n1 = apx * abx, n2 = apy * aby;
        const double num = n1 + n2;
        t = num / len2;
    }
    if (!(t > 0.0)) t = 0.0;
    if (t > 1.0) t = 1.0;
    const double sx = abx * t, sy = aby * t;
    const double cx = ax + sx, cy = ay + sy;
    const double dx = px - cx, dy = py - cy;
    const double q1 = dx * dx, q2 = dy * dy;
    return q1 + q2;
}
// is the state invalid because of segment j?  valid iff sqrt(d2) > clearance  <=>  d2 > seg_thr
__device__ __forceinline__ bool segment_hit(const DevParams& p, const double s[3], uint32_t j) {
    return !(point_segment_d2(p, s[0], s[1], j) > p.seg_thr);
}

// rrt_connect.rs:166-189 for `nthreads` callers (a multiple of 64); returns this caller's flag.  The steps are
// dealt to the waves and the segments to the lanes: the interpolated state (one normalisation chain) is computed
// once per step and is wave-uniform, no index division is needed.
__device__ __forceinline__ bool se2_motion_invalid_partial(const DevParams& p, const double from[3], const double to[3],
                                                           uint32_t tid, uint32_t nthreads) {
    if (p.n_segs == 0) return false;
    const double dist = se2_distance(from, to);
    const uint32_t nsteps = num_steps_u32(dist, p.res);
    const uint32_t wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
    bool bad = false;
    if (nsteps <= 1) {
        for (uint32_t j = tid; j < p.n_segs; j += nthreads) bad = bad || segment_hit(p, to, j);
        return bad;
    }
    const double dn = (double)nsteps;
    for (uint32_t step = wave + 1; step <= nsteps && step > wave; step += nwaves) {   // `step > wave`: no wrap at 2^32
        double s[3];
        se2_interpolate(from, to, (double)step / dn, s);
        for (uint32_t j = lane; j < p.n_segs; j += 64) bad = bad || segment_hit(p, s, j);
    }
    return bad;
}

constexpr int kSe2LdsSegs = 512;   // segments staged in LDS (16 KB); larger soups are read from HBM / L2

struct Se2Shared {
    uint32_t rng_buf[16][64];
    Exact wave_exact[kSe2Waves];
    double segs[kSe2LdsSegs][4];
};

// extend() of rrt_connect.rs:121-159; 0 = motion invalid, 1 = Advanced, 2 = Reached
__device__ __forceinline__ int se2_extend(const DevParams& p, Se2Shared& sh, double* tree, int32_t* parent, size_t cap,
                                          uint32_t& n, const double q[3], uint32_t& nearest, double q_new[3]) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    Exact e{__builtin_inf(), 0xFFFFFFFFu};
    for (uint32_t i = tid; i < n; i += kSe2Threads) {   // indices ascend within a thread: strict < keeps the lowest
        const double c[3] = {tree[i], tree[cap + i], tree[2 * cap + i]};
        const double d = se2_distance(c, q);
        if (d < e.dist) { e.dist = d; e.idx = i; }
    }
    e = exact_wave_reduce(e);
    if (lane == 0) sh.wave_exact[wave] = e;
    __syncthreads();
    e = sh.wave_exact[0];
#pragma unroll
    for (int w = 1; w < kSe2Waves; ++w) e = exact_combine(e, sh.wave_exact[w]);
    nearest = uni(e.idx);
    const double min_dist = unid(e.dist);
    const double q_near[3] = {tree[nearest], tree[cap + nearest], tree[2 * cap + nearest]};
    int result;
    if (min_dist > p.max_distance) {   // rrt_connect.rs:140-147
        se2_interpolate(q_near, q, p.max_distance / min_dist, q_new);
        result = 1;
    } else {
        q_new[0] = q[0]; q_new[1] = q[1]; q_new[2] = q[2];
        result = 2;
    }
    const bool bad = se2_motion_invalid_partial(p, q_near, q_new, tid, kSe2Threads);
    if (__syncthreads_or(bad ? 1 : 0)) return 0;
    if (tid == 0) {
        tree[n] = q_new[0]; tree[cap + n] = q_new[1]; tree[2 * cap + n] = q_new[2];
        parent[n] = (int32_t)nearest;
    }
    ++n;
    __syncthreads();
    return result;
}

__global__ __launch_bounds__(kSe2Threads) void rrt_connect_se2_kernel(DevParams p) {
    const uint32_t prob = blockIdx.x, tid = threadIdx.x;
    __shared__ Se2Shared sh;
    if (p.n_segs <= (uint32_t)kSe2LdsSegs) {   // the checker's table at LDS latency (made visible by the first barrier)
        for (uint32_t i = tid; i < 4 * p.n_segs; i += kSe2Threads) (&sh.segs[0][0])[i] = p.segs[i];
        p.segs = &sh.segs[0][0];
    }
    ProblemState st = p.state[prob];
    if (st.goal_node >= 0) return;
    const size_t cap = p.cap;
    double* tree_a = p.tree + (size_t)prob * 3 * cap;
    double* tree_b = p.tree_b + (size_t)prob * 3 * cap;
    int32_t* par_a = p.parent + (size_t)prob * cap;
    int32_t* par_b = p.parent_b + (size_t)prob * cap;
    const double goal_c[3] = {p.goal_c[(size_t)prob * 3], p.goal_c[(size_t)prob * 3 + 1], p.goal_c[(size_t)prob * 3 + 2]};
    const double goal_radius = p.goal_thr[prob];   // the radius itself: the goal test compares the compound distance

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint32_t na = st.n_nodes, nb = st.n_nodes_b;
    int32_t stop = 1;
    for (uint64_t it = 0; it < p.budget; ++it) {
        if (na >= p.max_nodes || nb >= p.max_nodes) { stop = 2; break; }
        const bool grow_start = na <= nb;   // rrt_connect.rs:249-254
        // sample: random_bool, then x, y, theta by random_range (lo/hi/scale[2] hold the clamped SO(2) bounds)
        double q_rand[3];
        sample_state<3>(rng, p, 3, goal_c, q_rand);
        uint32_t near_a = 0, near_b = 0;
        double qa[3], qb[3];
        const int ra = grow_start ? se2_extend(p, sh, tree_a, par_a, cap, na, q_rand, near_a, qa)
                                  : se2_extend(p, sh, tree_b, par_b, cap, nb, q_rand, near_a, qa);
        uint64_t h = fnv_mix(st.checksum, grow_start ? 1ull : 0ull);
        h = fnv_mix(h, (uint64_t)near_a);
#pragma unroll
        for (int k = 0; k < 3; ++k) h = fnv_mix(h, (uint64_t)__double_as_longlong(qa[k]));
        h = fnv_mix(h, (uint64_t)ra);
        st.iterations++;
        bool done = false;
        if (ra) {
            const uint32_t idx_a = (grow_start ? na : nb) - 1;
            if (grow_start && se2_distance(qa, goal_c) <= goal_radius) {   // rrt_connect.rs:271-274
                st.goal_node = (int32_t)idx_a;
                st.goal_node_b = -1;
                done = true;
            } else {
                const int rb = grow_start ? se2_extend(p, sh, tree_b, par_b, cap, nb, qa, near_b, qb)
                                          : se2_extend(p, sh, tree_a, par_a, cap, na, qa, near_b, qb);
                h = fnv_mix(h, (uint64_t)near_b);
#pragma unroll
                for (int k = 0; k < 3; ++k) h = fnv_mix(h, (uint64_t)__double_as_longlong(qb[k]));
                h = fnv_mix(h, (uint64_t)rb);
                if (rb == 2) {
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

void launch_rrt_connect_se2(const DevParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(rrt_connect_se2_kernel, dim3(p.n_problems), dim3(kSe2Threads), 0, stream, p);
}

// ---- stand-alone primitives (parity tests of the SO(2) / SE(2) arithmetic and of the checker)
__global__ void se2_op_kernel(uint32_t op, const double* a, const double* b, const double* t, uint32_t n, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x[3] = {a[3 * (size_t)i], a[3 * (size_t)i + 1], a[3 * (size_t)i + 2]};
    const double y[3] = {b[3 * (size_t)i], b[3 * (size_t)i + 1], b[3 * (size_t)i + 2]};
    double o[3];
    if (op == 0) {
        o[0] = se2_distance(x, y);
        o[1] = so2_normalise(x[2]);
        o[2] = so2_distance(x[2], y[2]);
    } else {
        se2_interpolate(x, y, t[i], o);
    }
    out[3 * (size_t)i] = o[0]; out[3 * (size_t)i + 1] = o[1]; out[3 * (size_t)i + 2] = o[2];
}
void launch_se2_op(uint32_t op, const double* a, const double* b, const double* t, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(se2_op_kernel, dim3((n + 255) / 256), dim3(256), 0, s, op, a, b, t, n, out);
}

__global__ void se2_is_valid_kernel(DevParams p, const double* states, uint32_t n, uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s[3] = {states[3 * (size_t)i], states[3 * (size_t)i + 1], states[3 * (size_t)i + 2]};
    bool bad = false;
    for (uint32_t j = 0; j < p.n_segs; ++j) bad = bad || segment_hit(p, s, j);
    out[i] = bad ? 0 : 1;
}
void launch_se2_is_valid(const DevParams& p, const double* states, uint32_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(se2_is_valid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, states, n, out);
}

// one wave per motion
__global__ __launch_bounds__(256) void se2_check_motion_kernel(DevParams p, const double* from, const double* to, uint32_t n,
                                                                uint8_t* out) {
    const uint32_t m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= n) return;
    const double f[3] = {from[3 * (size_t)m], from[3 * (size_t)m + 1], from[3 * (size_t)m + 2]};
    const double g[3] = {to[3 * (size_t)m], to[3 * (size_t)m + 1], to[3 * (size_t)m + 2]};
    const bool bad = se2_motion_invalid_partial(p, f, g, lane, 64);
    const bool any = __ballot(bad) != 0;
    if (lane == 0) out[m] = any ? 0 : 1;
}
void launch_se2_check_motion(const DevParams& p, const double* from, const double* to, uint32_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(se2_check_motion_kernel, dim3((n + 3) / 4), dim3(256), 0, s, p, from, to, n, out);
}

}  // namespace oxhip
