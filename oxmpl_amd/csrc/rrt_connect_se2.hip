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
// ONE WAVE PER PROBLEM (a 64-thread workgroup, four per CU): an iteration is a chain of dependent steps -- sample, scan, steer,
// motion check, insert, twice -- so what counts is the length of that chain, not throughput.  Both trees are mirrored in LDS
// (the SoA [3][cap] arrays in HBM receive every store), nothing waits at a barrier, and the motion check looks its segments up
// in a grid instead of sweeping the soup (seg_grid_kernel below).
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

#define OXHIP_PI 3.14159265358979323846   // std::f64::consts::PI = 0x400921FB54442D18

// f64::rem_euclid(a, 2*PI): r = a % (2*PI); if r < 0.0 { r + 2*PI } else { r }.  fmod is exact, and for
// -2*PI < a < 4*PI it needs no division: a in [2*PI, 4*PI) gives a - 2*PI (exact by Sterbenz' lemma, which is
// fmod's value), a in [0, 2*PI) gives a, a in (-2*PI, 0) gives a, to which the reference then adds 2*PI -- the
// same rounded addition.  Everything a planner run produces lies in that window; the rest takes fmod.
__device__ __forceinline__ double rem_euclid_2pi(double a) {
    const double b = 2.0 * OXHIP_PI;
    // the window, by selects (one uniform branch instead of a chain of divergent ones: this runs several times per extend)
    double r = a;                                  // a in [0, b), a in (-b, 0), -0.0: fmod's value is a itself
    r = (a >= b) ? a - b : r;                      // a in [b, 2 b): fmod's value, exactly (Sterbenz)
    if (__builtin_expect(__ballot(!(a > -b && a < 2.0 * b)) != 0, 0)) {   // outside the window (or NaN): fmod proper
        if (!(a > -b && a < 2.0 * b)) r = fmod(a, b);
    }
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
__device__ __forceinline__ double point_segment_d2(const double* segs, double px, double py, uint32_t j) {
    const double ax = segs[4 * (size_t)j], ay = segs[4 * (size_t)j + 1];
    const double bx = segs[4 * (size_t)j + 2], by = segs[4 * (size_t)j + 3];
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
    return !(point_segment_d2(p.segs, s[0], s[1], j) > p.seg_thr);
}

// ---- the checker's segments, looked up instead of swept (DevParams::seg_grid)
//
// A motion check tests 7 states against 256 segments in configs[3], and all but a handful of those 1,792 tests are against segments
// far away.  seg_grid_kernel (run when the segments change) files the soup in a kSegGridG x kSegGridG grid over the (x, y) bounds:
// a cell lists the (up to kSegSlots) segments that can come within `clearance` of ANY point of the cell, ascending, 0xFFFF = no more;
// a cell that more segments reach says kSegOverflow in its first slot, and its states -- like states outside the bounds -- are
// tested against every segment.  "Can come within": d(centre, segment) <= clearance + h, h = the cell's half diagonal taken 2^-8
// wider (the triangle inequality; the widening and the relative 1e-9 are orders of magnitude above the roundings of
// point_segment_d2 and of the cell index).  A segment a cell does not list is farther than the clearance from every state in the
// cell, its test would come out "no hit", and is_valid is the conjunction of those tests: same verdicts, by construction.
constexpr uint32_t kSegGridG = 256;
constexpr int kSegSlots = 8;
constexpr uint32_t kSegNone = 0xFFFFu, kSegOverflow = 0xFFFEu;

__global__ __launch_bounds__(256) void seg_grid_kernel(DevParams p, uint16_t* grid, double clearance) {
    const uint32_t G = p.seg_grid_G, idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= G * G) return;
    const uint32_t ix = idx % G, iy = idx / G;
    const double wx = (p.hi[0] - p.lo[0]) / (double)G, wy = (p.hi[1] - p.lo[1]) / (double)G;
    const double cx = p.lo[0] + ((double)ix + 0.5) * wx, cy = p.lo[1] + ((double)iy + 0.5) * wy;
    const double h = 0.5 * sqrt(wx * wx + wy * wy) * (1.0 + 0x1p-8);
    const double R = (clearance > 0.0 ? clearance : 0.0) + h, R2 = R * R * (1.0 + 1e-9);
    uint32_t cnt = 0;
    uint16_t ids[kSegSlots];
#pragma unroll
    for (int k = 0; k < kSegSlots; ++k) ids[k] = (uint16_t)kSegNone;
    const uint32_t ns = p.n_segs < kSegOverflow ? p.n_segs : kSegOverflow;   // (ids beyond 16 bits: the cell overflows below)
    for (uint32_t j = 0; j < ns; ++j) {
        if (point_segment_d2(p.segs, cx, cy, j) > R2) continue;   // (NaN: listed)
        if (cnt < (uint32_t)kSegSlots) {
#pragma unroll
            for (int k = 0; k < kSegSlots; ++k) if ((uint32_t)k == cnt) ids[k] = (uint16_t)j;
        }
        ++cnt;
    }
    if (cnt > (uint32_t)kSegSlots || p.n_segs > ns) ids[0] = (uint16_t)kSegOverflow;
    uint4 e;
    e.x = ids[0] | ((uint32_t)ids[1] << 16); e.y = ids[2] | ((uint32_t)ids[3] << 16);
    e.z = ids[4] | ((uint32_t)ids[5] << 16); e.w = ids[6] | ((uint32_t)ids[7] << 16);
    reinterpret_cast<uint4*>(grid)[idx] = e;
}
uint32_t seg_grid_side() { return kSegGridG; }
void launch_seg_grid(const DevParams& p, uint16_t* grid, double clearance, hipStream_t stream) {
    const uint32_t cells = p.seg_grid_G * p.seg_grid_G;
    hipLaunchKernelGGL(seg_grid_kernel, dim3((cells + 255u) / 256u), dim3(256), 0, stream, p, grid, clearance);
}

// rrt_connect.rs:166-189 for ONE WAVE; returns the wave-uniform verdict "some tested state is invalid".  Only (x, y) enter the
// checker, so only they are interpolated (is_valid is pure: the heading's interpolation has no effect the reference could observe).
// Up to eight states: eight lanes per state, lane k of a state takes slot k of its cell's list -- the whole check is one
// point_segment_d2 deep.  More states: a lane per state, which walks its cell's list.
// tdiv (optional, LDS): tdiv[n - 1][s] = (s + 1) / n for n, s + 1 <= 8 -- the quotients themselves, computed once per launch.
__device__ __forceinline__ bool se2_motion_invalid_wave(const DevParams& p, const double* segs, const double from[3], const double to[3],
                                                        uint32_t nsteps, uint32_t lane, const double (*tdiv)[8] = nullptr) {
    if (p.n_segs == 0) return false;
    const uint32_t S = nsteps <= 1 ? 1u : nsteps;               // states tested: `to` alone, or steps 1 ..= nsteps
    const uint32_t g = S <= 8u ? 8u : 1u, per_pass = 64u / g;
    const uint32_t k0 = lane & (g - 1u), sub = lane / g;
    const double dn = (double)nsteps;
    const uint4* grid = reinterpret_cast<const uint4*>(p.seg_grid);
    for (uint32_t s0 = 0; s0 < S && s0 + per_pass > s0; s0 += per_pass) {
        const uint32_t s = s0 + sub;
        bool bad = false;
        if (s < S) {
            double x = to[0], y = to[1];
            if (nsteps > 1) {
                const double t = (tdiv && nsteps <= 8u) ? tdiv[nsteps - 1u][s] : (double)(s + 1u) / dn;
                double xy[2];
                lerp<2>(from, to, t, xy, 2);
                x = xy[0]; y = xy[1];
            }
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            bool all = true;
            if (grid) {
                const double fx = (x - p.lo[0]) * p.seg_grid_inv[0], fy = (y - p.lo[1]) * p.seg_grid_inv[1];
                const double G = (double)p.seg_grid_G;
                if (fx >= 0.0 && fx < G && fy >= 0.0 && fy < G) {   // (NaN: every segment)
                    e = grid[(uint32_t)fy * p.seg_grid_G + (uint32_t)fx];
                    all = (e.x & 0xFFFFu) == kSegOverflow;
                }
            }
            if (all) {
                for (uint32_t j = k0; j < p.n_segs; j += g) bad = bad || !(point_segment_d2(segs, x, y, j) > p.seg_thr);
            } else {
                const uint32_t w[4] = {e.x, e.y, e.z, e.w};
                for (uint32_t k = k0; k < (uint32_t)kSegSlots; k += g) {
                    uint32_t pair = w[0];
#pragma unroll
                    for (int q = 1; q < 4; ++q) if ((k >> 1) == (uint32_t)q) pair = w[q];
                    const uint32_t id = (k & 1u) ? pair >> 16 : pair & 0xFFFFu;
                    if (id == kSegNone) break;
                    bad = bad || !(point_segment_d2(segs, x, y, id) > p.seg_thr);
                }
            }
        }
        if (__ballot(bad) != 0) return true;
    }
    return false;
}

constexpr int kSe2N = 768;          // nodes of each tree shadowed in LDS (binary32); beyond that every node is evaluated exactly
constexpr int kSe2NSmall = 512;     // ... in the launches of more problems than the chip has SIMDs (below)
constexpr int kSe2LdsSegs = 256;    // segments staged in LDS; larger soups are read from HBM / L2

// LDS of one problem = one wave = one workgroup.  NS = 768, SEGS = 256: 39 KB, four problems per CU = one per SIMD -- the shape for
// batches the chip holds at once, where a problem's latency is what counts.  NS = 512, SEGS = 1 (segments from HBM / L2): 22.5 KB,
// seven per CU -- for larger batches, where the waves of other problems fill the gaps of a latency-bound one (VALU busy is 0.10 at
// one wave per SIMD).  Same results: the shadow's size only decides how often the exact path runs.
template <int NS, int SEGS>
struct Se2Shared {
    static constexpr int kN = NS;
    uint32_t rng_buf[16][64];
    float4 shadow_a[NS];            // fl32(x, y, theta) of the start tree's nodes
    float4 shadow_b[NS];            // ... of the goal tree's
    double segs[SEGS][4];
    double q[3][64];                // the samples of the current block of 64 iterations ...
    uint64_t pos_after[64];         // ... and the stream position after each of them
    double tdiv[8][8];              // (s + 1) / n, n = 1 .. 8 (se2_motion_invalid_wave)
};
static_assert(sizeof(Se2Shared<kSe2N, kSe2LdsSegs>) <= 40960, "four problems per CU");
static_assert(sizeof(Se2Shared<kSe2NSmall, 1>) <= 23405, "seven problems per CU");
static_assert(kSe2N % 32 == 0 && kSe2NSmall % 32 == 0, "se2_round reads the shadow four slots of eight lanes at a time");

// Lane-parallel sampling of m <= 64 consecutive iterations (rrt_connect.rs:258-262 + the SE(2) sample_uniform; the scheme of
// rrt_cells.hip's cells_sample): lane j draws iteration j.  Where its words start depends on how many of the iterations before
// it sampled the goal (one word instead of four), so the goal mask is iterated to its fixed point: in round r the first r lanes
// are right.  Returns false, nothing written, when a range draw was rejected or the window is too short.
template <class SH>
__device__ __forceinline__ bool se2_sample64(RngWindow& rng, const DevParams& p, const double* goal_c, uint32_t m, uint32_t lane, SH& sh) {
    const uint64_t win_lo = rng.base_blk * 8;
    const uint64_t pos0 = rng.pos;
    if (pos0 < win_lo || pos0 + (uint64_t)m * 4u > win_lo + 512) return false;
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
            off = act ? 4u * lane - 3u * (uint32_t)__popcll(goal_mask & below) : 0u;
            const uint64_t now = __ballot(act && word(off) < p.p_int);
            if (now == goal_mask) break;
            goal_mask = now;
        }
    }
    const bool goal = (goal_mask >> lane) & 1ull;
    double q[3];
    bool redraw = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint64_t bits = (word(act && !goal ? off + 1u + (uint32_t)k : 0u) >> 12) | 0x3FF0000000000000ull;
        const double v01 = __longlong_as_double((long long)bits) - 1.0;
        double res = v01 * p.scale[k];
        res = res + p.lo[k];
        redraw = redraw || !(res < p.hi[k]);
        q[k] = goal ? goal_c[k] : res;
    }
    if (__ballot(act && redraw && !goal) != 0) return false;
    const uint32_t cnt = always_goal ? 0u : (goal ? 1u : 4u);
    if (act) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sh.q[k][lane] = q[k];
        sh.pos_after[lane] = pos0 + off + cnt;
    }
    rng.pos = pos0 + (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), (int)(m - 1));
    return true;
}
template <class SH>
__device__ __forceinline__ void se2_sample_block(RngWindow& rng, const DevParams& p, const double* goal_c, uint32_t m, uint32_t lane, SH& sh) {
    const uint64_t need_hi = rng.pos + (uint64_t)m * 4u;
    if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
        rng.base_blk = uni64(rng.pos >> 3);
        uint32_t o[16];
        chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
        for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
    }
    if (!se2_sample64(rng, p, goal_c, m, lane, sh)) {
        for (uint32_t b = 0; b < m; ++b) {   // (never expected) a redraw: one by one
            double qn[3];
            sample_state<3, false>(rng, p, 3, goal_c, qn);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) sh.q[k][b] = qn[k];
                sh.pos_after[b] = rng.pos;
            }
        }
    }
}

// One tree: the binary64 SoA arrays in HBM (every store lands there; the scan reads only its candidates back) and the binary32
// shadow in LDS the scan screens with.
struct Se2Tree {
    double* g;        // SoA [3][cap] in HBM
    float4* sh;       // LDS [N]
    size_t cap;
    uint32_t N;       // nodes the shadow holds
    __device__ __forceinline__ void load(uint32_t i, double c[3]) const { c[0] = g[i]; c[1] = g[cap + i]; c[2] = g[2 * cap + i]; }
};

// what the screen's error bound depends on, wave-uniform, kept current by every insert
struct Se2Range {
    float mag;        // largest |x|, |y| of any node of either tree
    bool theta_ok;    // every heading lies in [-PI, PI] (then the heading distance is min(|d|, 2 PI - |d|))
};
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
constexpr float kSe2PiUp = 3.14159274f;   // fl32(PI), which is > PI
__device__ __forceinline__ float se2_screen(const float4 s, float qx, float qy, float qt) {
    const float dx = s.x - qx, dy = s.y - qy;
    const float r = __builtin_amdgcn_sqrtf(__builtin_fmaf(dy, dy, dx * dx));
    const float a = fabsf(s.z - qt);
    return __builtin_fmaf(0.5f, fminf(a, 6.28318548f - a), r);   // (fused or not: the estimate's error bound covers either)
}
// smallest and second smallest estimate of a lane and the node with the smallest (the first one: a repeated value shows up as b2 == b1)
__device__ __forceinline__ void se2_top2(float& b1, float& b2, uint32_t& i1, float d, uint32_t i) {
    i1 = d < b1 ? i : i1;
    float m;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(d), "v"(b1), "v"(b2));   // the second smallest of three with b1 <= b2
    b2 = m;
    b1 = fminf(b1, d);
}


// Nearest node of rrt_connect.rs:128-136 under the compound distance: the lexicographic (distance, index) minimum over the tree, the
// distance evaluated exactly as the reference does (sqrt + rem_euclid per node).  Evaluating that for every node is what an
// iteration's time went into; here every node gets a binary32 estimate d^ first (|d^ - d| <= E, derivation in DESIGN.md section 11:
// E = 32 * 2^-24 * (mag + PI) covers the roundings of the stored coordinates, of the differences, the 1-ulp square root, the fl32
// 2 PI and the final sum about twice over), and only nodes with d^ <= min d^ + 2 E -- the true winner is always one of them -- are
// evaluated exactly and compared.  Almost always that is one node.
__device__ __forceinline__ void se2_nearest(const Se2Tree& tree, uint32_t n, const double q[3], const Se2Range& rg, uint32_t lane,
                                            uint32_t& nearest, double& min_dist, double q_near[3]) {
    const float qx = (float)q[0], qy = (float)q[1], qt = (float)q[2];
    const uint32_t ns = n < tree.N ? n : tree.N;
    const bool screen = rg.theta_ok && fabsf(qt) <= kSe2PiUp;
    float thr = __builtin_inff();
    bool slow = !screen || n > ns;
    float b1 = __builtin_inff(), b2 = __builtin_inff();
    uint32_t i1 = 0xFFFFFFFFu;
    // (the loads below read what this wave stored earlier: drain its stores first -- they were issued an extend ago, no wait in practice)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    Exact e{__builtin_inf(), 0xFFFFFFFFu};
    double c[3] = {0.0, 0.0, 0.0};
    if (screen) {
        const uint32_t trips = (ns + 63u) >> 6;   // uniform: four independent LDS reads per turn
        for (uint32_t t = 0; t < trips; t += 4u) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t i = lane + ((t + (uint32_t)u) << 6); v[u] = tree.sh[i < tree.N ? i : 0u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // (slots past the tree hold kSe2Far: estimate +inf, never a candidate)
                const uint32_t i = lane + ((t + (uint32_t)u) << 6);
                se2_top2(b1, b2, i1, i < tree.N ? se2_screen(v[u], qx, qy, qt) : __builtin_inff(), i);
            }
        }
        if (!slow && i1 != 0xFFFFFFFFu) tree.load(i1, c);   // in flight across the reduction: almost always it is the lane's only candidate
        const float m = __uint_as_float(wave_min_u32(__float_as_uint(b1)));   // (estimates are >= +0: their bit patterns order like they do)
        const float mag = fmaxf(rg.mag, fmaxf(fabsf(qx), fabsf(qy)));
        thr = m + (mag + 3.2f) * (2.02f * 32.0f * 0x1p-24f);
        slow = slow || __ballot(b2 <= thr) != 0;
    }
    if (!slow) {   // every lane holds at most one candidate: its best
        const bool cand = b1 <= thr;
        if (cand) {
            e.dist = se2_distance(c, q);
            e.idx = i1;
        }
        const uint64_t cm = __ballot(cand);
        if ((cm & (cm - 1ull)) != 0) {   // several lanes: the reference's comparison among them
            const Exact w = exact_wave_reduce(e);
            const uint64_t wm = __ballot(cand && e.idx == w.idx);
            const int L = __builtin_ctzll(wm);
            nearest = w.idx;
            min_dist = w.dist;
#pragma unroll
            for (int k = 0; k < 3; ++k) q_near[k] = readlane_f64(c[k], L);
            return;
        }
        const int L = __builtin_ctzll(cm);
        nearest = (uint32_t)__builtin_amdgcn_readlane((int)e.idx, L);
        min_dist = readlane_f64(e.dist, L);
#pragma unroll
        for (int k = 0; k < 3; ++k) q_near[k] = readlane_f64(c[k], L);
        return;
    }
    for (uint32_t i = lane; i < n; i += 64u) {   // indices ascend within a lane: strict < keeps the lowest
        if (screen && i < ns && !(se2_screen(tree.sh[i], qx, qy, qt) <= thr)) continue;
        tree.load(i, c);
        const double d = se2_distance(c, q);
        if (d < e.dist) { e.dist = d; e.idx = i; }
    }
    e = exact_wave_reduce(e);
    nearest = e.idx;
    min_dist = e.dist;
    tree.load(nearest, q_near);
}

// cycle stamps of problem 0 (diagnostic instantiation, oxhip_rrt_batch_enable_stamps): DevParams::dbg[0..11] = cycles spent sampling,
// in the rounds' nearest-neighbour searches, their steers, their motion checks, in the whole-wave extends, committing (checksums, inserts,
// goal tests; includes the whole-wave extends), in the whole loop; iterations; whole-wave extends; rounds; iterations a round committed as failures
template <bool STAMP>
__device__ __forceinline__ uint64_t se2_clock() { return STAMP ? (uint64_t)__builtin_readcyclecounter() : 0ull; }

// steer of rrt_connect.rs:140-147 and the step count of the motion check that follows; 1 = Advanced, 2 = Reached
__device__ __forceinline__ int se2_steer(const DevParams& p, const Se2Range& rg, const double q_near[3], const double q[3], double min_dist,
                                         double q_new[3], uint32_t& nsteps) {
    if (min_dist > p.max_distance) {
        se2_interpolate(q_near, q, p.max_distance / min_dist, q_new);
        // check_motion's step count is ceil(distance(q_near, q_new) / res), and q_new lies max_distance along the geodesic from q_near:
        // the computed distance is max_distance up to a few roundings of quantities no larger than mag + PI (< 2^-45 (mag + 4 + max_distance)
        // by a wide margin).  When max_distance / res is farther than that from an integer (adv_slack, in distance units, from the host)
        // the count is the constant adv_steps, and the square root, the division and the ceil it would take are not evaluated.
        const bool known = p.adv_steps != 0u && rg.theta_ok && fabsf((float)q[2]) <= kSe2PiUp &&
                           ((double)(rg.mag + fabsf((float)q[0]) + fabsf((float)q[1]) + 4.0f) + p.max_distance) * 0x1p-45 < p.adv_slack;   // (the distance itself is one of those quantities)
        nsteps = known ? p.adv_steps : num_steps_u32(se2_distance(q_near, q_new), p.res);
        return 1;
    }
    q_new[0] = q[0]; q_new[1] = q[1]; q_new[2] = q[2];
    nsteps = num_steps_u32(min_dist, p.res);   // distance(q_near, q): the value the scan computed for this very pair
    return 2;
}

// tree.push of rrt_connect.rs:150-157
__device__ __forceinline__ void se2_insert(const Se2Tree& tree, int32_t* parent, uint32_t& n, Se2Range& rg, uint32_t nearest, const double q_new[3],
                                           uint32_t lane) {
    if (lane == 0) {
        tree.g[n] = q_new[0]; tree.g[tree.cap + n] = q_new[1]; tree.g[2 * tree.cap + n] = q_new[2];
        parent[n] = (int32_t)nearest;
        if (n < tree.N) tree.sh[n] = make_float4((float)q_new[0], (float)q_new[1], (float)q_new[2], 0.0f);
    }
    rg.mag = fmaxf(rg.mag, fmaxf(fabsf((float)q_new[0]), fabsf((float)q_new[1])));
    rg.theta_ok = rg.theta_ok && fabsf((float)q_new[2]) <= kSe2PiUp;
    // the next scan is this wave's own and LDS is in order per wave (the fence keeps the compiler from moving the store)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    ++n;
}

// extend() of rrt_connect.rs:121-159 by the whole wave, up to the verdict (no insert); 0 = motion invalid, 1 = Advanced, 2 = Reached
template <bool STAMP>
__device__ __forceinline__ int se2_extend_try(const DevParams& p, const double* segs, const Se2Tree& tree, uint32_t n, const Se2Range& rg,
                                              const double q[3], uint32_t& nearest, double q_new[3], const double (*tdiv)[8], uint64_t* acc) {
    const uint32_t lane = threadIdx.x & 63u;
    double min_dist, q_near[3];
    const uint64_t t0 = se2_clock<STAMP>();
    se2_nearest(tree, n, q, rg, lane, nearest, min_dist, q_near);
    uint32_t nsteps;
    const int result = se2_steer(p, rg, q_near, q, min_dist, q_new, nsteps);
    const bool invalid = se2_motion_invalid_wave(p, segs, q_near, q_new, nsteps, lane, tdiv);
    if (STAMP) { acc[4] += se2_clock<STAMP>() - t0; acc[8] += 1; }
    return invalid ? 0 : result;
}

// ---- eight iterations side by side
//
// Four of five iterations end with "motion invalid" (configs[3]), and an iteration that fails changes nothing: not the trees, not which
// tree grows next.  So the next R <= 8 iterations' first extends are evaluated together against the trees as they are -- eight lanes
// per iteration: they share the scan of the tree (node i to lane i mod 8), all evaluate the steer, and lane s takes state s of the
// motion check, walking its cell's segment list -- and committed in order up to and including the first one that succeeds; what was
// computed for the iterations behind a success is dropped (the tree has changed).  Arithmetic and decisions are se2_extend_try's.
// An iteration that needs one of the rare paths (a near-tie among the screen's candidates, a tree beyond the shadow, headings outside
// [-PI, PI], more than eight states, a state outside the grid or in a crowded cell) is flagged `slow`; the round is cut in front of it
// and it runs through se2_extend_try.
struct Se2Spec {
    uint32_t near;
    double q_new[3];
    int result;
    bool slow;
};
__device__ __forceinline__ double group8_min_f64(double v) {
    v = dpp_min_step<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_min_step<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    return dpp_min_step<0x141, 0xf>(v);   // row_half_mirror: every lane holds the minimum of its eight
}
__device__ __forceinline__ uint32_t group8_min_u32(uint32_t v) {
    v = dpp_umin_step<0xB1, 0xf>(v);
    v = dpp_umin_step<0x4E, 0xf>(v);
    return dpp_umin_step<0x141, 0xf>(v);
}
template <bool STAMP, class SH>
__device__ __forceinline__ void se2_round(const DevParams& p, const double* segs, const Se2Tree& t_spec, uint32_t n_spec, const Se2Tree& t_e2,
                                          uint32_t n_e2, bool pend, const double e2_q[3], const Se2Range& rg, const SH& sh, uint32_t slot0,
                                          uint32_t groups, uint32_t lane, Se2Spec& o, uint64_t* acc) {
    const uint64_t t0 = se2_clock<STAMP>();
    const uint32_t grp = lane >> 3, sub = lane & 7u;
    const uint64_t gm = 0xFFull << (grp * 8u);
    // group 0 of a round with a pending iteration extends the OTHER tree towards that iteration's new node; every other group
    // extends the tree that grows next towards its own iteration's sample (groups past the round's last repeat the first one's)
    const bool e2 = pend && grp == 0u;
    const uint32_t first = pend ? 1u : 0u;
    const uint32_t slot = slot0 + ((grp >= first && grp < groups) ? grp - first : 0u);
    const double q[3] = {e2 ? e2_q[0] : sh.q[0][slot], e2 ? e2_q[1] : sh.q[1][slot], e2 ? e2_q[2] : sh.q[2][slot]};
    const Se2Tree tree{e2 ? t_e2.g : t_spec.g, e2 ? t_e2.sh : t_spec.sh, t_spec.cap, t_spec.N};
    const uint32_t n = e2 ? n_e2 : n_spec;
    const float qx = (float)q[0], qy = (float)q[1], qt = (float)q[2];
    bool slow = !rg.theta_ok || !(fabsf(qt) <= kSe2PiUp) || n > tree.N || (p.n_segs != 0u && p.seg_grid == nullptr);
    const uint32_t ns = n < tree.N ? n : tree.N;
    const uint32_t ns_max = n_spec > n_e2 && pend ? (n_spec < tree.N ? n_spec : tree.N) : (pend ? (n_e2 < tree.N ? n_e2 : tree.N) : ns);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (this wave's stores first: se2_nearest)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float b1 = __builtin_inff(), b2 = __builtin_inff();
    uint32_t i1 = 0xFFFFFFFFu;
    const uint32_t trips = (ns_max + 7u) >> 3;   // (uniform; the shadow's size is a multiple of 32: a turn of four never leaves it, and the slots past a tree say "far")
    for (uint32_t t = 0; t < trips; t += 4u) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = tree.sh[sub + ((t + (uint32_t)u) << 3)];
#pragma unroll
        for (int u = 0; u < 4; ++u) se2_top2(b1, b2, i1, se2_screen(v[u], qx, qy, qt), sub + ((t + (uint32_t)u) << 3));
    }
    double c[3] = {0.0, 0.0, 0.0};
    if (i1 != 0xFFFFFFFFu) tree.load(i1, c);
    const float m = __uint_as_float(group8_min_u32(__float_as_uint(b1)));
    const float mag = fmaxf(rg.mag, fmaxf(fabsf(qx), fabsf(qy)));
    const float thr = m + (mag + 3.2f) * (2.02f * 32.0f * 0x1p-24f);
    slow = slow || (__ballot(b2 <= thr) & gm) != 0;
    const bool cand = b1 <= thr;
    Exact e{__builtin_inf(), 0xFFFFFFFFu};
    if (cand) { e.dist = se2_distance(c, q); e.idx = i1; }
    const double min_dist = group8_min_f64(e.dist);
    const uint32_t nearest = group8_min_u32(e.dist == min_dist ? e.idx : 0xFFFFFFFFu);
    const uint32_t wbits = (uint32_t)((__ballot(cand && e.idx == nearest) >> (grp * 8u)) & 0xFFull);
    const int wl = (int)(grp * 8u) + (wbits ? __builtin_ctz(wbits) : 0);
    const double q_near[3] = {__shfl(c[0], wl), __shfl(c[1], wl), __shfl(c[2], wl)};
    const uint64_t t1 = se2_clock<STAMP>();
    uint32_t nsteps;
    const int result = se2_steer(p, rg, q_near, q, min_dist, o.q_new, nsteps);
    const uint64_t t2 = se2_clock<STAMP>();
    const uint32_t S = nsteps <= 1u ? 1u : nsteps;
    bool slow_l = S > 8u, bad = false;
    if (p.n_segs != 0u && sub < S && !slow && !slow_l) {
        double x = o.q_new[0], y = o.q_new[1];
        if (nsteps > 1u) {
            double xy[2];
            lerp<2>(q_near, o.q_new, sh.tdiv[nsteps - 1u][sub], xy, 2);
            x = xy[0]; y = xy[1];
        }
        const double fx = (x - p.lo[0]) * p.seg_grid_inv[0], fy = (y - p.lo[1]) * p.seg_grid_inv[1];
        const double G = (double)p.seg_grid_G;
        if (fx >= 0.0 && fx < G && fy >= 0.0 && fy < G) {
            const uint4 ce = reinterpret_cast<const uint4*>(p.seg_grid)[(uint32_t)fy * p.seg_grid_G + (uint32_t)fx];
            if ((ce.x & 0xFFFFu) == kSegOverflow) slow_l = true;
            else {
                const uint32_t w[4] = {ce.x, ce.y, ce.z, ce.w};
#pragma unroll
                for (int k = 0; k < kSegSlots; ++k) {
                    const uint32_t id = (k & 1) ? w[k >> 1] >> 16 : w[k >> 1] & 0xFFFFu;
                    if (__ballot(id != kSegNone && !bad) == 0) break;   // (uniform: lists are short)
                    if (id != kSegNone && !bad) bad = !(point_segment_d2(segs, x, y, id) > p.seg_thr);
                }
            }
        } else slow_l = true;
    }
    o.slow = slow || (__ballot(slow_l) & gm) != 0;
    o.result = (__ballot(bad) & gm) != 0 ? 0 : result;
    o.near = nearest;
    if (STAMP) { const uint64_t t3 = se2_clock<STAMP>(); acc[1] += t1 - t0; acc[2] += t2 - t1; acc[3] += t3 - t2; acc[9] += 1; }
}

// shadow of the first nodes of a tree an earlier launch (or setup) left in HBM; folds them into the screen's range
__device__ __forceinline__ void se2_shadow_load(const Se2Tree& tree, uint32_t n, uint32_t lane, Se2Range& rg) {
    float mag = 0.0f;
    bool ok = true;
    for (uint32_t i = lane; i < n; i += 64u) {
        double c[3];
        tree.load(i, c);
        const float4 s = make_float4((float)c[0], (float)c[1], (float)c[2], 0.0f);
        if (i < tree.N) tree.sh[i] = s;
        mag = fmaxf(mag, fmaxf(fabsf(s.x), fabsf(s.y)));
        ok = ok && fabsf(s.z) <= kSe2PiUp;
    }
    mag = __uint_as_float(~wave_min_u32(~__float_as_uint(mag)));   // (maximum of non-negative values through their bit patterns)
    rg.mag = fmaxf(rg.mag, mag);
    rg.theta_ok = rg.theta_ok && __ballot(!ok) == 0;
}

template <int NS, bool LDS_SEGS, bool STAMP>
__global__ __launch_bounds__(64) void rrt_connect_se2_kernel(DevParams p) {
    const uint32_t prob = blockIdx.x, lane = threadIdx.x;
    __shared__ Se2Shared<NS, LDS_SEGS ? kSe2LdsSegs : 1> sh;
    const double* segs = p.segs;
    if (LDS_SEGS) {   // the checker's table at LDS latency
        for (uint32_t i = lane; i < 4 * p.n_segs; i += 64u) (&sh.segs[0][0])[i] = p.segs[i];
        segs = &sh.segs[0][0];
    }
    ProblemState st = p.state[prob];
    if (st.goal_node >= 0) return;
    const size_t cap = p.cap;
    const Se2Tree tree_a{p.tree + (size_t)prob * 3 * cap, sh.shadow_a, cap, (uint32_t)NS};
    const Se2Tree tree_b{p.tree_b + (size_t)prob * 3 * cap, sh.shadow_b, cap, (uint32_t)NS};
    int32_t* par_a = p.parent + (size_t)prob * cap;
    int32_t* par_b = p.parent_b + (size_t)prob * cap;
    const double goal_c[3] = {p.goal_c[(size_t)prob * 3], p.goal_c[(size_t)prob * 3 + 1], p.goal_c[(size_t)prob * 3 + 2]};
    const double goal_radius = p.goal_thr[prob];   // the radius itself: the goal test compares the compound distance

    RngWindow rng;
    rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint32_t na = st.n_nodes, nb = st.n_nodes_b;
    sh.tdiv[lane >> 3][lane & 7u] = (double)((lane & 7u) + 1u) / (double)((lane >> 3) + 1u);
    for (uint32_t i = lane; i < (uint32_t)NS; i += 64u) sh.shadow_a[i] = sh.shadow_b[i] = make_float4(1e30f, 0.0f, 0.0f, 0.0f);   // "far"
    Se2Range rg{0.0f, true};   // a solve call continues the trees an earlier one left in HBM
    se2_shadow_load(tree_a, na, lane, rg);
    se2_shadow_load(tree_b, nb, lane, rg);
    __syncthreads();
    int32_t stop = 1;
    uint64_t draws = st.draws;   // stream position after the last iteration that ran (the block sampler runs ahead of it)
    uint64_t acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t t_begin = se2_clock<STAMP>();
    uint64_t it = 0, sampled = 0;   // iterations completed; iterations sampled so far (whole blocks of 64)
    uint64_t h = uni64(st.checksum);   // (wave-uniform by construction: said so, the folds run on the scalar unit)
    // an iteration whose first extend succeeded waits for the other tree's extend ("pending"): that extend rides in group 0 of the
    // next round, and the round's other groups evaluate the iterations behind it as if it were going to fail -- nine times in ten
    // it does (configs[3]), and then nothing has changed before them; when it succeeds they are dropped
    bool pend = false, pend_gs = false;
    double pend_q[3] = {0.0, 0.0, 0.0};
    uint32_t pend_idx = 0;
    while (it < p.budget) {
        if (!pend && (na >= p.max_nodes || nb >= p.max_nodes)) { stop = 2; break; }
        const uint64_t it_s = it + (pend ? 1u : 0u);   // the first iteration whose first extend is still to come
        const bool can_spec = it_s < p.budget && !(na >= p.max_nodes || nb >= p.max_nodes);   // (the node cap is looked at before any draw)
        const bool gs = na <= nb;   // rrt_connect.rs:249-254, for the iterations from it_s on (until one of them inserts)
        const uint64_t ts = se2_clock<STAMP>();
        // sample: random_bool, then x, y, theta by random_range (lo/hi/scale[2] hold the clamped SO(2) bounds); 64 iterations at a time
        if (can_spec && it_s >= sampled) {
            const uint32_t m = p.budget - sampled < 64u ? (uint32_t)(p.budget - sampled) : 64u;
            se2_sample_block(rng, p, goal_c, m, lane, sh);
            sampled += m;
        }
        if (STAMP) acc[0] += se2_clock<STAMP>() - ts;
        const uint32_t slot = (uint32_t)it_s & 63u;
        uint32_t R = 0;
        if (can_spec) {
            R = pend ? 7u : 8u;
            if (64u - slot < R) R = 64u - slot;
            if (p.budget - it_s < (uint64_t)R) R = (uint32_t)(p.budget - it_s);
        }
        const uint32_t first = pend ? 1u : 0u, groups = first + R;
        // the tree that grows in the iterations from it_s on, and the tree the pending iteration still has to extend
        const Se2Tree t_spec{gs ? tree_a.g : tree_b.g, gs ? tree_a.sh : tree_b.sh, cap, (uint32_t)NS};
        const Se2Tree t_e2{pend_gs ? tree_b.g : tree_a.g, pend_gs ? tree_b.sh : tree_a.sh, cap, (uint32_t)NS};
        uint32_t n_spec = gs ? na : nb, n_e2 = pend_gs ? nb : na;
        Se2Spec sp;
        se2_round<STAMP>(p, segs, t_spec, n_spec, t_e2, n_e2, pend, pend_q, rg, sh, slot, groups, lane, sp, acc);
        const uint64_t tc = se2_clock<STAMP>();
        uint32_t nfast = groups;
        {
            uint64_t sm = __ballot(sp.slow) & 0x0101010101010101ull;   // lane 8 g speaks for group g
            if (groups < 8u) sm &= (1ull << (8u * groups)) - 1ull;
            if (sm) nfast = (uint32_t)__builtin_ctzll(sm) >> 3;
        }
        bool done = false;
        if (pend) {   // the pending iteration's second extend: group 0, or the whole wave on a rare path
            uint32_t near_b = 0;
            double qb[3];
            int rb;
            if (nfast == 0u) rb = se2_extend_try<STAMP>(p, segs, t_e2, n_e2, rg, pend_q, near_b, qb, sh.tdiv, acc);
            else {
                near_b = (uint32_t)__builtin_amdgcn_readlane((int)sp.near, 0);
                rb = __builtin_amdgcn_readlane(sp.result, 0);
#pragma unroll
                for (int k = 0; k < 3; ++k) qb[k] = readlane_f64(sp.q_new[k], 0);
            }
            if (rb) se2_insert(t_e2, pend_gs ? par_b : par_a, n_e2, rg, near_b, qb, lane);
            h = fnv_mix(h, (uint64_t)near_b);
#pragma unroll
            for (int k = 0; k < 3; ++k) h = fnv_mix(h, (uint64_t)__double_as_longlong(qb[k]));
            h = fnv_mix(h, (uint64_t)rb);
            if (pend_gs) nb = n_e2; else na = n_e2;
            if (rb == 2) {   // Reached: rrt_connect.rs:281-305
                const uint32_t idx_b = n_e2 - 1;
                st.goal_node = (int32_t)(pend_gs ? pend_idx : idx_b);
                st.goal_node_b = (int32_t)(pend_gs ? idx_b : pend_idx);
                done = true;
            }
            pend = false;
            it += 1;
            if (STAMP) acc[7] += 1;
            if (done) { stop = 0; break; }
            if (rb != 0 || nfast == 0u) {   // the iterations behind it were evaluated as if it failed (or a rare path cut the round): start over
                if (STAMP) acc[5] += se2_clock<STAMP>() - tc;
                continue;
            }
        }
        // the iterations it, it + 1, ... : groups first .. nfast - 1 hold their first extends
        uint32_t near_a = 0, j = first;
        double qa[3];
        int ra = 0;
        if (nfast <= first) {   // a rare path (first == 0 here: the pending case has left above): this iteration alone, by the whole wave
            if (R == 0u) continue;   // (nothing to run: the loop's head decides)
            const double q_rand[3] = {sh.q[0][slot], sh.q[1][slot], sh.q[2][slot]};
            ra = se2_extend_try<STAMP>(p, segs, t_spec, n_spec, rg, q_rand, near_a, qa, sh.tdiv, acc);
        } else {
            for (; j < nfast; ++j) {
                const int src = (int)(8u * j);
                near_a = (uint32_t)__builtin_amdgcn_readlane((int)sp.near, src);
                ra = __builtin_amdgcn_readlane(sp.result, src);
#pragma unroll
                for (int k = 0; k < 3; ++k) qa[k] = readlane_f64(sp.q_new[k], src);
                if (ra != 0) break;
                h = fnv_mix(h, gs ? 1ull : 0ull);   // an iteration whose motion was invalid: its checksum, nothing else
                h = fnv_mix(h, (uint64_t)near_a);
#pragma unroll
                for (int k = 0; k < 3; ++k) h = fnv_mix(h, (uint64_t)__double_as_longlong(qa[k]));
                h = fnv_mix(h, 0ull);
                st.iterations++;
            }
            if (STAMP) acc[10] += j - first;
            if (j == nfast) {
                draws = sh.pos_after[slot + (nfast - first) - 1u];
                it += nfast - first;
                if (STAMP) { acc[5] += se2_clock<STAMP>() - tc; acc[7] += nfast - first; }
                continue;
            }
        }
        // iteration it + (j - first): its first extend is (near_a, qa, ra)
        it += j - first;
        if (STAMP) acc[7] += j - first;
        draws = sh.pos_after[slot + (j - first)];
        h = fnv_mix(h, gs ? 1ull : 0ull);
        h = fnv_mix(h, (uint64_t)near_a);
#pragma unroll
        for (int k = 0; k < 3; ++k) h = fnv_mix(h, (uint64_t)__double_as_longlong(qa[k]));
        h = fnv_mix(h, (uint64_t)ra);
        st.iterations++;
        if (ra == 0) { it += 1; if (STAMP) { acc[5] += se2_clock<STAMP>() - tc; acc[7] += 1; } continue; }
        se2_insert(t_spec, gs ? par_a : par_b, n_spec, rg, near_a, qa, lane);
        if (gs) na = n_spec; else nb = n_spec;
        if (gs && se2_distance(qa, goal_c) <= goal_radius) {   // rrt_connect.rs:271-274
            st.goal_node = (int32_t)(n_spec - 1);
            st.goal_node_b = -1;
            it += 1;
            stop = 0;
            break;
        }
        pend = true;
        pend_gs = gs;
        pend_idx = n_spec - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) pend_q[k] = qa[k];
        if (STAMP) acc[5] += se2_clock<STAMP>() - tc;
    }
    st.checksum = h;
    if (STAMP && prob == 0 && lane == 0 && p.dbg) {
        acc[6] = se2_clock<STAMP>() - t_begin;
        for (int k = 0; k < 12; ++k) p.dbg[k] = acc[k];
    }
    if (lane == 0) {
        st.n_nodes = na;
        st.n_nodes_b = nb;
        st.draws = draws;
        st.stop_reason = stop;
        p.state[prob] = st;
    }
}

void launch_rrt_connect_se2(const DevParams& p, hipStream_t stream) {
    const dim3 grid(p.n_problems), block(64);
    const bool lds = p.n_segs <= (uint32_t)kSe2LdsSegs;
    if (p.dbg) {   // diagnostic instantiation (cycle stamps of problem 0)
        if (lds) hipLaunchKernelGGL((rrt_connect_se2_kernel<kSe2N, true, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((rrt_connect_se2_kernel<kSe2N, false, true>), grid, block, 0, stream, p);
        return;
    }
    // more problems than SIMDs (or the test switch): the small-LDS shape, seven waves per CU
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (p.n_problems > 4u * (uint32_t)cus || (p.dbg_flags & OXHIP_DEBUG_SE2_SMALL_LDS) != 0)
        hipLaunchKernelGGL((rrt_connect_se2_kernel<kSe2NSmall, false, false>), grid, block, 0, stream, p);
    else if (lds) hipLaunchKernelGGL((rrt_connect_se2_kernel<kSe2N, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((rrt_connect_se2_kernel<kSe2N, false, false>), grid, block, 0, stream, p);
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
    const bool any = se2_motion_invalid_wave(p, p.segs, f, g, num_steps_u32(se2_distance(f, g), p.res), lane);
    if (lane == 0) out[m] = any ? 0 : 1;
}
void launch_se2_check_motion(const DevParams& p, const double* from, const double* to, uint32_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(se2_check_motion_kernel, dim3((n + 3) / 4), dim3(256), 0, s, p, from, to, n, out);
}

}  // namespace oxhip
