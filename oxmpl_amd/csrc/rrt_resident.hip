// rrt_resident.hip -- register-resident RRT grow kernel for gfx950.
//
// One 1024-thread workgroup (16 wave64 = the whole register file of one CU) per problem.
// Thread t keeps tree nodes {t + 1024*s : s < S} in its VGPRs (S*DIM f64 = 60 VGPRs for
// R^3, 10,240 nodes), so the O(n) nearest-neighbour scan of rrt.rs:187-196 reads no memory
// at all: it is bounded by f64 VALU issue, not by HBM.  The SoA tree in HBM is only the
// persistent copy (written once per insert, read once at launch).
//
// Per iteration: every wave scans its 640 nodes (d2 compare, no sqrt), reduces with DPP,
// publishes (d2, index, candidate coordinates) to LDS; after one barrier every wave derives
// the same nearest node, steers, and the 16 waves split the motion check's interpolated
// states (one state per wave, one obstacle per lane); a second barrier ORs the verdicts.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

namespace oxhip {

constexpr int kResThreads = 1024;
constexpr int kResWaves = kResThreads / 64;

// ---- wave64 min of an f64 with DPP (VALU only, no LDS crossbar); result is wave-uniform
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min_step(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl = false + old = own value: lanes without a source keep their own value
    int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmin(v, __hiloint2double(ohi, olo));
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

__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bound3(double g) {
    return __longlong_as_double((long long)((uint64_t)__double_as_longlong(g) + 3));
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t hi32(double v) { return (uint32_t)__double2hiint(v); }

// Per-lane scan state.  d2 >= +0, so the high dword orders like the value: h2 tracks the second
// smallest HIGH DWORD (with multiplicity) in one v_med3_u32 per node.  Two d2 whose square roots
// could coincide differ by <= 3 ulps, hence their high dwords differ by <= 1: `h2 <= hi(b1) + 1`
// is a conservative (never missing, ~1e-6 false-positive) near-tie detector.
struct Scan {
    double b1;      // smallest d2 of this lane
    uint32_t slot;  // its slot (lowest index: slots are visited in increasing index order)
    uint32_t h2;    // second smallest high dword
};
__device__ __forceinline__ void scan_push(Scan& v, double d, uint32_t s) {
    v.h2 = umed3(hi32(d), hi32(v.b1), v.h2);
    const bool lt = d < v.b1;  // strict: the earlier (lower) index keeps exact ties
    v.b1 = lt ? d : v.b1;
    v.slot = lt ? s : v.slot;
}

template <int DIM>
struct WavePub {      // one wave's nearest-neighbour candidate
    double b1;        // its smallest d2
    uint32_t i1;      // lowest index attaining it
    uint32_t amb;     // the wave saw another d2 whose high dword is within 1 of b1's
    double c[DIM];    // the candidate's coordinates (written by the owning lane)
};
template <int DIM>
struct WaveExact {
    double dist;
    uint32_t idx;
    uint32_t pad;
    double c[DIM];
};

// next query, written by the sampler wave one iteration ahead
template <int DIM>
struct QRec {
    double q[DIM];
    uint64_t pos_after;  // stream position after this query's draws
};

// the resolver's verdict for the insert step
template <int DIM>
struct Work {
    double q_new[DIM];
    uint32_t nearest;
    uint32_t ok;
    uint32_t mode;   // 0 = resolved, 1 = exact re-scan requested
    uint32_t pad;
};

// lane-predicated store of the owning lane's slot `slot` (wave-uniform) into LDS
template <int DIM, int S>
__device__ __forceinline__ void store_slot(const double (&tr)[DIM][S], uint32_t slot, bool mine, double* dst) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (slot == (uint32_t)s) {  // uniform branch; indices below are compile-time constants
            if (mine) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) dst[k] = tr[k][s];
            }
        }
    }
}

// steer (rrt.rs:199-208) + check_motion (rrt.rs:90-116) by the resolver wave: one obstacle per
// lane from the LDS table.  A conservative midpoint filter settles most motions with a single
// distance per sphere: every interpolated state lies within max_distance/2 of the segment
// midpoint, so d2(centre, mid) > (r + max_distance/2 + margin)^2 proves the sphere cannot be hit.
// Only if some sphere fails the filter are the interpolated states tested, exactly as the
// reference does (the verdict is identical either way; the filter only skips provably valid work).
template <int DIM>
__device__ __forceinline__ void resolve_tail(const DevParams& p, uint32_t lane, uint32_t nearest, bool have_dist,
                                             double g_or_dist, const double q_near[DIM], const double q[DIM],
                                             const double (*obs)[64], uint32_t ns64, Work<DIM>& work) {
    double q_new[DIM];
    const bool far = have_dist ? (g_or_dist > p.max_distance) : (g_or_dist > p.t_steer);
    if (far) {
        const double md = have_dist ? g_or_dist : sqrt(g_or_dist);
        const double t = p.max_distance / md;
        lerp<DIM>(q_near, q, t, q_new, DIM);
    } else {
#pragma unroll
        for (int k = 0; k < DIM; ++k) q_new[k] = q[k];
    }
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    bool ok = true;
    if (nobs > 0) {
        double c[DIM];
#pragma unroll
        for (int k = 0; k < DIM; ++k) c[k] = obs[k][lane];
        const double thr = obs[DIM][lane], filt = obs[DIM + 1][lane];
        double mid[DIM];
        lerp<DIM>(q_near, q_new, 0.5, mid, DIM);
        const bool maybe = !(dist2<DIM>(c, mid, DIM) > filt);
        const bool extras = nobs > ns64;  // spheres beyond the first 64 and every box: always stepped
        if (__ballot(maybe) != 0 || extras) {
            const double dist = sqrt(dist2<DIM>(q_near, q_new, DIM));
            const uint32_t nsteps = num_steps_u32(dist, p.res);
            bool bad = false;
            if (nsteps <= 1) {
                bad = !(dist2<DIM>(c, q_new, DIM) > thr);
                for (uint32_t j = ns64 + lane; j < nobs; j += 64) bad = bad || obstacle_hit<DIM>(p, DIM, q_new, j);
            } else {
                const double dn = (double)nsteps;
                const double tl = (double)(lane + 1) / dn;  // lane s-1 holds s / nsteps (one division, all lanes)
                for (uint32_t s = 1; s <= nsteps; ++s) {
                    const double t = (s <= 64) ? readlane_f64(tl, (int)(s - 1)) : ((double)s / dn);
                    double x[DIM];
                    lerp<DIM>(q_near, q_new, t, x, DIM);
                    bad = bad || !(dist2<DIM>(c, x, DIM) > thr);
                    for (uint32_t j = ns64 + lane; j < nobs; j += 64) bad = bad || obstacle_hit<DIM>(p, DIM, x, j);
                    if (__ballot(bad) != 0) break;  // the reference also stops at the first invalid state
                }
            }
            ok = __ballot(bad) == 0;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) work.q_new[k] = q_new[k];
        work.nearest = nearest;
        work.ok = ok ? 1u : 0u;
        work.mode = 0;
    }
}

template <int DIM, int S, bool STAMP>
__global__ __launch_bounds__(kResThreads) void rrt_resident_kernel(DevParams p) {
    constexpr int D = DIM;
    constexpr uint32_t kResolver = 0, kSampler = kResWaves - 1;
    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;

    __shared__ uint32_t rng_buf[16][64];
    __shared__ WavePub<DIM> pub[kResWaves];
    __shared__ WaveExact<DIM> epub[kResWaves];
    __shared__ QRec<DIM> qrec[2];
    __shared__ Work<DIM> work;
    __shared__ double obs[DIM + 2][64];  // first 64 spheres: centre, validity threshold, filter threshold

    ProblemState st = p.state[prob];
    if (p.stop_at_goal && st.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    const double goal_thr = p.goal_thr[prob];
    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;

    uint32_t n = st.n_nodes;

    // the tree, in registers: node (tid + 1024*s) in tr[.][s]; empty slots hold +inf (d2 = inf never wins)
    double tr[DIM][S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint32_t i = tid + kResThreads * s;
#pragma unroll
        for (int k = 0; k < DIM; ++k) tr[k][s] = (i < n) ? tree[(size_t)k * cap + i] : __builtin_inf();
    }
    if (tid < 64) {  // unused lanes hold a sphere that can never be hit
#pragma unroll
        for (int k = 0; k < D; ++k) obs[k][tid] = tid < ns64 ? p.sph_c[(size_t)k * p.n_spheres + tid] : 0.0;
        obs[D][tid] = tid < ns64 ? p.sph_thr[tid] : -1.0;
        obs[D + 1][tid] = tid < ns64 ? p.sph_filt[tid] : -1.0;
    }

    // the sampler wave owns the RNG window and works one query ahead (rrt.rs:177-184)
    RngWindow rng;
    rng.init(rng_buf, p.seed, p.first_problem_id + prob, st.draws);
    uint64_t draws_done = st.draws;
    if (wave == kSampler && p.budget > 0) {
        double q0[D];
        sample_state<D, false>(rng, p, DIM, goal_c, q0);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) qrec[0].q[k] = q0[k];
            qrec[0].pos_after = rng.pos;
        }
    }
    __syncthreads();

    uint64_t t_scan = 0, t_b1 = 0, t_res = 0, t_b2 = 0, t_ins = 0, t_mark = 0, wave_arr = 0, t_rel = 0;
#define OXHIP_STAMP(acc)                                   \
    if (STAMP) {                                           \
        uint64_t now_ = (uint64_t)clock64();               \
        acc += now_ - t_mark;                              \
        t_mark = now_;                                     \
    }
    if (STAMP) { t_mark = (uint64_t)clock64(); t_rel = t_mark; }

    uint32_t par = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    for (uint64_t it = 0; it < p.budget; ++it, par ^= 1) {
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }

        // ---- phase A (all waves): scan the register tree for this iteration's query (rrt.rs:187-196)
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q[k] = qrec[par].q[k];
        const uint32_t nslots = uni((n + kResThreads - 1) / kResThreads);
        Scan sc{__builtin_inf(), 0u, 0xFFFFFFFFu};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if ((uint32_t)s < nslots) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = tr[k][s];
                scan_push(sc, dist2<D>(c, q, DIM), (uint32_t)s);
            }
        }
        {
            const double wmin = wave_min_f64(sc.b1);
            const uint64_t eqm = __ballot(sc.b1 == wmin);
            const int wl = eqm ? (__ffsll((unsigned long long)eqm) - 1) : 0;
            const uint32_t wslot = __builtin_amdgcn_readlane(sc.slot, wl);
            const uint32_t hb = hi32(wmin) + 1;
            const bool amb_l = ((int)lane != wl && hi32(sc.b1) <= hb) || (sc.h2 <= hb);
            const uint32_t wamb = __ballot(amb_l) != 0 ? 1u : 0u;
            store_slot<DIM, S>(tr, wslot, (int)lane == wl, pub[wave].c);
            if (lane == 0) {
                pub[wave].b1 = wmin;
                pub[wave].i1 = (wave << 6) + (uint32_t)wl + (wslot << 10);
                pub[wave].amb = wamb;
            }
        }
        if (STAMP) wave_arr += (uint64_t)clock64() - t_rel;
        OXHIP_STAMP(t_scan)
        __syncthreads();
        OXHIP_STAMP(t_b1)

        // ---- phase B: wave 0 resolves (nearest over 16 candidates, steer, motion check);
        //      wave 15 samples the next query meanwhile; the others wait
        if (wave == kResolver) {
            const bool in = lane < kResWaves;
            const double pb = in ? pub[in ? lane : 0].b1 : __builtin_inf();
            const uint32_t pamb = in ? pub[in ? lane : 0].amb : 0u;
            const double g = wave_min_f64(pb);
            const uint64_t m2 = __ballot(in && pb == g);
            const int ww = m2 ? (__ffsll((unsigned long long)m2) - 1) : 0;
            const uint32_t hb = hi32(g) + 1;
            const bool amb = (__popcll(m2) > 1) || (__ballot(in && (pamb != 0 || ((int)lane != ww && hi32(pb) <= hb))) != 0);
            if (!amb) {
                double q_near[D];
#pragma unroll
                for (int k = 0; k < D; ++k) q_near[k] = pub[ww].c[k];
                resolve_tail<DIM>(p, lane, pub[ww].i1, false, g, q_near, q, obs, ns64, work);
            } else if (lane == 0) {
                work.mode = 1;
            }
        } else if (wave == kSampler) {
            if (it + 1 < p.budget) {
                double qn[D];
                sample_state<D, false>(rng, p, DIM, goal_c, qn);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < D; ++k) qrec[par ^ 1].q[k] = qn[k];
                    qrec[par ^ 1].pos_after = rng.pos;
                }
            }
        }
        OXHIP_STAMP(t_res)
        __syncthreads();
        if (uni(work.mode) != 0) {
            // rare: two d2 with (nearly) equal high dwords -> post-sqrt compare with lowest-index ties,
            // literally as the reference does
            Exact e{__builtin_inf(), 0xFFFFFFFFu};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if ((uint32_t)s < nslots) {
                    double c[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = tr[k][s];
                    double d = sqrt(dist2<D>(c, q, DIM));
                    if (d < e.dist) { e.dist = d; e.idx = tid + kResThreads * s; }
                }
            }
            e = exact_wave_reduce(e);
            const uint32_t eidx = uni(e.idx);
            store_slot<DIM, S>(tr, eidx >> 10, lane == (eidx & 63u), epub[wave].c);
            if (lane == 0) {
                epub[wave].dist = unid(e.dist);
                epub[wave].idx = eidx;
            }
            __syncthreads();
            if (wave == kResolver) {
                int bw = 0;
                Exact be{epub[0].dist, epub[0].idx};
                for (int w = 1; w < kResWaves; ++w) {
                    Exact o{epub[w].dist, epub[w].idx};
                    if ((o.dist < be.dist) || (o.dist == be.dist && o.idx < be.idx)) { be = o; bw = w; }
                }
                bw = (int)uni((uint32_t)bw);
                double q_near[D];
#pragma unroll
                for (int k = 0; k < D; ++k) q_near[k] = epub[bw].c[k];
                resolve_tail<DIM>(p, lane, uni(be.idx), true, unid(be.dist), q_near, q, obs, ns64, work);
            }
            __syncthreads();
        }
        if (STAMP) t_rel = (uint64_t)clock64();
        OXHIP_STAMP(t_b2)

        // ---- phase D (all waves): verdict, insert into the owner's registers, goal test
        const bool ok = uni(work.ok) != 0;
        const uint32_t nearest = uni(work.nearest);
        double q_new[D];
#pragma unroll
        for (int k = 0; k < D; ++k) q_new[k] = work.q_new[k];
        draws_done = qrec[par].pos_after;
        if (wave == 0) {  // bookkeeping is only ever read back from thread 0
            uint64_t h = fnv_mix(st.checksum, (uint64_t)nearest);
#pragma unroll
            for (int k = 0; k < D; ++k) h = fnv_mix(h, uni64((uint64_t)__double_as_longlong(q_new[k])));
            st.checksum = fnv_mix(h, ok ? 1ull : 0ull);
            st.iterations++;
            if (ok) st.accepted++;
        }
        bool hit = false;
        if (ok && !p.freeze) {
            // 6. insert (rrt.rs:213-217): the owner thread takes the node into its registers
            const uint32_t slot = n >> 10;
            const bool owner = tid == (n & (kResThreads - 1));
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (slot == (uint32_t)s) {
#pragma unroll
                    for (int k = 0; k < D; ++k) tr[k][s] = owner ? q_new[k] : tr[k][s];
                }
            }
            if (owner) {
#pragma unroll
                for (int k = 0; k < D; ++k) tree[(size_t)k * cap + n] = q_new[k];
                parent[n] = (int32_t)nearest;
            }
            ++n;
            // 7. goal test (rrt.rs:220-223)
            if (dist2<D>(q_new, goal_c, DIM) <= goal_thr) {
                if (st.goal_node < 0) st.goal_node = (int32_t)(n - 1);
                hit = true;
            }
        }
        OXHIP_STAMP(t_ins)
        if (hit && p.stop_at_goal) { stop = 0; break; }
    }
#undef OXHIP_STAMP

    if (STAMP && p.dbg && prob == 0 && lane == 0) p.dbg[16 + wave] = wave_arr;
    if (tid == 0) {
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
        if (STAMP && p.dbg && prob == 0) {
            p.dbg[0] = t_scan; p.dbg[1] = t_b1; p.dbg[2] = t_res; p.dbg[3] = t_b2; p.dbg[4] = 0;
            p.dbg[5] = 0; p.dbg[6] = t_ins; p.dbg[7] = st.iterations;
        }
    }
}

// instantiations: (dim, slots) -> capacity 1024 * slots nodes
static int pick_slots(uint32_t cap) {
    const uint32_t need = (cap + kResThreads - 1) / kResThreads;
    if (need <= 2) return 2;
    if (need <= 4) return 4;
    if (need <= 10) return 10;
    return 0;
}

bool resident_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pick_slots(cap) != 0; }

void launch_rrt_resident(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kResThreads);
    const int s = pick_slots(p.cap);
#define OXHIP_LAUNCH(DIM_, S_)                                                                              \
    do {                                                                                                    \
        if (p.dbg) hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, true>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, false>), grid, block, 0, stream, p);         \
    } while (0)
    if (p.dim == 3) {
        if (s == 2) OXHIP_LAUNCH(3, 2); else if (s == 4) OXHIP_LAUNCH(3, 4); else OXHIP_LAUNCH(3, 10);
    } else {
        if (s == 2) OXHIP_LAUNCH(2, 2); else if (s == 4) OXHIP_LAUNCH(2, 4); else OXHIP_LAUNCH(2, 10);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
