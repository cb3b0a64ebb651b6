// rrt_pruned.hip -- register-resident RRT kernel with exact spatial pruning of the nearest-neighbour
// scan (rrt.rs:187-196).  Same pipeline as rrt_resident.hip (8 scanner waves + 1 resolver wave, LDS
// ring, no barrier while running), plus:
//
//  * (Re)build phase: the workgroup counting-sorts its live nodes by a 12-bit Morton cell key and the
//    scanner lanes gather them in that order, so every register row (one slot of one wave = 64
//    consecutive sorted positions) holds spatial neighbours and carries a tight axis-aligned box.
//    Nodes committed afterwards are appended behind the sorted ones; once they exceed
//    max(512, n/8) the resolver drains the ring, the workgroup meets at a barrier and re-sorts.
//  * Pruned scan: lane r of a wave holds the box of the wave's row r.  For a query every lane computes the
//    smallest (lb) and the largest (ub) squared distance from the query to its box; some node of the wave is
//    no farther than min_r ub, so only rows with lb <= min_r ub * (1 + 2^-30) can hold the wave's nearest
//    node or one tying with it after sqrt (ties differ by <= 3 ulps).  Those rows -- typically 2-3 of 21,
//    because consecutive chunks of the Morton order are dealt to the eight waves in turn, so every wave
//    covers the whole space sparsely -- are scanned under one static, wave-uniform branch each: no
//    dynamic register indexing, no dispatch ladder.  The (d2, index, near-tie flag) a wave publishes is
//    exactly what the full scan would publish, and everything downstream is unchanged.
//
// Row order inside a lane is no longer index order, so an exact d2 tie inside one lane is reported
// as ambiguous (it already is: the second-smallest-high-dword detector fires on equality) and takes
// the reference's literal post-sqrt loop.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"

#ifndef OXHIP_PRUNE_PER_QUERY
#define OXHIP_PRUNE_PER_QUERY 1   // 1: a selected row is scanned only for the queries that selected it; 0: for the whole pass
#endif

namespace oxhip {

constexpr int kCellBits = 12;                 // Morton key width: 4 bits per axis in R^3, 6 in R^2
constexpr int kCells = 1 << kCellBits;
constexpr uint32_t kNoResort = 0xFFFFFFFFu;

template <int S>
struct SortShared {
    uint32_t hist[kCells];                    // counting sort: histogram, then running offsets
    uint16_t perm[Layout<S>::kCapacity];      // sorted position -> node index (build phase only)
    uint16_t idx_tab[S][kScanThreads];        // (slot, scanner thread) -> node index
    double box[kScanWaves][2 * 3][64];         // [wave][lo_k / hi_k][row]: the bounding box of every register row
    uint32_t wave_tot[kPipeThreads / 64];
    uint32_t n_sorted;
    uint32_t resort_at;                       // scanners stop before this query and meet the resolver at the barrier
};

__device__ __forceinline__ double vmax_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_max_f64(double v) { return -wave_min_f64(-v); }

// squared distance from q to an axis-aligned box (0 inside); an empty box (lo = +inf, hi = -inf) gives +inf
template <int DIM>
__device__ __forceinline__ double box_lb2(const double lo[DIM], const double hi[DIM], const double q[DIM]) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double d = vmax_f64(vmax_f64(lo[k] - q[k], 0.0), q[k] - hi[k]);
        acc = acc + d * d;
    }
    return acc;
}

// largest squared distance from q to the box (farthest corner); an empty box gives +inf
template <int DIM>
__device__ __forceinline__ double box_ub2(const double lo[DIM], const double hi[DIM], const double q[DIM]) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const double a = q[k] - lo[k], b = hi[k] - q[k];
        const double d = vmax_f64(fabs(a), fabs(b));
        acc = acc + d * d;
    }
    return acc;
}

template <int DIM>
__device__ __forceinline__ uint32_t cell_key(const DevParams& p, const double c[DIM]) {
    constexpr int B = kCellBits / DIM;        // bits per axis
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        double u = (c[k] - p.lo[k]) / p.scale[k] * (double)(1 << B);
        int v = (u > 0.0) ? (int)u : 0;       // NaN / below the bounds -> cell 0
        v = v > (1 << B) - 1 ? (1 << B) - 1 : v;
#pragma unroll
        for (int b = 0; b < B; ++b) key |= (uint32_t)((v >> b) & 1) << (b * DIM + k);
    }
    return key;
}

// scan slot `slot` (wave-uniform) for one query: uniform two-level dispatch ending in static indices
template <int DIM, int S>
__device__ __forceinline__ void scan_row(const double (&tr)[DIM][S], uint32_t slot, const double q[DIM], Scan& sc) {
    const uint32_t grp = slot >> 2, sub = slot & 3u;
#pragma unroll
    for (int g = 0; g < (S + 3) / 4; ++g) {
        if (grp == (uint32_t)g) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (4 * g + t < S) {
                    if (sub == (uint32_t)t) {
                        double c[DIM];
#pragma unroll
                        for (int k = 0; k < DIM; ++k) c[k] = tr[k][4 * g + t];
                        scan_push(sc, dist2<DIM>(c, q, DIM), (uint32_t)(4 * g + t));
                    }
                }
            }
        }
    }
}

// the resolver's fetch of a scan winner's coordinates (see node_coord in rrt_resident.hip)
template <int DIM>
__device__ __forceinline__ double node_coord_p(const PipeShared<DIM>& sh, const double* tree, size_t cap, uint32_t n_start,
                                               uint32_t n_now, int k, uint32_t i) {
    if (i >= n_start && i + 64u >= n_now) return sh.newn[i & 63][k];
    return __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DIM, int S, bool STAMP>
__global__ __launch_bounds__(kPipeThreads) void rrt_pruned_kernel(DevParams p) {
    constexpr int D = DIM;
    using Lay = Layout<S>;
    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;
    const bool is_scanner = wave < (uint32_t)kScanWaves;

    __shared__ PipeShared<DIM> sh;
    __shared__ SortShared<S> ss;

    const ProblemState st0 = p.state[prob];
    if (p.stop_at_goal && st0.goal_node >= 0) return;

    const size_t cap = p.cap;
    double* tree = p.tree + (size_t)prob * DIM * cap;
    int32_t* parent = p.parent + (size_t)prob * cap;
    uint8_t* skip = p.skip + (size_t)prob * cap;
    const uint32_t budget = (uint32_t)p.budget;  // the host keeps a launch's budget below 2^31

    if (tid < kRing) sh.done[tid] = 0;
    if (tid == 0) {
        sh.sampled = 0;
        sh.resolved = 0;
        sh.committed = st0.n_nodes;
        sh.stop_flag = 0;
        ss.resort_at = kNoResort;
    }

    __syncthreads();

    // Each role owns its loop and its registers (declaring both at kernel scope would make every wave
    // carry the union).  Both branches execute the same sequence of workgroup barriers:
    // kBuildBarriers per (re)build, one at the end of every run.
    constexpr int kBuildBarriers = 6;
    if (is_scanner) {
        // ---- scanner state: the tree rows, their boxes (lane r holds the box of row r of this wave), the next query
        double tr[DIM][S];
        uint32_t j = 0, n_local = 0;
        uint32_t n_base = 0, pos_base = 0;  // node i >= n_base sits at sorted position pos_base + (i - n_base)
        uint64_t ts_wait = 0, ts_pre = 0, ts_scan = 0, ts_red = 0, ts_rows = 0, t_mark = 0;
        uint64_t n_sort = 0, t_sort = 0;
        for (;;) {
            // =========================================================== (re)build: the 512 scanner threads
                uint64_t t_b0 = STAMP ? (uint64_t)clock64() : 0;
            const uint32_t n0 = uni(lds_peek(&sh.committed));
            for (uint32_t c = tid; c < (uint32_t)kCells; c += kScanThreads) ss.hist[c] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < n0; i += kScanThreads) {
                if (skip[i] == 0) {
                    double c[D];
    #pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = tree[(size_t)k * cap + i];
                    atomicAdd(&ss.hist[cell_key<DIM>(p, c)], 1u);
                }
            }
            __syncthreads();
            {   // exclusive prefix sum over the 4096 bins: 8 bins per thread of waves 0..7
                uint32_t local[8], sum = 0;
                if (tid < (uint32_t)kScanThreads) {
    #pragma unroll
                    for (int b = 0; b < 8; ++b) { local[b] = ss.hist[tid * 8 + b]; sum += local[b]; }
                }
                uint32_t incl = sum;
    #pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
                    if ((int)lane >= d) incl += o;
                }
                if (lane == 63) ss.wave_tot[wave] = incl;
                __syncthreads();
                uint32_t base = 0;
                for (uint32_t w = 0; w < wave; ++w) base += ss.wave_tot[w];
                if (tid < (uint32_t)kScanThreads) {
                    uint32_t run = base + incl - sum;
    #pragma unroll
                    for (int b = 0; b < 8; ++b) { ss.hist[tid * 8 + b] = run; run += local[b]; }
                }
                if (tid == 0) {
                    uint32_t tot = 0;
                    for (int w = 0; w < kScanWaves; ++w) tot += ss.wave_tot[w];
                    ss.n_sorted = tot;
                    ss.resort_at = kNoResort;
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < n0; i += kScanThreads) {
                if (skip[i] == 0) {
                    double c[D];
    #pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = tree[(size_t)k * cap + i];
                    const uint32_t pos = atomicAdd(&ss.hist[cell_key<DIM>(p, c)], 1u);
                    ss.perm[pos] = (uint16_t)i;
                }
            }
            __syncthreads();
            const uint32_t n_sorted = uni(ss.n_sorted);
            n_base = n0;
            pos_base = n_sorted;
            {
                n_local = n0;
    #pragma unroll
                for (int s = 0; s < S; ++s) {
                    const uint32_t pos = ((uint32_t)s < Lay::kCommon || Lay::heavy(wave)) ? Lay::node_index(wave, lane, (uint32_t)s) : kNoNode;
                    const bool live = pos < n_sorted;
                    const uint32_t idx = live ? (uint32_t)ss.perm[live ? pos : 0] : 0u;
                    ss.idx_tab[s][tid] = (uint16_t)idx;
    #pragma unroll
                    for (int k = 0; k < DIM; ++k) tr[k][s] = live ? tree[(size_t)k * cap + idx] : __builtin_inf();
                }
                // boxes in LDS: entry r of this wave = the box of its row r (+inf / -inf when the row is empty)
    #pragma unroll
                for (int k = 0; k < D; ++k) { ss.box[wave][k][lane] = __builtin_inf(); ss.box[wave][D + k][lane] = -__builtin_inf(); }
    #pragma unroll
                for (int s = 0; s < S; ++s) {
    #pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double v = tr[k][s];
                        const double mn = wave_min_f64(v);
                        const double mx = wave_max_f64(v < __builtin_inf() ? v : -__builtin_inf());
                        if (lane == 0) { ss.box[wave][k][s] = mn; ss.box[wave][D + k][s] = mx; }
                    }
                }
            }
            __syncthreads();  // perm / hist are free again; idx_tab is complete
                if (STAMP) { ++n_sort; t_sort += (uint64_t)clock64() - t_b0; t_mark = (uint64_t)clock64(); }

            // ============================================================= run until stop / re-sort
            uint32_t seen_sampled = uni(lds_peek(&sh.sampled));
            for (;;) {
                // the pass covers queries [j, j + nb): up to kBatch, never across the budget or a re-sort point
                uint32_t limit = budget;
                bool leave = false;
                uint32_t nb = 0;
                for (uint32_t spins = 0;; ++spins) {
                    const uint32_t ra = uni(lds_peek(&ss.resort_at));
                    limit = ra < budget ? ra : budget;
                    if (j >= limit || lds_peek(&sh.stop_flag) != 0 || spins > kMaxSpins) { leave = true; break; }
                    nb = (limit - j < (uint32_t)kBatch) ? (limit - j) : (uint32_t)kBatch;
                    if (seen_sampled >= j + nb) break;
                    seen_sampled = uni(lds_peek(&sh.sampled));
                    if (seen_sampled >= j + nb) break;
                    // fewer than a full pass sampled: take what there is once the sampler has stopped at a re-sort point
                    if (ra != kNoResort && seen_sampled > j) { nb = seen_sampled - j; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (leave) break;
                if (STAMP) { uint64_t now = (uint64_t)clock64(); ts_wait += now - t_mark; t_mark = now; }
                // absorb the nodes committed since this wave's last snapshot
                const uint32_t nc = uni(lds_peek(&sh.committed));
                for (uint32_t i = n_local; i < nc; ++i) {
                    uint32_t owner_thread, sl;
                    Lay::locate(pos_base + (i - n_base), owner_thread, sl);
                    const bool mine = tid == owner_thread;
                    const bool my_wave = (owner_thread >> 6) == wave;
                    double c[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) c[k] = unid(sh.newn[i & 63][k]);
                    if (my_wave) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            if (sl == (uint32_t)s) {
                                if (mine) {
#pragma unroll
                                    for (int k = 0; k < D; ++k) tr[k][s] = c[k];
                                    ss.idx_tab[s][tid] = (uint16_t)i;
                                }
                            }
                        }
                        if (lane == sl && c[0] < __builtin_inf()) {   // grow the row's box (skipped duplicates are +inf)
#pragma unroll
                            for (int k = 0; k < D; ++k) {
                                const double l0 = ss.box[wave][k][sl], h0 = ss.box[wave][D + k][sl];
                                ss.box[wave][k][sl] = c[k] < l0 ? c[k] : l0;
                                ss.box[wave][D + k][sl] = c[k] > h0 ? c[k] : h0;
                            }
                        }
                    }
                }
                n_local = nc;
                if (STAMP) { uint64_t now = (uint64_t)clock64(); ts_pre += now - t_mark; t_mark = now; }
                // ---- row selection: lane r holds the box of row r; per query one lb, one ub, one wave minimum
                double q[kBatch][D];
                uint64_t rows[kBatch];          // wave-uniform: bit r set = row r must be scanned for query b
                {
                    double blo[D], bhi[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) { blo[k] = ss.box[wave][k][lane]; bhi[k] = ss.box[wave][D + k][lane]; }
#pragma unroll
                    for (int b = 0; b < kBatch; ++b) {
                        const uint32_t slot = (j + ((uint32_t)b < nb ? (uint32_t)b : 0u)) & (kRing - 1);
#pragma unroll
                        for (int k = 0; k < D; ++k) q[b][k] = unid(sh.qring[slot].q[k]);
                        const double lb = box_lb2<D>(blo, bhi, q[b]);
                        const double ub = box_ub2<D>(blo, bhi, q[b]);          // +inf for an empty row
                        const double um = wave_min_f64(ub);
                        const double bound = __builtin_fma(um, 0x1p-30, um);   // * (1 + 2^-30); 2^-30 is an inline literal
                        rows[b] = (uint32_t)b < nb ? __ballot(lb <= bound) : 0ull;   // empty rows: lb = +inf, never selected
                    }
                }
                const uint64_t any_rows = rows[0] | rows[1] | rows[2] | rows[3];
                if (STAMP) ts_rows += (uint64_t)__popcll(rows[0]);
                Scan sc[kBatch];
#pragma unroll
                for (int b = 0; b < kBatch; ++b) sc[b] = Scan{__builtin_inf(), 0u, 0xFFFFFFFFu};
                // ---- scan: static loop over the rows, wave-uniform branches on the selection bits
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if ((any_rows >> s) & 1ull) {
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = tr[k][s];
#pragma unroll
                        for (int b = 0; b < kBatch; ++b) {
#if OXHIP_PRUNE_PER_QUERY
                            if ((rows[b] >> s) & 1ull)
#endif
                                scan_push(sc[b], dist2<D>(c, q[b], DIM), (uint32_t)s);
                        }
                    }
                }
                if (STAMP) { uint64_t now = (uint64_t)clock64(); ts_scan += now - t_mark; t_mark = now; }
                // ---- reduce + publish (d2, index, near-tie flag); the resolver fetches coordinates itself (node_coord)
#pragma unroll
                for (int b = 0; b < kBatch; ++b) {
                    if ((uint32_t)b < nb) {
                        const uint32_t slot = (j + (uint32_t)b) & (kRing - 1);
                        const double wmin = wave_min_f64(sc[b].b1);
                        const uint64_t eqm = __ballot(sc[b].b1 == wmin);
                        const int wl = eqm ? (__ffsll((unsigned long long)eqm) - 1) : 0;
                        const uint32_t wslot = __builtin_amdgcn_readlane(sc[b].slot, wl);
                        const uint32_t hb = hi32(wmin) + 1;
                        const bool amb_l = ((int)lane != wl && hi32(sc[b].b1) <= hb) || (sc[b].h2 <= hb);
                        const uint32_t wamb = __ballot(amb_l) != 0 ? 1u : 0u;
                        if (lane == 0) {
                            WavePub<DIM>& out = sh.pub[slot][wave];
                            out.b1 = wmin;
                            out.i1 = wmin < __builtin_inf() ? (uint32_t)ss.idx_tab[wslot < (uint32_t)S ? wslot : 0][(wave << 6) + (uint32_t)wl] : kNoNode;
                            out.amb_nc = (nc << 1) | wamb;
                            lds_bump(&sh.done[slot]);
                        }
                    }
                }
                j += nb;
                if (STAMP) { uint64_t now = (uint64_t)clock64(); ts_red += now - t_mark; t_mark = now; }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the resolver's tree / skip stores, before the next build reads them
            if (uni(lds_peek(&sh.stop_flag)) != 0) break;
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) {
            p.dbg[16 + wave] = ts_wait;
            p.dbg[24 + wave] = ts_pre + ts_scan + ts_red;
            if (wave == 5) { p.dbg[8] = ts_pre; p.dbg[9] = ts_scan; p.dbg[10] = ts_red; p.dbg[11] = ts_rows; p.dbg[5] = n_sort; p.dbg[6] = t_sort; }
        }
    } else {
        // ---- resolver state
        ProblemState st = st0;
        double goal_c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
        const double goal_thr = p.goal_thr[prob];
        const uint32_t nobs = p.n_spheres + p.n_boxes;
        const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
        const bool extras = nobs > ns64;
        double oc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) oc[k] = lane < ns64 ? p.sph_c[(size_t)k * p.n_spheres + lane] : 0.0;
        const double othr = lane < ns64 ? p.sph_thr[lane] : -1.0;
        const double ofilt = lane < ns64 ? p.sph_filt[lane] : -1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) sh.obs[k][lane] = oc[k];
        sh.obs[D][lane] = ofilt;
        RngWindow rng;
        rng.init(sh.rng_buf, p.seed, p.first_problem_id + prob, st.draws);
        uint64_t draws_done = st.draws;
        uint32_t n = st.n_nodes, js = 0, jr = 0;
        uint32_t n_base = st.n_nodes;       // tree size at the last (re)build
        int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
        const uint32_t row = lane >> 4, sub = lane & 15;
        uint64_t t_wait = 0, t_work = 0, t_samp = 0, t_comb = 0, n_amb = 0, t_mark = 0;
        // one query, the reference's sequential semantics in full (tails, batch conflicts, near-ties)
        auto resolve_one = [&](uint32_t jq, uint32_t& nearest, double (&q_new)[D], bool& dup) -> bool {
            const uint32_t slot = jq & (kRing - 1);
            double q[D];
    #pragma unroll
            for (int k = 0; k < D; ++k) q[k] = unid(sh.qring[slot].q[k]);
            const bool inS = lane < (uint32_t)kScanWaves;
            const WavePub<DIM>& mine = sh.pub[slot][inS ? lane : 0];
            const double pb = inS ? mine.b1 : __builtin_inf();
            const uint32_t pan = inS ? mine.amb_nc : 0xFFFFFFFFu;
            const uint32_t pamb = inS ? (pan & 1u) : 0u;
            const uint32_t pidxS = inS ? mine.i1 : kNoNode;
            const uint32_t base_min = wave_min_u32(pan >> 1);
            uint32_t pidx = kNoNode;
            if (n > lane) {
                const uint32_t i = lane + (((n - 1u - lane) >> 6) << 6);
                if (i >= base_min) pidx = i;
            }
            double pn[D];
    #pragma unroll
            for (int k = 0; k < D; ++k) pn[k] = sh.newn[lane][k];
            const bool pv = pidx != kNoNode;
            const double d2p = pv ? dist2<D>(pn, q, DIM) : __builtin_inf();
            const double g = wave_min_f64(d2p < pb ? d2p : pb);
            const uint32_t hb = hi32(g) + 1;
            const bool nearS = inS && hi32(pb) <= hb;
            const bool nearP = pv && hi32(d2p) <= hb;
            const uint64_t mS = __ballot(nearS), mP = __ballot(nearP);
            const bool from_scan = mS != 0;
            const int wl = from_scan ? (__ffsll((unsigned long long)mS) - 1) : (mP ? (__ffsll((unsigned long long)mP) - 1) : 0);
            nearest = from_scan ? (uint32_t)__builtin_amdgcn_readlane((int)pidxS, wl)
                                : (uint32_t)__builtin_amdgcn_readlane((int)pidx, wl);
            const bool amb = __ballot((nearS && (pidxS != nearest || pamb != 0)) || (nearP && pidx != nearest)) != 0;
            double q_near[D];
            double dist_or_g;
            if (!amb) {
    #pragma unroll
                for (int k = 0; k < D; ++k)
                    q_near[k] = from_scan ? unid(node_coord_p<DIM>(sh, tree, cap, st0.n_nodes, n, k, nearest)) : unid(sh.newn[wl][k]);
                dist_or_g = g;
                dup = g == 0.0;
            } else {
                if (STAMP) ++n_amb;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                Exact e{__builtin_inf(), kNoNode};
                for (uint32_t i = lane; i < n; i += 64) {
                    double c[D];
    #pragma unroll
                    for (int k = 0; k < D; ++k)
                        c[k] = __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double d = sqrt(dist2<D>(c, q, DIM));
                    if (d < e.dist) { e.dist = d; e.idx = i; }
                }
                e = exact_wave_reduce(e);
                nearest = uni(e.idx);
    #pragma unroll
                for (int k = 0; k < D; ++k)
                    q_near[k] = unid(__hip_atomic_load(&tree[(size_t)k * cap + nearest], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                dist_or_g = unid(e.dist);
                dup = dist_or_g == 0.0;
            }
            steer<DIM>(p, amb, dist_or_g, q_near, q, q_new);
            if (nobs == 0) return true;
            double mid[D];
            lerp<DIM>(q_near, q_new, 0.5, mid, DIM);
            if (__ballot(sphere_maybe_hit<DIM>(oc, ofilt, mid)) == 0 && !extras) return true;
            return motion_lanes<DIM>(p, lane, q_near, q_new, oc, othr, ofilt, ns64);
        };


        for (;;) {
            for (int bq = 0; bq < kBuildBarriers; ++bq) __syncthreads();  // the scanners (re)build
            n_base = uni(lds_peek(&sh.committed));
            if (STAMP) t_mark = (uint64_t)clock64();
            // ============================================================= run until stop / re-sort
            __builtin_amdgcn_s_setprio(3);
            bool finished = false;
            uint32_t resort_at = kNoResort;
            while (!finished) {
                if (jr >= budget) { finished = true; break; }
                if (!p.freeze && n >= p.max_nodes) { stop = 2; finished = true; break; }
                if (resort_at != kNoResort && jr >= resort_at) break;  // ring drained: meet the scanners at the barrier
                // re-sort when the unsorted tail has grown to max(512, n_base / 8) nodes
                if (resort_at == kNoResort && !p.freeze && (n - n_base) >= (n_base / 8 > 512u ? n_base / 8 : 512u)) {
                    resort_at = js;
                    if (lane == 0) lds_post(&ss.resort_at, js);
                    if (jr >= resort_at) break;
                }
                // sample ahead (rrt.rs:177-184), unless a re-sort is pending
                if (resort_at == kNoResort && js < budget && js - jr <= (uint32_t)(kRing / 2)) {
                    uint32_t m = jr + kRing - js;
                    if (m > budget - js) m = budget - js;
                    const uint64_t need_hi = rng.pos + (uint64_t)m * (1 + D) + 64;
                    if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
                        rng.base_blk = uni64(rng.pos >> 3);
                        uint32_t o[16];
                        chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
                        for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
                    }
                    if (!sample_batch<DIM>(rng, p, goal_c, m, lane, sh.qring, js)) {
                        for (uint32_t b = 0; b < m; ++b) {
                            double qn[D];
                            sample_state<D, false>(rng, p, DIM, goal_c, qn);
                            QSlot<DIM>& qs = sh.qring[(js + b) & (kRing - 1)];
                            if (lane == 0) {
#pragma unroll
                                for (int k = 0; k < D; ++k) qs.q[k] = qn[k];
                                qs.pos_after = rng.pos;
                            }
                        }
                    }
                    js += m;
                    if (lane == 0) lds_post(&sh.sampled, js);
                }
                if (STAMP) { uint64_t now = (uint64_t)clock64(); t_samp += now - t_mark; t_mark = now; }

                uint32_t lim = budget < js ? budget : js;           // only sampled queries can be resolved
                if (resort_at != kNoResort && resort_at < lim) lim = resort_at;
                const uint32_t nbq = (lim - jr < (uint32_t)kBatch) ? (lim - jr) : (uint32_t)kBatch;
                uint32_t spins = 0;
                for (uint32_t b = 0; b < nbq; ++b) {
                    while (uni(lds_peek(&sh.done[(jr + b) & (kRing - 1)])) < (uint32_t)kScanWaves && spins <= kMaxSpins) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                    }
                }
                if (spins > kMaxSpins) { stop = 4; finished = true; break; }
                if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }

                // ---- row-parallel phase: DPP row r works on query jr + r against the tree of n0q nodes
                const uint32_t n0q = n;
                const bool active = row < nbq;
                const uint32_t slot_r = (jr + (active ? row : 0u)) & (kRing - 1);
                double q[D];
#pragma unroll
                for (int k = 0; k < D; ++k) q[k] = sh.qring[slot_r].q[k];
                const uint64_t pos_after_r = sh.qring[slot_r].pos_after;
                const bool inS = active && sub < (uint32_t)kScanWaves;
                const WavePub<DIM>& mine = sh.pub[slot_r][inS ? sub : 0];
                const double pb = inS ? mine.b1 : __builtin_inf();
                const uint32_t pan = inS ? mine.amb_nc : 0xFFFFFFFFu;
                const uint32_t pidxS = inS ? mine.i1 : kNoNode;
                const uint32_t bmin_r = row_min_u32(pan >> 1);
                if (__ballot(active && (n0q - bmin_r > 64u || bmin_r > n0q)) != 0) { stop = 4; finished = true; break; }
                Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};
                if (active) {
                    for (uint32_t i = bmin_r + sub; i < n0q; i += 16) {
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = sh.newn[i & 63][k];
                        scan_push(pd, dist2<D>(c, q, DIM), i);
                    }
                }
                const double g_r = row_min_f64(pd.b1 < pb ? pd.b1 : pb);
                const uint32_t hb = hi32(g_r) + 1;
                const bool nearS = inS && hi32(pb) <= hb;
                const bool nearP = active && pd.slot != kNoNode && hi32(pd.b1) <= hb;
                const uint32_t rowS = (uint32_t)(__ballot(nearS) >> (16 * row)) & 0xFFFFu;
                const uint32_t rowP = (uint32_t)(__ballot(nearP) >> (16 * row)) & 0xFFFFu;
                const bool from_scan = rowS != 0;
                const uint32_t wsub = from_scan ? (uint32_t)(__ffs((int)rowS) - 1) : (rowP ? (uint32_t)(__ffs((int)rowP) - 1) : 0u);
                const int src_lane = (int)(16 * row + wsub);
                const uint32_t wS = (uint32_t)__shfl((int)pidxS, src_lane, 64), wP = (uint32_t)__shfl((int)pd.slot, src_lane, 64);
                const uint32_t nearest_r = from_scan ? wS : wP;
                const bool amb_l = (nearS && (pidxS != nearest_r || (pan & 1u) != 0)) || (nearP && pd.slot != nearest_r) ||
                                   (active && pd.h2 <= hb);
                const bool amb_r = ((uint32_t)(__ballot(amb_l) >> (16 * row)) & 0xFFFFu) != 0 || (rowS == 0 && rowP == 0);
                double q_near[D], qn[D], mid[D];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    q_near[k] = from_scan ? node_coord_p<DIM>(sh, tree, cap, st0.n_nodes, n0q, k, active && nearest_r != kNoNode ? nearest_r : 0u)
                                          : sh.newn[nearest_r & 63][k];
                steer<DIM>(p, false, g_r, q_near, q, qn);
                lerp<DIM>(q_near, qn, 0.5, mid, DIM);
                bool maybe_l = false;
                if (nobs > 0) {
                    for (uint32_t o = sub; o < 64; o += 16) {
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = sh.obs[k][o];
                        maybe_l = maybe_l || sphere_maybe_hit<DIM>(c, sh.obs[D][o], mid);
                    }
                }
                const bool maybe_r = extras || (((uint32_t)(__ballot(maybe_l) >> (16 * row)) & 0xFFFFu) != 0);
                if (STAMP) { uint64_t now = (uint64_t)clock64(); t_comb += now - t_mark; t_mark = now; }

                // ---- sequential phase: commit in query order
                double cn[kBatch][D];
                bool cn_valid[kBatch];
#pragma unroll
                for (int b = 0; b < kBatch; ++b) cn_valid[b] = false;
                bool leave = false;
                uint32_t processed = 0;
#pragma unroll
                for (int b = 0; b < kBatch; ++b) {
                    if (!leave && (uint32_t)b < nbq) {
                        if (!p.freeze && n >= p.max_nodes) { stop = 2; leave = true; }
                    }
                    if (!leave && (uint32_t)b < nbq) {
                        const int l0 = 16 * b;
                        const uint32_t slot = (jr + (uint32_t)b) & (kRing - 1);
                        const double g_b = readlane_f64(g_r, l0);
                        double q_b[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) q_b[k] = readlane_f64(q[k], l0);
                        bool redo = __builtin_amdgcn_readlane(amb_r ? 1 : 0, l0) != 0;
#pragma unroll
                        for (int c = 0; c < b; ++c)
                            if (cn_valid[c] && hi32(dist2<D>(cn[c], q_b, DIM)) <= hi32(g_b) + 1) redo = true;
                        uint32_t nearest;
                        double q_new[D];
                        bool ok, dup;
                        if (redo) {
                            ok = resolve_one(jr + (uint32_t)b, nearest, q_new, dup);
                        } else {
                            nearest = (uint32_t)__builtin_amdgcn_readlane((int)nearest_r, l0);
#pragma unroll
                            for (int k = 0; k < D; ++k) q_new[k] = readlane_f64(qn[k], l0);
                            dup = g_b == 0.0;
                            ok = true;
                            if (nobs > 0 && __builtin_amdgcn_readlane(maybe_r ? 1 : 0, l0) != 0) {
                                double qnr[D];
#pragma unroll
                                for (int k = 0; k < D; ++k) qnr[k] = readlane_f64(q_near[k], l0);
                                ok = motion_lanes<DIM>(p, lane, qnr, q_new, oc, othr, ofilt, ns64);
                            }
                        }
                        st.checksum = uni64(chk_push(st.checksum, iter_digest<D>(nearest, q_new, DIM, ok)));
                        st.iterations++;
                        draws_done = uni64((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_r, l0) |
                                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_r >> 32), l0) << 32));
                        bool hit = false;
                        if (ok) {
                            st.accepted++;
                            if (!p.freeze) {
                                const uint32_t i = n;
#pragma unroll
                                for (int k = 0; k < D; ++k) cn[b][k] = dup ? __builtin_inf() : q_new[k];
                                cn_valid[b] = true;
                                if (lane == (i & 63)) {
#pragma unroll
                                    for (int k = 0; k < D; ++k) {
                                        sh.newn[i & 63][k] = cn[b][k];
                                        tree[(size_t)k * cap + i] = q_new[k];
                                    }
                                    parent[i] = (int32_t)nearest;
                                    skip[i] = dup ? 1 : 0;
                                }
                                ++n;
                                if (lane == 0) lds_post(&sh.committed, n);
                                if (dist2<D>(q_new, goal_c, DIM) <= goal_thr) {
                                    if (st.goal_node < 0) st.goal_node = (int32_t)i;
                                    hit = true;
                                }
                            }
                        }
                        if (lane == 0) {
                            lds_post(&sh.done[slot], 0);
                            lds_post(&sh.resolved, jr + (uint32_t)b + 1);
                        }
                        ++processed;
                        if (hit && p.stop_at_goal) { stop = 0; leave = true; }
                    }
                }
                jr += processed;
                if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
                if (leave) { finished = true; break; }
            }
            if (finished && lane == 0) lds_post(&sh.stop_flag, 1);
            if (!finished) {
                // the sorted copy is rebuilt from HBM: make this wave's tree / skip stores visible to every CU-local reader
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            }
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (uni(lds_peek(&sh.stop_flag)) != 0) break;
        }
        if (lane == 0) {
            st.n_nodes = n;
            st.draws = draws_done;
            st.stop_reason = stop;
            p.state[prob] = st;
            if (STAMP && p.dbg && prob == 0) {
                p.dbg[0] = t_samp; p.dbg[1] = t_wait; p.dbg[2] = t_work; p.dbg[3] = t_comb; p.dbg[4] = n_amb; p.dbg[7] = st.iterations;
            }
        }
    }
}

static int pruned_slots(uint32_t cap) {
    const uint32_t need = (cap + kScanThreads - 1) / kScanThreads;
    if (need <= 4) return 4;
    if (cap <= Layout<21>::kCapacity) return 21;
    return 0;
}

bool pruned_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pruned_slots(cap) != 0; }

void launch_rrt_pruned(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kPipeThreads);
    const int s = pruned_slots(p.cap);
#define OXHIP_LAUNCH(DIM_, S_)                                                                            \
    do {                                                                                                  \
        if (p.dbg) hipLaunchKernelGGL((rrt_pruned_kernel<DIM_, S_, true>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((rrt_pruned_kernel<DIM_, S_, false>), grid, block, 0, stream, p);         \
    } while (0)
    if (p.dim == 3) {
        if (s == 4) OXHIP_LAUNCH(3, 4); else OXHIP_LAUNCH(3, 21);
    } else {
        if (s == 4) OXHIP_LAUNCH(2, 4); else OXHIP_LAUNCH(2, 21);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
