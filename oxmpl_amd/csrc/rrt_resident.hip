// rrt_resident.hip -- register-resident RRT grow kernel for gfx950 (the fast path).
//
// A 10,240-node R^3 tree is 240 KB: too big for LDS (160 KB) but it fits the 512 KB vector
// register file of one CU.  One 576-thread workgroup per problem: 8 scanner waves hold the tree in
// VGPRs (126 VGPRs for R^3) and stream queries from an LDS ring, so the O(n) nearest-neighbour scan
// of rrt.rs:187-196 reads no memory at all and is bounded by f64 VALU issue; one resolver wave
// samples ahead, consumes the scanners' candidates in order, steers, checks the motion and commits.
// No workgroup barrier in steady state (see "Asynchronous pipeline" below).  The SoA tree in HBM
// is only the persistent copy: written once per insert, read once per launch.
//
// Replaces the loop body of RRT::solve, oxmpl/src/geometric/planners/rrt.rs:170-225.
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"
#include "rrt_resident_common.hpp"

namespace oxhip {
// Coordinates of node i for the resolver.  The scanners publish only (d2, index): copying the winning slot out of a
// register array costs every scanner wave a 21-way uniform branch ladder per query (~180 scalar instructions), and
// seven of eight waves lose anyway (167.7 -> 189.4 M it/s without it).  The resolver fetches the one node it needs:
// from the LDS ring of the last 64 commits when the node is that young (its HBM store may still be in flight),
// else from the persistent copy in HBM / L2, which this same wave wrote at least 64 commits -- and several
// `s_waitcnt vmcnt(0)` -- ago, or which predates the launch.
template <int DIM>
__device__ __forceinline__ double node_coord(const PipeShared<DIM>& sh, const double* tree, size_t cap, uint32_t n_start,
                                             uint32_t n_now, int k, uint32_t i) {
    if (i >= n_start && i + 64u >= n_now) return sh.newn[i & 63][k];
    return __hip_atomic_load(&tree[(size_t)k * cap + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DIM, int S, bool STAMP>
__global__ __launch_bounds__(kPipeThreads) void rrt_resident_kernel(DevParams p) {
    constexpr int D = DIM;
    const uint32_t prob = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = uni(tid >> 6), lane = tid & 63;

    __shared__ PipeShared<DIM> sh;

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
    }
    __syncthreads();  // the only workgroup barrier of the launch

    if (wave < kScanWaves) {
        // ================================================================= scanner waves
        using Lay = Layout<S>;
        uint32_t n_local = st0.n_nodes;
        double tr[DIM][S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t i = ((uint32_t)s < Lay::kCommon || Lay::heavy(wave)) ? Lay::node_index(wave, lane, (uint32_t)s) : kNoNode;
            const bool live = i < n_local && skip[i] == 0;  // duplicates of a lower-index node never win: hold +inf
#pragma unroll
            for (int k = 0; k < DIM; ++k) tr[k][s] = live ? tree[(size_t)k * cap + i] : __builtin_inf();
        }
        uint64_t t_wait = 0, t_work = 0, t_pre = 0, t_scan = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;
        uint32_t seen_sampled = 0;
        for (uint32_t j = 0; j < budget; j += kBatch) {
            // one pass scans kBatch queries (the tail pass may hold one): fixed costs are shared, the two
            // reductions are independent dependency chains
            const uint32_t nb = (budget - j < (uint32_t)kBatch) ? (budget - j) : (uint32_t)kBatch;
            const uint32_t need = j + nb;
            // wait until the pass's queries are sampled (implies their ring slots were consumed kRing queries ago)
            for (uint32_t spins = 0; seen_sampled < need; ++spins) {
                if (lds_peek(&sh.stop_flag) != 0 || spins > kMaxSpins) break;  // every spin is bounded
                seen_sampled = uni(lds_peek(&sh.sampled));
                if (seen_sampled < need) __builtin_amdgcn_s_sleep(2);
            }
            if (seen_sampled < need) break;  // stop requested
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }
            // absorb the nodes committed since this wave's last snapshot (the owner lane takes each)
            const uint32_t nc = uni(lds_peek(&sh.committed));
            for (uint32_t i = n_local; i < nc; ++i) {
                uint32_t owner_thread, sl;
                Lay::locate(i, owner_thread, sl);
                if ((owner_thread >> 6) != wave) continue;   // another wave's node: skip the slot ladder below
                const bool mine = tid == owner_thread;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (sl == (uint32_t)s) {
                        if (mine) {
#pragma unroll
                            for (int k = 0; k < D; ++k) tr[k][s] = sh.newn[i & 63][k];
                        }
                    }
                }
            }
            n_local = nc;
            double q[kBatch][D];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const uint32_t slot = (j + ((uint32_t)b < nb ? (uint32_t)b : 0u)) & (kRing - 1);
#pragma unroll
                for (int k = 0; k < D; ++k) q[b][k] = unid(sh.qring[slot].q[k]);
            }
            // nearest neighbour over this wave's nodes (rrt.rs:187-196), d2 compare
            const uint32_t nslots = Lay::slots_in_use(wave, nc);
            Scan sc[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) sc[b] = Scan{__builtin_inf(), 0u, 0xFFFFFFFFu};
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_pre += now - t_mark; t_mark = now; }
            // slots are visited in groups of kGroup under ONE uniform branch: inside a group the code is
            // straight-line, so the scheduler interleaves kGroup x kBatch independent sub/mul/add chains
            // (empty slots hold +inf and can never win)
#pragma unroll
            for (int g0 = 0; g0 < S; g0 += group_len<S>(g0)) {
                if ((uint32_t)g0 < nslots) {
#pragma unroll
                    for (int s = g0; s < g0 + group_len<S>(g0); ++s) {
                        double c[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) c[k] = tr[k][s];
#pragma unroll
                        for (int b = 0; b < kBatch; ++b) scan_push(sc[b], dist2<D>(c, q[b], DIM), (uint32_t)s);
                    }
                }
            }
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_scan += now - t_mark; t_mark = now; }
            // reduce.  d2 >= +0, so its high dword orders like the value: the wave minimum of the HIGH DWORDS (a 32-bit
            // DPP chain, half the work of the 64-bit one) names the winner outright whenever exactly one lane's best
            // lies within one high-dword step of it and no lane saw a second value that close -- which is also
            // precisely the "unambiguous" verdict.  Only otherwise (near-ties, an all-empty wave: ~1e-3 of passes)
            // the full 64-bit minimum and tie analysis run.  The four chains of a pass interleave; one branch per pass.
            double wmin[kBatch];
            int wl[kBatch];
            uint32_t wslot[kBatch], wamb[kBatch];
            uint32_t mh[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) mh[b] = wave_min_u32(hi32(sc[b].b1));
            uint64_t nearA[kBatch];
            bool fast = true;
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                nearA[b] = __ballot(hi32(sc[b].b1) <= mh[b] + 1u);
                const uint64_t nearB = __ballot(sc[b].h2 <= mh[b] + 1u);
                fast = fast && __popcll(nearA[b]) == 1 && nearB == 0;
            }
            if (fast) {
#pragma unroll
                for (int b = 0; b < kBatch; ++b) {
                    wl[b] = __ffsll((unsigned long long)nearA[b]) - 1;
                    wmin[b] = readlane_f64(sc[b].b1, wl[b]);
                    wslot[b] = __builtin_amdgcn_readlane(sc[b].slot, wl[b]);
                    wamb[b] = 0u;
                }
            } else {
#pragma unroll
                for (int b = 0; b < kBatch; ++b) wmin[b] = wave_min_f64(sc[b].b1);
#pragma unroll
                for (int b = 0; b < kBatch; ++b) {
                    const uint64_t eqm = __ballot(sc[b].b1 == wmin[b]);
                    wl[b] = eqm ? (__ffsll((unsigned long long)eqm) - 1) : 0;
                    wslot[b] = __builtin_amdgcn_readlane(sc[b].slot, wl[b]);
                    const uint32_t hb = hi32(wmin[b]) + 1;
                    const bool amb_l = ((int)lane != wl[b] && hi32(sc[b].b1) <= hb) || (sc[b].h2 <= hb);
                    wamb[b] = __ballot(amb_l) != 0 ? 1u : 0u;
                }
            }
            // publish: the owning lane stores the coordinates, lane 0 the 16-byte head, then the slot is counted
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                if ((uint32_t)b < nb) {
                    const uint32_t slot = (j + (uint32_t)b) & (kRing - 1);
                    WavePub<DIM>& out = sh.pub[slot][wave];
                    // (the candidate's coordinates are not published: see node_coord)
                    if (lane == 0) {
                        out.b1 = wmin[b];
                        out.i1 = Lay::node_index(wave, (uint32_t)wl[b], wslot[b]);
                        out.amb_nc = (nc << 1) | wamb[b];
                        lds_bump(&sh.done[slot]);
                    }
                }
            }
            if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        }
        if (STAMP && p.dbg && prob == 0 && lane == 0) {
            p.dbg[16 + wave] = t_wait;
            p.dbg[24 + wave] = t_work + t_pre + t_scan;
            if (wave == 5) { p.dbg[8] = t_pre; p.dbg[9] = t_scan; p.dbg[10] = t_work; }
        }
        return;
    }

    // ===================================================================== resolver wave
    __builtin_amdgcn_s_setprio(3);  // the youngest wave of its SIMD would otherwise queue behind two scanners
    ProblemState st = st0;
    double goal_c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) goal_c[k] = p.goal_c[(size_t)prob * DIM + k];
    const double goal_thr = p.goal_thr[prob];
    const uint32_t nobs = p.n_spheres + p.n_boxes;
    const uint32_t ns64 = p.n_spheres < 64 ? p.n_spheres : 64;
    const bool extras = nobs > ns64;  // spheres beyond the first 64 and every box: always stepped, never filtered
    // this lane's obstacle (unused lanes hold a sphere that can never be hit), also mirrored to LDS
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
    uint32_t n = st.n_nodes;
    uint32_t js = 0, jr = 0;
    int32_t stop = 1;  // OXHIP_STOP_ITERATIONS
    const uint32_t row = lane >> 4, sub = lane & 15;
    uint64_t t_wait = 0, t_work = 0, t_samp = 0, t_comb = 0, n_amb = 0, t_mark = STAMP ? (uint64_t)clock64() : 0;

    // ---- one query, the reference's sequential semantics in full (tails, batch conflicts, near-ties):
    //      candidates = the 8 scanner waves' + every node committed after the oldest scan snapshot
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
        const uint32_t base_min = wave_min_u32(pan >> 1);  // oldest snapshot among the 8 scans
        // the ring entry of this lane holds the latest node i with (i & 63) == lane
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
        // unambiguous iff every near candidate is that one node and no near wave saw a second near node
        const bool amb = __ballot((nearS && (pidxS != nearest || pamb != 0)) || (nearP && pidx != nearest)) != 0;
        double q_near[D];
        double dist_or_g;
        if (!amb) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                q_near[k] = from_scan ? unid(node_coord<DIM>(sh, tree, cap, st0.n_nodes, n, k, nearest)) : unid(sh.newn[wl][k]);
            dist_or_g = g;
            dup = g == 0.0;
        } else {
            // rare (~1e-6 of queries): the reference's own loop -- post-sqrt compare with lowest-index ties --
            // over the persistent copy of the tree in global memory
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

    while (jr < budget) {
        if (!p.freeze && n >= p.max_nodes) { stop = 2; break; }
        // sample ahead (rrt.rs:177-184); a query may only reuse a ring slot after its previous tenant was resolved
        if (js < budget && js - jr <= (uint32_t)(kRing / 2)) {
            uint32_t m = jr + kRing - js;  // free ring slots
            if (m > budget - js) m = budget - js;
            // keep the batch's words inside the LDS window: refill (64 blocks from the current position) when short
            const uint64_t need_hi = rng.pos + (uint64_t)m * (1 + D) + 64;
            if ((rng.pos >> 3) - rng.base_blk >= 64 || need_hi > (rng.base_blk + 64) * 8) {
                rng.base_blk = uni64(rng.pos >> 3);
                uint32_t o[16];
                chacha12_block(rng.seed, rng.base_blk + lane, rng.stream, o);
#pragma unroll
                for (int w = 0; w < 16; ++w) rng.buf[w][lane] = o[w];
            }
            if (!sample_batch<DIM>(rng, p, goal_c, m, lane, sh.qring, js)) {
                for (uint32_t b = 0; b < m; ++b) {  // (never expected) a redraw ran past the window: one by one
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

        // the scanners publish kBatch queries per pass: resolve them as one batch
        const uint32_t nbq = (budget - jr < (uint32_t)kBatch) ? (budget - jr) : (uint32_t)kBatch;
        uint32_t spins = 0;
        for (uint32_t b = 0; b < nbq; ++b) {
            while (uni(lds_peek(&sh.done[(jr + b) & (kRing - 1)])) < (uint32_t)kScanWaves && spins <= kMaxSpins) {
                __builtin_amdgcn_s_sleep(1);
                ++spins;
            }
        }
        if (spins > kMaxSpins) { stop = 4; break; }  // OXHIP_STOP_INTERNAL: a scanner never published (bug guard)
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_wait += now - t_mark; t_mark = now; }

        // ---- row-parallel phase: DPP row r (16 lanes) works on query jr + r against the tree of n0 nodes
        const uint32_t n0 = n;
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
        const uint32_t bmin_r = row_min_u32(pan >> 1);  // oldest scan snapshot for this row's query
        if (__ballot(active && (n0 - bmin_r > 64u || bmin_r > n0)) != 0) { stop = 4; break; }  // ring would have wrapped (bug guard)
        // nodes committed after that snapshot (at most a few): lanes of the row stride over them
        Scan pd{__builtin_inf(), kNoNode, 0xFFFFFFFFu};  // .slot is used as the node index here
        if (active) {
            for (uint32_t i = bmin_r + sub; i < n0; i += 16) {
                double c[D];
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = sh.newn[i & 63][k];
                scan_push(pd, dist2<D>(c, q, DIM), i);  // ascending i: ties keep the lower index
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
        // ambiguous iff a second node is near: another near candidate, a near wave that saw a second near node,
        // or a lane whose second-best recent node is near too
        const bool amb_l = (nearS && (pidxS != nearest_r || (pan & 1u) != 0)) || (nearP && pd.slot != nearest_r) ||
                           (active && pd.h2 <= hb);
        const bool amb_r = ((uint32_t)(__ballot(amb_l) >> (16 * row)) & 0xFFFFu) != 0 || (rowS == 0 && rowP == 0);
        double q_near[D], qn[D], mid[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            q_near[k] = from_scan ? node_coord<DIM>(sh, tree, cap, st0.n_nodes, n0, k, active && nearest_r != kNoNode ? nearest_r : 0u)
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

        // ---- sequential phase: commit in query order; a node committed earlier in this batch that is
        //      closer (or near-tied) to a later query forces that query through resolve_one
        double cn[kBatch][D];   // coordinates the scanners will hold for the nodes committed in this batch
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
                // bookkeeping (wave-uniform, on the scalar unit where the compiler can)
                st.checksum = uni64(chk_push(st.checksum, iter_digest<D>(nearest, q_new, DIM, ok)));
                st.iterations++;
                draws_done = uni64((uint64_t)__builtin_amdgcn_readlane((int)(uint32_t)pos_after_r, l0) |
                                   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos_after_r >> 32), l0) << 32));
                bool hit = false;
                if (ok) {
                    st.accepted++;
                    if (!p.freeze) {
                        // 6. insert (rrt.rs:213-217): LDS hand-off to the owning scanner lane + HBM copy.  A node at
                        // distance 0 from its nearest node repeats that node's coordinates, and the strict '<' of
                        // rrt.rs:192 can never prefer it over the lower index: the scanners keep +inf for it.
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
                        // 7. goal test (rrt.rs:220-223)
                        if (dist2<D>(q_new, goal_c, DIM) <= goal_thr) {
                            if (st.goal_node < 0) st.goal_node = (int32_t)i;
                            hit = true;
                        }
                    }
                }
                if (lane == 0) {
                    lds_post(&sh.done[slot], 0);                 // free the slot ...
                    lds_post(&sh.resolved, jr + (uint32_t)b + 1); // ... before the sampler may hand it out again
                }
                ++processed;
                if (hit && p.stop_at_goal) { stop = 0; leave = true; }
            }
        }
        jr += processed;
        if (STAMP) { uint64_t now = (uint64_t)clock64(); t_work += now - t_mark; t_mark = now; }
        if (leave) break;
    }
    if (lane == 0) {
        lds_post(&sh.stop_flag, 1);
        st.n_nodes = n;
        st.draws = draws_done;
        st.stop_reason = stop;
        p.state[prob] = st;
        if (STAMP && p.dbg && prob == 0) {
            p.dbg[0] = t_samp; p.dbg[1] = t_wait; p.dbg[2] = t_work; p.dbg[3] = t_comb; p.dbg[4] = n_amb; p.dbg[7] = st.iterations;
        }
    }
}

// instantiations: (dim, slots) -> capacity 512 * slots nodes
static int pick_slots(uint32_t cap) {
    const uint32_t need = (cap + kScanThreads - 1) / kScanThreads;
    if (need <= 4) return 4;
    if (cap <= Layout<21>::kCapacity) return 21;
    return 0;
}

bool resident_supported(uint32_t dim, uint32_t cap) { return (dim == 2 || dim == 3) && pick_slots(cap) != 0; }

void launch_rrt_resident(const DevParams& p, hipStream_t stream) {
    dim3 grid(p.n_problems), block(kPipeThreads);
    const int s = pick_slots(p.cap);
#define OXHIP_LAUNCH(DIM_, S_)                                                                              \
    do {                                                                                                    \
        if (p.dbg) hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, true>), grid, block, 0, stream, p);    \
        else hipLaunchKernelGGL((rrt_resident_kernel<DIM_, S_, false>), grid, block, 0, stream, p);         \
    } while (0)
    if (p.dim == 3) {
        if (s == 4) OXHIP_LAUNCH(3, 4); else OXHIP_LAUNCH(3, 21);
    } else {
        if (s == 4) OXHIP_LAUNCH(2, 4); else OXHIP_LAUNCH(2, 21);
    }
#undef OXHIP_LAUNCH
}

}  // namespace oxhip
