// oxhip_prm_api.hip -- the PRM part of the C ABI (include/oxmpl_hip.h, oxhip_prm_*).
//
// Host side of oxmpl's PRM (oxmpl/src/geometric/planners/prm.rs): owns the device roadmap, drives
// the construction phases of prm_kernels.hip, and runs the breadth-first query (prm.rs:270-307) and
// reconstruct_path (prm.rs:189-208) on the CSR roadmap it copies back once per construction.  Start
// validity, start connections and goal milestones (prm.rs:243-264) are computed on the device.
// There is no CPU fallback: without a HIP device every computing entry point fails.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <deque>

#include "oxhip_host.hpp"
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

using namespace oxhip;

struct oxhip_prm {
    oxhip_prm_config cfg{};
    DevParams dp{};           // space, resolution and validity field (the RRT fields stay zero)
    PrmArgs args{};
    double thr_conn = -1.0;   // d2 <= thr_conn  <=>  distance < connection_radius
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    DevBuf<double> ms, sph_c, sph_thr, sph_r, box_lo, box_hi;
    DevBuf<float> ms32;   // fl32 shadow of the milestones (pair-search screen)
    double maxabs = 1.0;      // largest |coordinate| of bounds and sphere centres (midpoint filter margin)
    DevBuf<PrmState> state;
    DevBuf<uint2> cand;
    // k-nearest variant: per-row candidate radii, the sorted candidates and their distances, the selected pairs
    DevBuf<double> knn_thr, knn_dist;
    DevBuf<float> knn_thr32;
    DevBuf<uint64_t> knn_keys, knn_sorted;
    DevBuf<uint2> knn_sel;
    DevBuf<uint32_t> knn_counters, knn_failed;
    uint32_t knn_failed_rows = 0;   // rows of the last construct_roadmap that needed the exact search
    DevBuf<uint64_t> keys, keys_sorted;
    DevBuf<uint8_t> sort_tmp;
    uint32_t n_keys = 0;       // directed edge entries of the constructed roadmap
    bool host_copy = false;    // h_offsets / h_nbrs / h_states are current
    DevBuf<uint32_t> offsets, nbrs, start_valid;
    DevBuf<uint8_t> flags;
    // parallel sampler scratch (one round)
    DevBuf<double> spec_tmp;
    DevBuf<uint64_t> spec_vbits;
    DevBuf<uint32_t> spec_off, spec_flag;
    double valid_rate = 1.0;   // running estimate of P(sample is valid), sizes the rounds
    bool is_setup = false;
    PrmQuery query{};
    // host copy of the constructed roadmap
    uint32_t n = 0;
    uint64_t n_samples = 0;
    uint32_t redraw_batches = 0;
    std::vector<uint32_t> h_offsets, h_nbrs;
    std::vector<double> h_states;  // AoS [n][dim]
    // last query
    std::vector<uint32_t> start_conn, goal_idx;
    // phase timings of the last construct / solve (ms): sample, pairs, edges, sort+csr, query kernel, bfs (host)
    double t_ms[6] = {};
    uint64_t n_candidates = 0;
};

namespace {

int32_t read_state(oxhip_prm* h, PrmState& st) {
    HIP_TRY(hipMemcpyAsync(&st, h->state.p, sizeof(PrmState), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return OXHIP_OK;
}

int32_t write_state(oxhip_prm* h, const PrmState& st) {
    HIP_TRY(hipMemcpyAsync(h->state.p, &st, sizeof(PrmState), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return OXHIP_OK;
}

int32_t set_query(oxhip_prm* h, const double* start, const double* goal_centre, double goal_radius) {
    const uint32_t dim = h->cfg.dim;
    for (uint32_t k = 0; k < dim; ++k)
        if (!(std::fabs(start[k]) <= kMaxMagnitude) || !(std::fabs(goal_centre[k]) <= kMaxMagnitude))
            return fail(OXHIP_ERR_BAD_ARG, "start / goal centre not finite or beyond 1e150");
    h->query = PrmQuery{};
    std::memcpy(h->query.start, start, dim * sizeof(double));
    std::memcpy(h->query.goal_c, goal_centre, dim * sizeof(double));
    h->query.goal_thr = sqrt_le_threshold(goal_radius);
    return OXHIP_OK;
}

void clear_roadmap(oxhip_prm* h) {
    h->n = 0;
    h->n_keys = 0;
    h->host_copy = false;
    h->n_samples = 0;
    h->redraw_batches = 0;
    h->h_offsets.clear();
    h->h_nbrs.clear();
    h->h_states.clear();
    h->start_conn.clear();
    h->goal_idx.clear();
}

// Draw samples until `target` milestones exist or `max_samples` were drawn.  Rounds of the parallel sampler
// sized by the observed validity rate; a round in which rand's range sampler would have rejected a draw is
// replayed by the sequential kernel (exact for any bounds, and about as likely as a 2^-52 event).
int32_t sample_until(oxhip_prm* h, PrmState& st, uint32_t target, uint64_t max_samples) {
    constexpr uint64_t kRoundMax = 1ull << 22;
    const uint32_t dim = h->cfg.dim;
    h->args.n_target = target;
    while (st.n_milestones < target && st.n_samples < max_samples) {
        const uint32_t need = target - st.n_milestones;
        uint64_t want = (uint64_t)((double)need / h->valid_rate * 1.02) + 256;
        want = std::min(want, std::min(max_samples - st.n_samples, kRoundMax));
        const uint32_t m = (uint32_t)want, nw = (m + 63) / 64;
        if (h->spec_tmp.n < (size_t)m * dim) HIP_TRY(h->spec_tmp.alloc((size_t)m * dim));
        if (h->spec_vbits.n < nw) { HIP_TRY(h->spec_vbits.alloc(nw)); HIP_TRY(h->spec_off.alloc(nw)); }
        if (h->spec_flag.n == 0) HIP_TRY(h->spec_flag.alloc(1));
        HIP_TRY(hipMemsetAsync(h->spec_flag.p, 0, sizeof(uint32_t), h->stream));
        PrmSpec sp{};
        sp.pos0 = st.draws; sp.m = m;
        sp.tmp = h->spec_tmp.p; sp.vbits = h->spec_vbits.p; sp.wave_off = h->spec_off.p;
        sp.redraw_flag = h->spec_flag.p; sp.result = h->state.p;   // the scan kernel advances the device state
        launch_prm_sample_spec(h->dp, h->args, sp, st.n_milestones, h->stream);
        HIP_TRY(hipGetLastError());
        PrmState after{};
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&after, h->state.p, sizeof(PrmState), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(&flag, h->spec_flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (flag == 0) {
            const double drawn = (double)(after.n_samples - st.n_samples), got = (double)(after.n_milestones - st.n_milestones);
            h->valid_rate = std::max(1.0 / 64.0, std::min(1.0, (got + 1.0) / (drawn + 1.0)));
            st = after;
        } else {
            // a draw was rejected somewhere in the round: put the state back and replay this window in order
            OX_TRY(write_state(h, st));
            h->args.max_samples = st.n_samples + m;
            launch_prm_sample(h->dp, h->args, h->stream);
            HIP_TRY(hipGetLastError());
            OX_TRY(read_state(h, st));
        }
    }
    return OXHIP_OK;
}

// host copy of the roadmap for get_roadmap and the breadth-first query (once per construction)
int32_t fetch_roadmap(oxhip_prm* h) {
    if (h->host_copy) return OXHIP_OK;
    const uint32_t n = h->n, n_keys = h->n_keys;
    h->h_offsets.assign((size_t)n + 1, 0);
    h->h_nbrs.assign(n_keys, 0);
    h->h_states.assign((size_t)n * h->cfg.dim, 0.0);
    if (n) {
        HIP_TRY(hipMemcpyAsync(h->h_offsets.data(), h->offsets.p, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        if (n_keys) HIP_TRY(hipMemcpyAsync(h->h_nbrs.data(), h->nbrs.p, (size_t)n_keys * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(h->h_states.data(), h->ms.p, (size_t)n * h->cfg.dim * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->host_copy = true;
    return OXHIP_OK;
}

double elapsed_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    return hipEventElapsedTime(&ms, a, b) == hipSuccess ? (double)ms : 0.0;
}

}  // namespace

extern "C" {

int32_t oxhip_prm_create(const oxhip_prm_config* cfg, oxhip_prm** out) {
    if (!cfg || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(oxhip_prm_config)) return fail(OXHIP_ERR_BAD_ARG, "struct_size mismatch");
    if (cfg->max_milestones == 0 || cfg->max_milestones > (1u << 26))
        return fail(OXHIP_ERR_BAD_ARG, "max_milestones must be in 1..2^26");
    if (std::isnan(cfg->connection_radius)) return fail(OXHIP_ERR_BAD_ARG, "connection_radius is NaN");
    double fraction = cfg->lvs_fraction, res = 0.0;
    OX_TRY(space_resolution(cfg->dim, cfg->bounds, fraction, res));
    // the longest motion PRM ever checks is shorter than the connection radius
    if (std::isfinite(cfg->connection_radius) && cfg->connection_radius / res > 1e6)
        return fail(OXHIP_ERR_BAD_ARG, "more than 1e6 validity checks per edge");
    OX_TRY(select_device(cfg->device));

    auto* h = new oxhip_prm();
    h->cfg = *cfg;
    h->cfg.lvs_fraction = fraction;
    const uint32_t dim = cfg->dim;
    const uint32_t cap = ((cfg->max_milestones + 1023u) / 1024u) * 1024u;
    DevParams& dp = h->dp;
    dp.dim = dim;
    for (uint32_t k = 0; k < dim; ++k) {
        dp.lo[k] = cfg->bounds[2 * k];
        dp.hi[k] = cfg->bounds[2 * k + 1];
        dp.scale[k] = dp.hi[k] - dp.lo[k];
    }
    dp.res = res;
    dp.seed = cfg->seed;
    for (uint32_t k = 0; k < 2 * dim; ++k) h->maxabs = std::fmax(h->maxabs, std::fabs(cfg->bounds[k]));
    dp.filt_abs = 1e-9 * h->maxabs;
    h->thr_conn = sqrt_lt_threshold(cfg->connection_radius);

    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    chk(oxhip_stream_acquire(cfg->device, &h->stream));
    for (auto& ev : h->ev) chk(hipEventCreate(&ev));
    chk(h->ms.alloc((size_t)dim * cap));
    chk(h->ms32.alloc((size_t)dim * cap));
    chk(h->state.alloc(1));
    chk(h->flags.alloc(cap));
    chk(h->start_valid.alloc(1));
    chk(h->offsets.alloc((size_t)cap + 1));
    if (e != hipSuccess) {
        std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e);
        oxhip_prm_destroy(h);
        return fail(OXHIP_ERR_HIP, msg);
    }
    h->args.ms = h->ms.p;
    h->args.ms32 = h->ms32.p;
    h->args.cap = cap;
    h->args.stream = cfg->stream;
    h->args.state = h->state.p;
    *out = h;
    return OXHIP_OK;
}

int32_t oxhip_prm_destroy(oxhip_prm* h) {
    if (!h) return OXHIP_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) oxhip_stream_release(h->cfg.device, h->stream);
    delete h;
    return OXHIP_OK;
}

int32_t oxhip_prm_set_spheres(oxhip_prm* h, const double* centres, const double* radii, uint32_t n) {
    if (!h || (n && (!centres || !radii))) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    OX_TRY(select_device(h->cfg.device));
    const uint32_t dim = h->cfg.dim;
    std::vector<double> c((size_t)dim * n), thr(n);
    for (uint32_t j = 0; j < n; ++j) {
        for (uint32_t k = 0; k < dim; ++k) {
            double v = centres[(size_t)j * dim + k];
            if (!(std::fabs(v) <= kMaxMagnitude)) return fail(OXHIP_ERR_BAD_ARG, "sphere centre not finite / too large");
            c[(size_t)k * n + j] = v;  // SoA [dim][n]
        }
        thr[j] = sqrt_le_threshold(radii[j]);
    }
    OX_TRY(upload(h->sph_c, c, h->stream));
    OX_TRY(upload(h->sph_thr, thr, h->stream));
    OX_TRY(upload(h->sph_r, std::vector<double>(radii, radii + n), h->stream));
    h->dp.n_spheres = n; h->dp.sph_c = h->sph_c.p; h->dp.sph_thr = h->sph_thr.p; h->dp.sph_r = h->sph_r.p;
    h->maxabs = 1.0;
    for (uint32_t k = 0; k < 2 * dim; ++k) h->maxabs = std::fmax(h->maxabs, std::fabs(h->cfg.bounds[k]));
    for (double v : c) h->maxabs = std::fmax(h->maxabs, std::fabs(v));
    h->dp.filt_abs = 1e-9 * h->maxabs;
    return OXHIP_OK;
}

int32_t oxhip_prm_set_boxes(oxhip_prm* h, const double* lo, const double* hi, uint32_t n) {
    if (!h || (n && (!lo || !hi))) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    OX_TRY(select_device(h->cfg.device));
    const uint32_t dim = h->cfg.dim;
    std::vector<double> l((size_t)dim * n), u((size_t)dim * n);
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t k = 0; k < dim; ++k) {
            l[(size_t)k * n + j] = lo[(size_t)j * dim + k];
            u[(size_t)k * n + j] = hi[(size_t)j * dim + k];
        }
    OX_TRY(upload(h->box_lo, l, h->stream));
    OX_TRY(upload(h->box_hi, u, h->stream));
    h->dp.n_boxes = n; h->dp.box_lo = h->box_lo.p; h->dp.box_hi = h->box_hi.p;
    return OXHIP_OK;
}

// Planner::setup (prm.rs:217-225)
int32_t oxhip_prm_setup(oxhip_prm* h, const double* start, const double* goal_centre, double goal_radius) {
    if (!h || !start || !goal_centre) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    OX_TRY(select_device(h->cfg.device));
    OX_TRY(set_query(h, start, goal_centre, goal_radius));
    clear_roadmap(h);  // self.roadmap.clear()
    PrmState st{};
    OX_TRY(write_state(h, st));
    h->is_setup = true;
    return OXHIP_OK;
}

// set_problem_definition (prm.rs:88-90): the roadmap is kept
int32_t oxhip_prm_set_problem(oxhip_prm* h, const double* start, const double* goal_centre, double goal_radius) {
    if (!h || !start || !goal_centre) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    return set_query(h, start, goal_centre, goal_radius);
}

// construct_roadmap (prm.rs:96-154)
int32_t oxhip_prm_construct_roadmap(oxhip_prm* h) {
    if (!h) return fail(OXHIP_ERR_BAD_ARG, "null handle");
    if (!h->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");  // prm.rs:97-104
    if (h->n != 0) return OXHIP_OK;  // prm.rs:106-113: "Roadmap already constructed"
    OX_TRY(select_device(h->cfg.device));
    const auto t0 = std::chrono::steady_clock::now();
    const bool has_timeout = h->cfg.timeout > 0.0 && std::isfinite(h->cfg.timeout);
    const uint32_t n_max = h->cfg.max_milestones;
    // 0 = "unlimited", which still has to end when (almost) no sample is valid: 4096 draws per requested milestone
    const uint64_t max_samples = h->cfg.max_samples ? h->cfg.max_samples : 4096ull * n_max + (1ull << 22);
    for (double& t : h->t_ms) t = 0.0;
    h->n_candidates = 0;
    h->knn_failed_rows = 0;
    PrmState st{};
    OX_TRY(write_state(h, st));
    h->valid_rate = 1.0;
    // without a wall clock the roadmap is built in one round; with one, in doubling rounds with the
    // clock read in between (the reference reads it before every sample, prm.rs:118)
    uint32_t target = has_timeout ? std::min<uint32_t>(n_max, 4096u) : n_max;
    uint32_t n_done = 0;  // milestones whose pairs are already connected
    for (;;) {
        // ---- 1. sample until `target` milestones
        HIP_TRY(hipEventRecord(h->ev[0], h->stream));
        OX_TRY(sample_until(h, st, target, max_samples));
        HIP_TRY(hipEventRecord(h->ev[1], h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->t_ms[0] += elapsed_ms(h->ev[0], h->ev[1]);
        const uint32_t n_now = st.n_milestones;
        // ---- 2. pairs (j in [n_done, n_now), i < j) within the connection radius
        if (n_now > n_done && n_now >= 2 && (h->thr_conn >= 0.0 || h->cfg.knn_k)) {
            if (h->cand.n == 0) {
                HIP_TRY(h->cand.alloc(std::min<size_t>(std::max<size_t>(1u << 20, (size_t)64 * n_max), (size_t)1 << 28)));
                h->args.cand = h->cand.p;
                h->args.cand_cap = (uint32_t)h->cand.n;
            }
            const uint32_t knn_k = h->cfg.knn_k;
            if (knn_k) {
                // k-nearest variant: row j searches a radius expected to hold ~6 k earlier milestones -- the milestones are uniform
                // over the valid part of the box (fraction n / n_samples of its volume V): count(r) ~ j c_D r^D / (V f) -- so that its
                // k nearest are almost surely among the hits; a row that falls short gets the exact search afterwards
                const uint32_t dim = h->cfg.dim;
                double vol = 1.0;
                for (uint32_t k2 = 0; k2 < dim; ++k2) vol *= h->cfg.bounds[2 * k2 + 1] - h->cfg.bounds[2 * k2];
                const double frac = st.n_samples ? std::fmin(1.0, (double)n_now / (double)st.n_samples) : 1.0;
                const double c_d = std::pow(3.14159265358979323846, 0.5 * dim) / std::tgamma(0.5 * dim + 1.0);
                const double want = 6.0 * knn_k;
                std::vector<double> thr(h->args.cap, std::numeric_limits<double>::infinity());
                std::vector<float> thr32(h->args.cap, std::numeric_limits<float>::infinity());
                for (uint32_t j = n_done; j < n_now; ++j) {
                    if ((double)j <= want) continue;   // few earlier milestones: all of them are candidates
                    const double rd = want * vol * frac / (c_d * (double)j);
                    const double r = std::pow(rd, 1.0 / dim);
                    thr[j] = r * r;
                    thr32[j] = prm_screen_threshold(h->dp, thr[j]);
                }
                if (h->knn_thr.n < h->args.cap) { HIP_TRY(h->knn_thr.alloc(h->args.cap)); HIP_TRY(h->knn_thr32.alloc(h->args.cap)); }
                HIP_TRY(hipMemcpyAsync(h->knn_thr.p, thr.data(), thr.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(hipMemcpyAsync(h->knn_thr32.p, thr32.data(), thr32.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
            }
            for (;;) {
                st.n_cand = 0;
                OX_TRY(write_state(h, st));
                HIP_TRY(hipEventRecord(h->ev[2], h->stream));
                if (knn_k) launch_prm_pairs(h->dp, h->args, n_done, n_now, std::numeric_limits<double>::infinity(), h->stream, h->knn_thr.p, h->knn_thr32.p);
                else launch_prm_pairs(h->dp, h->args, n_done, n_now, h->thr_conn, h->stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(h->ev[3], h->stream));
                PrmState s2{};
                OX_TRY(read_state(h, s2));
                h->t_ms[1] += elapsed_ms(h->ev[2], h->ev[3]);
                st.n_cand = s2.n_cand;
                if (st.n_cand <= h->args.cand_cap) break;
                // candidate buffer too small: the kernel kept counting, so the exact need is known; run again
                if (st.n_cand > 0x7FFFFFFFull) return fail(OXHIP_ERR_CAPACITY, "more than 2^31 in-radius pairs in one round");
                HIP_TRY(h->cand.alloc((size_t)st.n_cand));
                h->args.cand = h->cand.p;
                h->args.cand_cap = (uint32_t)st.n_cand;
            }
            h->n_candidates += st.n_cand;
            uint2* const cand_all = h->args.cand;
            if (knn_k) {
                // ---- 2b. each row's k nearest among its candidates (sorted by (j, i)); the exact search for rows that fell short
                const uint32_t nc = (uint32_t)st.n_cand, rows = n_now - n_done;
                if (h->knn_keys.n < nc + 1) {
                    HIP_TRY(h->knn_keys.alloc(nc + 1)); HIP_TRY(h->knn_sorted.alloc(nc + 1)); HIP_TRY(h->knn_dist.alloc(nc + 1));
                }
                if (h->knn_sel.n < (size_t)rows * knn_k) HIP_TRY(h->knn_sel.alloc((size_t)rows * knn_k));
                if (h->knn_failed.n < rows) HIP_TRY(h->knn_failed.alloc(rows));
                if (h->knn_counters.n < 2) HIP_TRY(h->knn_counters.alloc(2));
                size_t tb = 0;
                if (nc) {
                    HIP_TRY(prm_sort_keys(nullptr, tb, h->knn_keys.p, h->knn_sorted.p, nc, h->args.cap, h->stream));
                    if (h->sort_tmp.n < tb) HIP_TRY(h->sort_tmp.alloc(tb));
                }
                HIP_TRY(hipEventRecord(h->ev[2], h->stream));
                HIP_TRY(hipMemsetAsync(h->knn_counters.p, 0, 2 * sizeof(uint32_t), h->stream));
                launch_prm_knn_keys(h->args, nc, h->knn_keys.p, h->stream);
                if (nc) HIP_TRY(prm_sort_keys(h->sort_tmp.p, tb, h->knn_keys.p, h->knn_sorted.p, nc, h->args.cap, h->stream));
                launch_prm_knn_select(h->dp, h->args, h->knn_sorted.p, h->knn_dist.p, nc, n_done, n_now, knn_k, h->knn_sel.p, h->knn_counters.p,
                                      h->knn_failed.p, h->stream);
                HIP_TRY(hipGetLastError());
                uint32_t cnt[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(cnt, h->knn_counters.p, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                if (cnt[1]) {
                    launch_prm_knn_brute(h->dp, h->args, h->knn_failed.p, cnt[1], knn_k, h->knn_sel.p, h->knn_counters.p, h->stream);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipMemcpyAsync(cnt, h->knn_counters.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                    HIP_TRY(hipStreamSynchronize(h->stream));
                    h->knn_failed_rows += cnt[1];
                }
                HIP_TRY(hipEventRecord(h->ev[3], h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                h->t_ms[1] += elapsed_ms(h->ev[2], h->ev[3]);
                st.n_cand = cnt[0];
                h->args.cand = h->knn_sel.p;   // the edge kernel checks the selected pairs
            }
            // ---- 3. check_motion per candidate; keys grow by at most 2 per candidate
            const uint64_t need = (uint64_t)st.n_keys + 2ull * st.n_cand;
            if (need > 0xFFFFFFFFull) return fail(OXHIP_ERR_CAPACITY, "more than 2^32 directed edges");
            if (need > h->keys.n) {
                DevBuf<uint64_t> bigger;
                HIP_TRY(bigger.alloc(std::max<size_t>((size_t)need, h->keys.n * 2)));
                if (st.n_keys)
                    HIP_TRY(hipMemcpyAsync(bigger.p, h->keys.p, (size_t)st.n_keys * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                           h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                std::swap(bigger.p, h->keys.p);
                std::swap(bigger.n, h->keys.n);
                h->args.keys = h->keys.p;
            }
            HIP_TRY(hipEventRecord(h->ev[2], h->stream));
            launch_prm_edges(h->dp, h->args, (uint32_t)st.n_cand, h->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(h->ev[3], h->stream));
            OX_TRY(read_state(h, st));
            h->t_ms[2] += elapsed_ms(h->ev[2], h->ev[3]);
            h->args.cand = cand_all;
        }
        n_done = n_now;
        if (n_now < target) break;                       // max_samples exhausted
        if (n_now >= n_max) break;
        if (has_timeout && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > h->cfg.timeout)
            break;                                       // prm.rs:118-120
        target = target > n_max / 2 ? n_max : target * 2;
    }
    // ---- 4. sort the directed keys -> every node's neighbours in ascending order (the reference's `edges`)
    const uint32_t n = st.n_milestones, n_keys = st.n_keys;
    size_t tmp_bytes = 0;
    if (n_keys) {
        // (re)size the sort's buffers before the timed region: hipMalloc is a blocking call
        if (h->keys_sorted.n < n_keys) { HIP_TRY(h->keys_sorted.alloc(n_keys)); HIP_TRY(h->nbrs.alloc(n_keys)); }
        HIP_TRY(prm_sort_keys(nullptr, tmp_bytes, h->keys.p, h->keys_sorted.p, n_keys, h->args.cap, h->stream));
        if (h->sort_tmp.n < tmp_bytes) HIP_TRY(h->sort_tmp.alloc(tmp_bytes));
    }
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
    if (n_keys) {
        HIP_TRY(prm_sort_keys(h->sort_tmp.p, tmp_bytes, h->keys.p, h->keys_sorted.p, n_keys, h->args.cap, h->stream));
        launch_prm_csr(h->keys_sorted.p, n_keys, n, h->args.cap, h->offsets.p, h->nbrs.p, h->stream);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemsetAsync(h->offsets.p, 0, ((size_t)n + 1) * sizeof(uint32_t), h->stream));
    }
    HIP_TRY(hipEventRecord(h->ev[5], h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->t_ms[3] = elapsed_ms(h->ev[4], h->ev[5]);
    h->n_keys = n_keys;
    h->host_copy = false;   // fetched by the first get_roadmap / solve
    h->n = n;
    h->n_samples = st.n_samples;
    h->redraw_batches = st.redraw_batches;
    return OXHIP_OK;
}

int32_t oxhip_prm_knn_exact_rows(oxhip_prm* h, uint32_t* rows) {
    if (!h || !rows) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    *rows = h->knn_failed_rows;
    return OXHIP_OK;
}

int32_t oxhip_prm_get_sizes(oxhip_prm* h, uint32_t* n_milestones, uint64_t* n_edge_entries, uint64_t* n_samples) {
    if (!h) return fail(OXHIP_ERR_BAD_ARG, "null handle");
    if (n_milestones) *n_milestones = h->n;
    if (n_edge_entries) *n_edge_entries = h->n_keys;
    if (n_samples) *n_samples = h->n_samples;
    return OXHIP_OK;
}

// get_roadmap (prm.rs:82-84)
int32_t oxhip_prm_get_roadmap(oxhip_prm* h, double* states, uint32_t cap_nodes, uint64_t* offsets, uint32_t* neighbours,
                              uint64_t cap_entries) {
    if (!h) return fail(OXHIP_ERR_BAD_ARG, "null handle");
    if (h->n) {
        OX_TRY(select_device(h->cfg.device));
        OX_TRY(fetch_roadmap(h));
    }
    if ((states || offsets) && cap_nodes < h->n) return fail(OXHIP_ERR_CAPACITY, "roadmap buffers too small");
    if (neighbours && cap_entries < h->h_nbrs.size()) return fail(OXHIP_ERR_CAPACITY, "neighbour buffer too small");
    if (states && h->n) std::memcpy(states, h->h_states.data(), h->h_states.size() * sizeof(double));
    if (offsets) {
        for (uint32_t i = 0; i <= h->n; ++i) offsets[i] = h->h_offsets.empty() ? 0 : h->h_offsets[i];
    }
    if (neighbours && !h->h_nbrs.empty()) std::memcpy(neighbours, h->h_nbrs.data(), h->h_nbrs.size() * sizeof(uint32_t));
    return OXHIP_OK;
}

// Planner::solve (prm.rs:227-307)
int32_t oxhip_prm_solve(oxhip_prm* h, double timeout_s, double* path, uint32_t cap_states, uint32_t* len) {
    if (!h || !len) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    *len = 0;
    h->start_conn.clear();
    h->goal_idx.clear();
    if (!h->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");      // prm.rs:229-236
    if (h->n == 0) return fail(OXHIP_ERR_UNSAMPLED_STATE_SPACE, "construct_roadmap() left no milestones");  // prm.rs:239-241
    OX_TRY(select_device(h->cfg.device));
    OX_TRY(fetch_roadmap(h));
    const uint32_t n = h->n, dim = h->cfg.dim;
    DevParams qdp = h->dp;   // the start may lie anywhere: widen the filter's absolute margin for this launch
    for (uint32_t k = 0; k < dim; ++k) qdp.filt_abs = std::fmax(qdp.filt_abs, 1e-9 * std::fabs(h->query.start[k]));
    HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    launch_prm_query(qdp, h->args, n, h->query, h->thr_conn, h->flags.p, h->start_valid.p, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev[1], h->stream));
    std::vector<uint8_t> flags(n);
    uint32_t start_valid = 0;
    HIP_TRY(hipMemcpyAsync(flags.data(), h->flags.p, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&start_valid, h->start_valid.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->t_ms[4] = elapsed_ms(h->ev[0], h->ev[1]);
    if (!start_valid) return fail(OXHIP_ERR_INVALID_START_STATE, "start state is in collision");  // prm.rs:243-246
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < n; ++i) {
        if (flags[i] & 1) h->start_conn.push_back(i);   // prm.rs:249-256
        if (flags[i] & 2) h->goal_idx.push_back(i);     // prm.rs:259-264
    }
    if (h->start_conn.empty() || h->goal_idx.empty())
        return fail(OXHIP_ERR_NO_SOLUTION_FOUND, "start or goal does not connect to the roadmap");  // prm.rs:266-268
    // breadth-first search, prm.rs:270-301 (start connections are enqueued twice, as there)
    std::deque<uint32_t> queue(h->start_conn.begin(), h->start_conn.end());
    std::vector<int64_t> parent(n, -2);   // parent_map: -2 absent, -1 = Some(None)
    std::vector<uint8_t> visited(n, 0);
    for (uint32_t s : h->start_conn) {
        queue.push_back(s);
        parent[s] = -1;
        visited[s] = 1;
    }
    const bool has_timeout = timeout_s > 0.0 && std::isfinite(timeout_s);
    int64_t goal_reached = -1;
    while (!queue.empty()) {
        const uint32_t cur = queue.front();
        queue.pop_front();
        if (has_timeout && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            h->t_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return fail(OXHIP_ERR_TIMEOUT, "graph search timed out");  // prm.rs:285-287
        }
        if (flags[cur] & 2) { goal_reached = cur; break; }             // goal_indices.contains(&current_idx)
        for (uint32_t e = h->h_offsets[cur]; e < h->h_offsets[cur + 1]; ++e) {
            const uint32_t nb = h->h_nbrs[e];
            if (!visited[nb]) {
                visited[nb] = 1;
                parent[nb] = (int64_t)cur;
                queue.push_back(nb);
            }
        }
    }
    h->t_ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (goal_reached < 0) return fail(OXHIP_ERR_NO_SOLUTION_FOUND, "goal is not reachable on the roadmap");  // prm.rs:304
    // reconstruct_path (prm.rs:189-208): [start] ++ (root connection ... goal milestone)
    std::vector<uint32_t> chain;
    for (int64_t c = goal_reached; c >= 0; c = parent[(size_t)c]) chain.push_back((uint32_t)c);
    *len = (uint32_t)chain.size() + 1;
    if (!path) return OXHIP_OK;
    if (*len > cap_states) return fail(OXHIP_ERR_CAPACITY, "path buffer too small");
    std::memcpy(path, h->query.start, dim * sizeof(double));
    for (size_t j = 0; j < chain.size(); ++j)
        std::memcpy(path + (j + 1) * dim, h->h_states.data() + (size_t)chain[chain.size() - 1 - j] * dim, dim * sizeof(double));
    return OXHIP_OK;
}

int32_t oxhip_prm_get_query_sets(oxhip_prm* h, uint32_t* start_connections, uint32_t cap_start, uint32_t* n_start,
                                 uint32_t* goal_indices, uint32_t cap_goal, uint32_t* n_goal) {
    if (!h) return fail(OXHIP_ERR_BAD_ARG, "null handle");
    if (n_start) *n_start = (uint32_t)h->start_conn.size();
    if (n_goal) *n_goal = (uint32_t)h->goal_idx.size();
    if (start_connections) {
        if (cap_start < h->start_conn.size()) return fail(OXHIP_ERR_CAPACITY, "start_connections buffer too small");
        if (!h->start_conn.empty()) std::memcpy(start_connections, h->start_conn.data(), h->start_conn.size() * sizeof(uint32_t));
    }
    if (goal_indices) {
        if (cap_goal < h->goal_idx.size()) return fail(OXHIP_ERR_CAPACITY, "goal_indices buffer too small");
        if (!h->goal_idx.empty()) std::memcpy(goal_indices, h->goal_idx.data(), h->goal_idx.size() * sizeof(uint32_t));
    }
    return OXHIP_OK;
}

int32_t oxhip_prm_last_timing(oxhip_prm* h, double* phase_ms, uint64_t* n_candidates, uint32_t* redraw_batches) {
    if (!h) return fail(OXHIP_ERR_BAD_ARG, "null handle");
    if (phase_ms) for (int i = 0; i < 6; ++i) phase_ms[i] = h->t_ms[i];
    if (n_candidates) *n_candidates = h->n_candidates;
    if (redraw_batches) *redraw_batches = h->redraw_batches;
    return OXHIP_OK;
}

}  // extern "C"
