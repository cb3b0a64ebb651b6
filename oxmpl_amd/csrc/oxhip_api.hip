// oxhip_api.hip -- implementation of the C ABI declared in include/oxmpl_hip.h.
//
// Host side of the drop-in boundary: validates what the reference would reject or panic on
// (real_vector_state_space.rs:69-93,239-244; rand's Bernoulli::new), precomputes the values
// the reference recomputes per call (extent, lvsl: rvss.rs:103-118,251-253), owns the device
// buffers and launches the kernels.  There is no CPU fallback: without a HIP device every
// entry point that computes returns OXHIP_ERR_NO_DEVICE.
#include "../../include/oxmpl_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "oxhip_host.hpp"
#include "oxhip_internal.hpp"
#include "rrt_device.hpp"

using namespace oxhip;

#include <mutex>
#include <utility>
namespace {
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;
constexpr size_t kStreamPoolMax = 32;
}
namespace oxhip {
hipError_t oxhip_stream_acquire(int device, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = g_stream_pool.size(); i-- > 0;)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void oxhip_stream_release(int device, hipStream_t s) {
    if (!s) return;
    (void)hipStreamSynchronize(s);
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (g_stream_pool.size() < kStreamPoolMax) { g_stream_pool.emplace_back(device, s); return; }
    }
    (void)hipStreamDestroy(s);
}
}  // namespace oxhip

struct oxhip_rrt_batch {
    oxhip_rrt_config cfg{};
    DevParams dp{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf<double> tree, goal_c, goal_thr, goal_r, sph_c, sph_thr, sph_filt, box_lo, box_hi;
    std::vector<double> sph_centres, sph_radii;  // host copies (AoS) for the filter thresholds
    bool filt_dirty = true;
    // rrt_cells.hip: cell blocks, flat lists of small trees, node-major coordinates, grid descriptors, accumulators and part
    // positions of split frozen launches, the sphere mask grid -- ONE allocation (a hipMalloc costs more than a small kernel:
    // the one-problem Planner::solve pays for every one of them), carved at 256-byte boundaries
    DevBuf<uint8_t> cell_arena;
    CellMeta* cell_meta_p = nullptr;
    uint64_t* sph_grid_p = nullptr;
    uint64_t* star_grid_p = nullptr;   // RRT*'s edge checks: the mask grid for motions up to search_radius long
    DevBuf<double> star_filt;
    ProblemState* h_states = nullptr;   // pinned: the state array as the host reads it after every launch
    DevBuf<double> tree_b;   // RRTConnect goal trees
    DevBuf<double> segs;            // SE(2): segment soup
    DevBuf<uint16_t> seg_grid;      // ... and the cells' segment lists (rrt_connect_se2.hip, seg_grid_kernel)
    DevBuf<uint64_t> conn_grid;     // RRTConnect in R^2 / R^3: the spheres that reach each cell of the bounds' grid (rrt_connect.hip)
    DevBuf<double> conn_filt;
    DevBuf<double> cost, nb_dist;   // RRT*: cost-to-come, neighbour scratch
    // RRT*: W of the checksum; the decoupled design's buffers (rrt_star_wire.hip)
    DevBuf<uint64_t> wire_chk;
    DevBuf<uint32_t> wired, nbr_cnt, nbr_off, nbr_take, nbr_total;
    DevBuf<double> d_near, sph_r;
    DevBuf<StarEntry> pool;
    DevBuf<StarChunk> chunks;
    DevBuf<uint32_t> chunk_cursor;
    bool star_wired = false;        // RRT*: geometry by rrt_lanes.hip / rrt_cells.hip + the wiring kernels (else rrt_star.hip)
    bool star_geo_cells = false;    // ... by rrt_cells.hip
    hipStream_t stream2 = nullptr;  // RRT*, decoupled: the wiring of a segment runs here while the next segment's pairs are checked on `stream`
    hipEvent_t ev_seg[8] = {}, ev_join = nullptr;
    DevBuf<float> tree32;           // stream / RRT* kernels: fl32 shadow of the tree
    DevBuf<uint32_t> shadow_state;  // [P][2]
    DevBuf<uint32_t> nb_idx;
    DevBuf<int32_t> parent;
    DevBuf<int32_t> parent_b;
    DevBuf<uint8_t> skip;
    DevBuf<ProblemState> state;
    DevBuf<uint64_t> dbg;
    std::vector<double> starts;  // host copy, [P][dim]
    bool is_setup = false;
    double last_kernel_ms = 0.0;
    uint32_t last_launches = 0;
    uint32_t kernel_kind = OXHIP_KERNEL_STREAM;   // the kind a launch uses (KERNEL_AUTO: resolved per launch, see solve)
    uint32_t last_kind = OXHIP_KERNEL_STREAM;     // what the last launch ran
};

extern "C" {

int32_t oxhip_abi_version(void) { return OXHIP_ABI_VERSION; }

const char* oxhip_status_string(int32_t s) {
    switch (s) {
        case OXHIP_OK: return "ok";
        case OXHIP_ERR_TIMEOUT: return "No solution found within timeout.";  // Display strings of error.rs:110-136
        case OXHIP_ERR_NO_SOLUTION_FOUND: return "No solution found.";
        case OXHIP_ERR_PLANNER_UNINITIALISED: return "<Planner>.setup() was not called, thus Planner is uninitialised.";
        case OXHIP_ERR_INVALID_START_STATE: return "Start state is not valid in the current StateSpace.";
        case OXHIP_ERR_UNSAMPLED_STATE_SPACE: return "StateSpace is not sampled. Either Tree or Roadmap is empty.";
        case OXHIP_ERR_BAD_ARG: return "bad argument";
        case OXHIP_ERR_UNBOUNDED: return "Cannot sample uniformly because a dimension is unbounded.";
        case OXHIP_ERR_ZERO_VOLUME: return "Cannot sample from a region with zero volume.";
        case OXHIP_ERR_CAPACITY: return "caller buffer too small";
        case OXHIP_ERR_HIP: return "HIP runtime error";
        case OXHIP_ERR_NO_DEVICE: return "no HIP device (no CPU fallback exists)";
        default: return "unknown status";
    }
}

const char* oxhip_last_error_string(void) { return g_last_error.c_str(); }

int32_t oxhip_device_count(int32_t* count) {
    if (!count) return fail(OXHIP_ERR_BAD_ARG, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = (e == hipSuccess) ? c : 0;
    if (e != hipSuccess || c <= 0) return fail(OXHIP_ERR_NO_DEVICE, "no HIP device visible");
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_create(const oxhip_rrt_config* cfg, oxhip_rrt_batch** out) {
    if (!cfg || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(oxhip_rrt_config)) return fail(OXHIP_ERR_BAD_ARG, "struct_size mismatch");
    if (cfg->dim == 0 || cfg->dim > OXHIP_MAX_DIM) return fail(OXHIP_ERR_BAD_ARG, "dim must be in 1..8");
    if (cfg->n_problems == 0 || cfg->max_nodes == 0) return fail(OXHIP_ERR_BAD_ARG, "n_problems and max_nodes must be > 0");
    if (cfg->max_nodes > (1u << 30)) return fail(OXHIP_ERR_BAD_ARG, "max_nodes too large");
    if (!(cfg->goal_bias >= 0.0 && cfg->goal_bias <= 1.0))
        return fail(OXHIP_ERR_BAD_ARG, "goal_bias outside [0,1] (rand Bernoulli::new would fail)");
    if (!(cfg->max_distance > 0.0) || !std::isfinite(cfg->max_distance))
        return fail(OXHIP_ERR_BAD_ARG, "max_distance must be finite and > 0");
    if (cfg->kernel > OXHIP_KERNEL_CELLS) return fail(OXHIP_ERR_BAD_ARG, "unknown kernel kind");
    if (cfg->kernel == OXHIP_KERNEL_RETIRED_3 || cfg->kernel == OXHIP_KERNEL_RETIRED_4)
        return fail(OXHIP_ERR_BAD_ARG, "kernel kinds 3 (box-pruned scan) and 4 (lane-group resolver) were retired in ABI version 2");
    if (cfg->planner > OXHIP_PLANNER_RRT_STAR) return fail(OXHIP_ERR_BAD_ARG, "unknown planner kind");
    if (cfg->frozen_split > 64) return fail(OXHIP_ERR_BAD_ARG, "frozen_split must be 0 (automatic) or 1 .. 64");
    if (cfg->kernel == OXHIP_KERNEL_CELLS && cfg->planner == OXHIP_PLANNER_RRT_CONNECT)
        return fail(OXHIP_ERR_BAD_ARG, "the cell-grid kernel runs RRT, and the geometry of the decoupled RRT*");
    if (cfg->planner == OXHIP_PLANNER_RRT_CONNECT && cfg->kernel >= OXHIP_KERNEL_RESIDENT)
        return fail(OXHIP_ERR_BAD_ARG, "RRTConnect runs on the stream kernel only");
    // RRT*: KERNEL_STREAM = rrt_star.hip (one workgroup per problem, everything in one kernel); KERNEL_LANES / KERNEL_CELLS = the
    // decoupled design (geometry by rrt_lanes.hip / rrt_cells.hip, then the wiring kernels of rrt_star_wire.hip); KERNEL_AUTO = the
    // decoupled design where it exists, its geometry by rrt_cells.hip in R^2 / R^3
    if (cfg->planner == OXHIP_PLANNER_RRT_STAR && cfg->kernel != OXHIP_KERNEL_AUTO && cfg->kernel != OXHIP_KERNEL_STREAM &&
        cfg->kernel != OXHIP_KERNEL_LANES && cfg->kernel != OXHIP_KERNEL_CELLS)
        return fail(OXHIP_ERR_BAD_ARG, "RRT* runs on the stream kernel or on the lane-per-query / cell-grid kernel + wiring kernels");
    if (cfg->planner == OXHIP_PLANNER_RRT_STAR && std::isnan(cfg->search_radius))
        return fail(OXHIP_ERR_BAD_ARG, "search_radius is NaN");
    if (cfg->goal_sampler > OXHIP_GOAL_SAMPLE_UNIFORM_DISC) return fail(OXHIP_ERR_BAD_ARG, "unknown goal sampler");
    if (cfg->goal_sampler == OXHIP_GOAL_SAMPLE_UNIFORM_DISC && (cfg->dim != 2 || cfg->space != OXHIP_SPACE_REAL_VECTOR))
        return fail(OXHIP_ERR_BAD_ARG, "the disc sampler (rrt_rvss_tests.rs:55-66) is defined for RealVectorStateSpace(2)");
    if (cfg->goal_sampler == OXHIP_GOAL_SAMPLE_UNIFORM_DISC && (cfg->planner == OXHIP_PLANNER_RRT_CONNECT || cfg->kernel == OXHIP_KERNEL_RESIDENT))
        return fail(OXHIP_ERR_BAD_ARG, "the disc sampler is built for RRT / RRT* on the stream, lane-per-query and cell-grid kernels");
    double fraction = cfg->lvs_fraction, res = 0.0;
    double th_lo = 0.0, th_hi = 0.0;
    if (cfg->space > OXHIP_SPACE_SE2) return fail(OXHIP_ERR_BAD_ARG, "unknown space kind");
    if (cfg->space == OXHIP_SPACE_SE2) {
        if (cfg->dim != 3) return fail(OXHIP_ERR_BAD_ARG, "SE(2) states are (x, y, theta): dim must be 3");
        if (cfg->planner != OXHIP_PLANNER_RRT_CONNECT) return fail(OXHIP_ERR_BAD_ARG, "SE(2) is built for RRTConnect only");
        // SO2StateSpace::new (so2_state_space.rs:57-72): lo >= hi is InvalidBound; bounds clamped to [-PI, PI]
        const double pi = 3.14159265358979323846;
        th_lo = cfg->bounds[4]; th_hi = cfg->bounds[5];
        if (!(th_lo < th_hi)) return fail(OXHIP_ERR_ZERO_VOLUME, "theta: lower bound >= upper bound");
        th_lo = th_lo > -pi ? th_lo : -pi;
        th_hi = th_hi < pi ? th_hi : pi;
        // extent = extent_xy + 0.5 * PI (so2_state_space.rs:78-80); lvsl = extent * fraction; res = lvsl * 0.1
        int32_t sr = space_resolution(2, cfg->bounds, fraction, res);   // validates the (x, y) bounds, clamps the fraction
        if (sr != OXHIP_OK) return sr;
        double acc = 0.0;
        for (uint32_t k = 0; k < 2; ++k) { double w = cfg->bounds[2 * k + 1] - cfg->bounds[2 * k]; double sq = w * w; acc = acc + sq; }
        const double extent = std::sqrt(acc) + 0.5 * pi;
        const double lvsl = extent * fraction;
        res = lvsl * 0.1;
    } else {
        int32_t sr = space_resolution(cfg->dim, cfg->bounds, fraction, res);
        if (sr != OXHIP_OK) return sr;
    }
    if (cfg->max_distance / res > 1e6) return fail(OXHIP_ERR_BAD_ARG, "more than 1e6 validity checks per edge");

    int32_t st = select_device(cfg->device);
    if (st != OXHIP_OK) return st;

    auto* b = new oxhip_rrt_batch();
    b->cfg = *cfg;
    b->cfg.lvs_fraction = fraction;
    const uint32_t P = cfg->n_problems, dim = cfg->dim;
    const uint32_t cap = ((cfg->max_nodes + 1023u) / 1024u) * 1024u;
    DevParams& dp = b->dp;
    dp.dim = dim; dp.n_problems = P; dp.cap = cap; dp.max_nodes = cfg->max_nodes;
    for (uint32_t k = 0; k < dim; ++k) {
        dp.lo[k] = cfg->bounds[2 * k];
        dp.hi[k] = cfg->bounds[2 * k + 1];
        dp.scale[k] = dp.hi[k] - dp.lo[k];
    }
    if (cfg->space == OXHIP_SPACE_SE2) { dp.lo[2] = th_lo; dp.hi[2] = th_hi; dp.scale[2] = th_hi - th_lo; }
    dp.space = cfg->space;
    dp.max_distance = cfg->max_distance;
    dp.res = res;
    if (cfg->planner == OXHIP_PLANNER_RRT_CONNECT) {   // rrt_connect.hip, rrt_connect_se2.hip: the step count of an Advanced extend's motion, when it is safely known
        const double r = cfg->max_distance / res, c = std::ceil(r);
        const double gap = std::fmin(r - (c - 1.0), c - r);
        if (std::isfinite(r) && c >= 1.0 && c < 4294967295.0 && gap > 1e-6) {
            dp.adv_steps = (uint32_t)c;
            dp.adv_slack = 0.5 * gap * res;
        }
    }
    dp.p_int = bernoulli_p_int(cfg->goal_bias);
    dp.seed = cfg->seed;
    dp.dbg_flags = cfg->debug_flags;   // oxhip_debug_flag bits: test-only, results identical (the library reads no environment variable)
    dp.goal_sampler = cfg->goal_sampler;
    dp.first_problem_id = cfg->first_problem_id;
    dp.stop_at_goal = cfg->stop_at_goal ? 1 : 0;
    dp.t_steer = sqrt_le_threshold(cfg->max_distance);
    dp.thr_search = sqrt_lt_threshold(cfg->search_radius);

    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    chk(oxhip_stream_acquire(cfg->device, &b->stream));
    chk(hipEventCreate(&b->ev0));
    chk(hipEventCreate(&b->ev1));
    chk(b->tree.alloc((size_t)P * dim * cap));
    chk(b->parent.alloc((size_t)P * cap));
    chk(b->skip.alloc((size_t)P * cap));
    if (cfg->planner == OXHIP_PLANNER_RRT_CONNECT) {
        chk(b->tree_b.alloc((size_t)P * dim * cap));
        chk(b->parent_b.alloc((size_t)P * cap));
    }
    if (cfg->planner == OXHIP_PLANNER_RRT_STAR) {
        const bool geo_cells = cfg->kernel != OXHIP_KERNEL_LANES && cells_supported(dim, cap);
        const bool geo_lanes = cfg->kernel != OXHIP_KERNEL_CELLS && lanes_supported(dim, cap);
        const bool can_wire = star_wire_supported(dim) && (geo_cells || geo_lanes);
        b->star_wired = cfg->kernel == OXHIP_KERNEL_LANES || cfg->kernel == OXHIP_KERNEL_CELLS || (cfg->kernel == OXHIP_KERNEL_AUTO && can_wire);
        if (b->star_wired && !can_wire) {
            oxhip_rrt_batch_destroy(b);
            return fail(OXHIP_ERR_BAD_ARG, "decoupled RRT*: neither geometry kernel supports this (dim, max_nodes)");
        }
        b->star_geo_cells = b->star_wired && geo_cells;
        chk(b->cost.alloc((size_t)P * cap));
        chk(b->wire_chk.alloc(P));
        if (b->star_wired && e == hipSuccess) {
            // neighbour lists: a pool segment per problem; a round wires the longest prefix of a problem's pending nodes whose
            // lists fit (mean list length at radius 1 in configs[1]'s world: ~40), so the size bounds memory, not the result.
            // Footprint per problem: share x 16 B of pool + (share / 32 + cap) x 144 B of chunk store + 4 x cap x 4..8 B; the
            // default share is 64 x cap, cut so that pool + chunks of the whole batch stay below 24 GB.
            uint64_t share = 64ull * cap;
            const uint64_t budget_entries = (24ull << 30) / (sizeof(StarEntry) + sizeof(StarChunk) / 32) / P;
            if (share > budget_entries) share = budget_entries;
            if (cfg->star_pool_share != 0 && cfg->star_pool_share < share) share = cfg->star_pool_share;
            if (share < cap) share = cap;   // a single list is at most cap entries long: every round wires at least one node
            dp.pool_share = (uint32_t)share;
            hipError_t ew = hipSuccess;
            auto chkw = [&](hipError_t r) { if (ew == hipSuccess) ew = r; };
            chkw(oxhip_stream_acquire(cfg->device, &b->stream2));
            for (auto& ev : b->ev_seg) chkw(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            chkw(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
            chkw(b->pool.alloc((size_t)P * share + 64));   // (+ padding: masked lanes of the wiring kernel read one entry past an empty list)
            dp.chunk_share = (uint32_t)(share / 32 + cap);   // enough for any set of lists that fits the pool segment (one partial chunk per node)
            chkw(b->chunks.alloc((size_t)P * dp.chunk_share));
            chkw(b->chunk_cursor.alloc(P));
            chkw(b->wired.alloc(P));
            chkw(b->nbr_take.alloc(P));
            chkw(b->nbr_total.alloc(P));
            chkw(b->nbr_cnt.alloc((size_t)P * cap));
            chkw(b->nbr_off.alloc((size_t)P * cap));
            chkw(b->d_near.alloc((size_t)P * cap));
            if (ew != hipSuccess && cfg->kernel == OXHIP_KERNEL_AUTO) {
                // KERNEL_AUTO promised "whatever runs": the one-kernel design (rrt_star.hip) needs 1/60 of this memory
                (void)hipGetLastError();
                b->pool.release(); b->chunks.release(); b->chunk_cursor.release(); b->wired.release(); b->nbr_take.release();
                b->nbr_total.release(); b->nbr_cnt.release(); b->nbr_off.release(); b->d_near.release();
                dp.pool_share = dp.chunk_share = 0;
                b->star_wired = false;
                b->star_geo_cells = false;
            } else {
                chk(ew);
            }
        }
        if (!b->star_wired) {
            chk(b->nb_idx.alloc((size_t)P * cap));
            chk(b->nb_dist.alloc((size_t)P * cap));
        }
    }
    chk(b->state.alloc(P));
    chk(b->goal_c.alloc((size_t)P * dim));
    chk(b->goal_thr.alloc(P));
    chk(b->goal_r.alloc(P));
    if (e != hipSuccess) {
        std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e);
        oxhip_rrt_batch_destroy(b);
        return fail(OXHIP_ERR_HIP, msg);
    }
    dp.tree = b->tree.p; dp.parent = b->parent.p; dp.skip = b->skip.p; dp.state = b->state.p;
    dp.tree_b = b->tree_b.p; dp.parent_b = b->parent_b.p;
    dp.cost = b->cost.p; dp.nb_idx = b->nb_idx.p; dp.nb_dist = b->nb_dist.p;
    dp.wire_chk = b->wire_chk.p; dp.wired = b->wired.p; dp.nbr_cnt = b->nbr_cnt.p; dp.nbr_off = b->nbr_off.p;
    dp.nbr_take = b->nbr_take.p; dp.nbr_total = b->nbr_total.p; dp.d_near = b->d_near.p; dp.pool = b->pool.p;
    dp.chunks = b->chunks.p; dp.chunk_cursor = b->chunk_cursor.p;
    dp.goal_c = b->goal_c.p; dp.goal_thr = b->goal_thr.p; dp.goal_r = b->goal_r.p;

    uint32_t kind = cfg->kernel;
    if (cfg->planner != OXHIP_PLANNER_RRT) kind = b->star_wired ? (b->star_geo_cells ? OXHIP_KERNEL_CELLS : OXHIP_KERNEL_LANES) : OXHIP_KERNEL_STREAM;
    if (kind == OXHIP_KERNEL_AUTO)
    {
        // rrt_cells.hip (one wave per problem, R^2 / R^3, trees up to 64,512 nodes) wherever it exists: since its rounds
        // repair overtaken queries in place it grows ONE tree to 10,000 nodes in 4.3 ms -- faster than rrt_lanes.hip, which gives
        // the problem a whole CU (5.6 ms) -- and 256 of them in 4.9 ms (6.2 ms), profiles/r3_single/.  rrt_lanes.hip serves
        // R^4 .. R^6 (and trees that fit its register rows), rrt_stream.hip everything else.
        kind = cells_supported(dim, cap) ? OXHIP_KERNEL_CELLS : lanes_supported(dim, cap) ? OXHIP_KERNEL_LANES : OXHIP_KERNEL_STREAM;
    }
    if (kind == OXHIP_KERNEL_CELLS) {
        if (!cells_supported(dim, cap)) {
            oxhip_rrt_batch_destroy(b);
            return fail(OXHIP_ERR_BAD_ARG, "cell-grid kernel: R^2 / R^3 trees of at most 64,512 nodes");
        }
        dp.cell_level_max = cells_level_max(dim, cap);
        // head blocks for the finest grid this capacity reaches + the overflow blocks (n nodes overflow into at most n / 7)
        dp.cell_blocks = cells_head_blocks(dim, cap) + cap / 7u + 64u;
        // frozen launches: a problem's iterations are independent, so they are divided over enough waves to fill the chip
        // (256 CUs x 12 waves of the frozen specialisation's register budget)
        uint32_t split = cfg->frozen_split;
        if (split == 0) { split = (3072u + P - 1u) / P; if (split > 8u) split = 8u; if (split < 1u) split = 1u; }
        dp.cells_split = split;
        dp.sph_grid_G = sphere_grid_side(dim);
        const size_t grid_cells = dim == 2 ? (size_t)dp.sph_grid_G * dp.sph_grid_G : (size_t)dp.sph_grid_G * dp.sph_grid_G * dp.sph_grid_G;
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255u) & ~(size_t)255u; return at; };
        const size_t o_meta = carve((size_t)P * sizeof(CellMeta)), o_acc = carve((size_t)P * sizeof(CellAcc));   // (zeroed together)
        const size_t zero_bytes = off;
        const size_t o_pos = carve((size_t)P * 64 * sizeof(uint64_t)), o_grid = carve(grid_cells * sizeof(uint64_t));
        const size_t o_grid2 = carve(b->star_wired ? grid_cells * sizeof(uint64_t) : 0);
        const size_t o_flat = carve((size_t)P * 4096 * 4 * sizeof(float)), o_xyz = carve((size_t)P * cap * 4 * sizeof(double));
        const size_t o_blk = carve((size_t)P * dp.cell_blocks * sizeof(CellBlock));
        hipError_t e2 = b->cell_arena.alloc(off);
        if (e2 == hipSuccess) e2 = hipMemset(b->cell_arena.p, 0, zero_bytes);
        if (e2 != hipSuccess) {
            std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e2);
            oxhip_rrt_batch_destroy(b);
            return fail(OXHIP_ERR_HIP, msg);
        }
        uint8_t* base = b->cell_arena.p;
        b->cell_meta_p = reinterpret_cast<CellMeta*>(base + o_meta);
        b->sph_grid_p = reinterpret_cast<uint64_t*>(base + o_grid);
        if (b->star_wired) b->star_grid_p = reinterpret_cast<uint64_t*>(base + o_grid2);
        dp.cell_blk = reinterpret_cast<CellBlock*>(base + o_blk); dp.cell_flat = reinterpret_cast<float*>(base + o_flat);
        dp.cell_xyz = reinterpret_cast<double*>(base + o_xyz); dp.cell_meta = b->cell_meta_p;
        dp.cell_acc = reinterpret_cast<CellAcc*>(base + o_acc); dp.cell_part_pos = reinterpret_cast<uint64_t*>(base + o_pos);
        dp.sph_grid = b->sph_grid_p;
    }
    if (kind == OXHIP_KERNEL_LANES && !lanes_supported(dim, cap)) {
        oxhip_rrt_batch_destroy(b);
        return fail(OXHIP_ERR_BAD_ARG, "resident (lane-per-query) kernel does not support this (dim, max_nodes)");
    }
    if (kind == OXHIP_KERNEL_RESIDENT && !resident_supported(dim, cap)) {
        oxhip_rrt_batch_destroy(b);
        return fail(OXHIP_ERR_BAD_ARG, "resident kernel does not support this (dim, max_nodes)");
    }
    b->kernel_kind = kind;
    b->last_kind = kind;
    if ((cfg->planner == OXHIP_PLANNER_RRT && kind == OXHIP_KERNEL_STREAM) || cfg->planner == OXHIP_PLANNER_RRT_STAR) {   // (RRT*: star_shadow's)
        // the streaming kernels screen their scans over an fl32 shadow of the tree, which they maintain themselves
        hipError_t e2 = b->tree32.alloc((size_t)P * dim * cap);
        if (e2 == hipSuccess) e2 = b->shadow_state.alloc((size_t)P * 2);
        if (e2 == hipSuccess) e2 = hipMemset(b->shadow_state.p, 0, (size_t)P * 2 * sizeof(uint32_t));
        if (e2 != hipSuccess) {
            std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e2);
            oxhip_rrt_batch_destroy(b);
            return fail(OXHIP_ERR_HIP, msg);
        }
        dp.tree32 = b->tree32.p;
        dp.shadow_state = b->shadow_state.p;
    }
    *out = b;
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_destroy(oxhip_rrt_batch* b) {
    if (!b) return OXHIP_OK;
    (void)hipSetDevice(b->cfg.device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->stream2) oxhip_stream_release(b->cfg.device, b->stream2);
    for (auto ev : b->ev_seg) if (ev) (void)hipEventDestroy(ev);
    if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) oxhip_stream_release(b->cfg.device, b->stream);
    if (b->h_states) (void)hipHostFree(b->h_states);
    delete b;
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_set_spheres(oxhip_rrt_batch* b, const double* centres, const double* radii, uint32_t n) {
    if (!b || (n && (!centres || !radii))) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (b->cfg.space == OXHIP_SPACE_SE2) return fail(OXHIP_ERR_BAD_ARG, "SE(2) batches take oxhip_rrt_batch_set_segments");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    const uint32_t dim = b->cfg.dim;
    std::vector<double> c((size_t)dim * n), thr(n);
    for (uint32_t j = 0; j < n; ++j) {
        for (uint32_t k = 0; k < dim; ++k) {
            double v = centres[(size_t)j * dim + k];
            if (!(std::fabs(v) <= kMaxMagnitude)) return fail(OXHIP_ERR_BAD_ARG, "sphere centre not finite / too large");
            c[(size_t)k * n + j] = v;  // SoA [dim][n]
        }
        thr[j] = sqrt_le_threshold(radii[j]);
    }
    if ((st = upload(b->sph_c, c, b->stream)) != OXHIP_OK) return st;
    if ((st = upload(b->sph_thr, thr, b->stream)) != OXHIP_OK) return st;
    b->dp.n_spheres = n; b->dp.sph_c = b->sph_c.p; b->dp.sph_thr = b->sph_thr.p;
    b->sph_centres.assign(centres, centres + (size_t)n * dim);
    b->sph_radii.assign(radii, radii + n);
    b->filt_dirty = true;
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_set_boxes(oxhip_rrt_batch* b, const double* lo, const double* hi, uint32_t n) {
    if (!b || (n && (!lo || !hi))) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (b->cfg.space == OXHIP_SPACE_SE2) return fail(OXHIP_ERR_BAD_ARG, "SE(2) batches take oxhip_rrt_batch_set_segments");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    const uint32_t dim = b->cfg.dim;
    std::vector<double> l((size_t)dim * n), h((size_t)dim * n);
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t k = 0; k < dim; ++k) {
            l[(size_t)k * n + j] = lo[(size_t)j * dim + k];
            h[(size_t)k * n + j] = hi[(size_t)j * dim + k];
        }
    if ((st = upload(b->box_lo, l, b->stream)) != OXHIP_OK) return st;
    if ((st = upload(b->box_hi, h, b->stream)) != OXHIP_OK) return st;
    b->dp.n_boxes = n; b->dp.box_lo = b->box_lo.p; b->dp.box_hi = b->box_hi.p;
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_set_segments(oxhip_rrt_batch* b, const double* segments, uint32_t n, double clearance) {
    if (!b || (n && !segments)) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (b->cfg.space != OXHIP_SPACE_SE2) return fail(OXHIP_ERR_BAD_ARG, "segments describe the SE(2) checker");
    if (std::isnan(clearance)) return fail(OXHIP_ERR_BAD_ARG, "clearance is NaN");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<double> v(segments, segments + (size_t)4 * n);
    for (double x : v) if (!(std::fabs(x) <= kMaxMagnitude)) return fail(OXHIP_ERR_BAD_ARG, "segment endpoint not finite / too large");
    if ((st = upload(b->segs, v, b->stream)) != OXHIP_OK) return st;
    b->dp.segs = b->segs.p;
    b->dp.n_segs = n;
    b->dp.seg_thr = sqrt_le_threshold(clearance);
    // the grid the motion check looks its segments up in: over the (x, y) bounds, which are finite here (create() refused others)
    b->dp.seg_grid = nullptr;
    const double wx = b->dp.hi[0] - b->dp.lo[0], wy = b->dp.hi[1] - b->dp.lo[1];
    if (n > 0 && (b->cfg.debug_flags & OXHIP_DEBUG_SE2_NO_SEGMENT_GRID) == 0 && std::isfinite(wx) && std::isfinite(wy) && wx > 0.0 && wy > 0.0) {
        const uint32_t G = seg_grid_side();
        if (b->seg_grid.n != (size_t)G * G * 8) HIP_TRY(b->seg_grid.alloc((size_t)G * G * 8));
        b->dp.seg_grid_G = G;
        b->dp.seg_grid_inv[0] = (double)G / wx;
        b->dp.seg_grid_inv[1] = (double)G / wy;
        launch_seg_grid(b->dp, b->seg_grid.p, clearance, b->stream);
        b->dp.seg_grid = b->seg_grid.p;
    }
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_setup(oxhip_rrt_batch* b, const double* starts, const double* goal_centres,
                              const double* goal_radii) {
    if (!b || !starts || !goal_centres || !goal_radii) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    const uint32_t P = b->cfg.n_problems, dim = b->cfg.dim, cap = b->dp.cap;
    for (size_t i = 0; i < (size_t)P * dim; ++i)
        if (!(std::fabs(starts[i]) <= kMaxMagnitude) || !(std::fabs(goal_centres[i]) <= kMaxMagnitude))
            return fail(OXHIP_ERR_BAD_ARG, "start / goal centre not finite or beyond 1e150");
    if (b->cfg.goal_sampler == OXHIP_GOAL_SAMPLE_UNIFORM_DISC)
        for (uint32_t p = 0; p < P; ++p)
            if (!(goal_radii[p] >= 0.0 && goal_radii[p] <= kMaxMagnitude)) return fail(OXHIP_ERR_BAD_ARG, "disc sampler: goal radius must be finite and >= 0");
    b->starts.assign(starts, starts + (size_t)P * dim);
    b->filt_dirty = true;
    std::vector<double> thr(P);
    // R^n: satisfied iff d2 <= T(radius); SE(2): the compound distance is compared with the radius itself
    for (uint32_t p = 0; p < P; ++p) thr[p] = b->cfg.space == OXHIP_SPACE_SE2 ? goal_radii[p] : sqrt_le_threshold(goal_radii[p]);
    std::vector<ProblemState> states(P);
    for (auto& s : states) {
        s = ProblemState{};
        s.checksum = kFnvBasis;
        s.n_nodes = 1;
        s.goal_node = -1;
        s.goal_node_b = -1;
        s.n_nodes_b = b->cfg.planner == OXHIP_PLANNER_RRT_CONNECT ? 1 : 0;
        s.stop_reason = OXHIP_STOP_NONE;
    }
    std::vector<int32_t> root(1, -1);
    HIP_TRY(hipMemcpyAsync(b->goal_c.p, goal_centres, (size_t)P * dim * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->goal_thr.p, thr.data(), P * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->goal_r.p, goal_radii, P * sizeof(double), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->state.p, states.data(), P * sizeof(ProblemState), hipMemcpyHostToDevice, b->stream));
    // tree.clear(); tree.push(Node{start_states[0], None})   rrt.rs:147-155
    // node 0 of coordinate k of problem p lives at tree[(p*dim + k)*cap]: strided 2-D copy
    HIP_TRY(hipMemcpy2DAsync(b->tree.p, (size_t)cap * sizeof(double), starts, sizeof(double), sizeof(double),
                             (size_t)P * dim, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemsetAsync(b->skip.p, 0, (size_t)P * cap, b->stream));
    if (b->shadow_state.p) HIP_TRY(hipMemsetAsync(b->shadow_state.p, 0, (size_t)P * 2 * sizeof(uint32_t), b->stream));  // shadows start over
    if (b->cell_meta_p) HIP_TRY(hipMemsetAsync(b->cell_meta_p, 0, (size_t)P * sizeof(CellMeta), b->stream));              // the grids too
    b->dp.cells_meta_ok = 0;
    std::vector<int32_t> minus1(P, -1);
    HIP_TRY(hipMemcpy2DAsync(b->parent.p, (size_t)cap * sizeof(int32_t), minus1.data(), sizeof(int32_t),
                             sizeof(int32_t), P, hipMemcpyHostToDevice, b->stream));
    if (b->cfg.planner == OXHIP_PLANNER_RRT_CONNECT) {
        // goal_tree.push(Node{goal.sample_goal(), None})  rrt_connect.rs:218-224 (the ball goal samples its centre)
        HIP_TRY(hipMemcpy2DAsync(b->tree_b.p, (size_t)cap * sizeof(double), goal_centres, sizeof(double), sizeof(double),
                                 (size_t)P * dim, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipMemcpy2DAsync(b->parent_b.p, (size_t)cap * sizeof(int32_t), minus1.data(), sizeof(int32_t),
                                 sizeof(int32_t), P, hipMemcpyHostToDevice, b->stream));
    }
    if (b->cfg.planner == OXHIP_PLANNER_RRT_STAR) {   // start node: cost 0.0 (rrt_star.rs:163-167)
        HIP_TRY(hipMemsetAsync(b->cost.p, 0, (size_t)P * cap * sizeof(double), b->stream));
        HIP_TRY(hipMemsetAsync(b->wire_chk.p, 0, (size_t)P * sizeof(uint64_t), b->stream));
        if (b->star_wired) {
            std::vector<uint32_t> one(P, 1u);   // the start node needs no wiring
            HIP_TRY(hipMemcpyAsync(b->wired.p, one.data(), (size_t)P * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->is_setup = true;
    return OXHIP_OK;
}

// Conservative midpoint filter of the resident kernel's motion check.  Every state the check
// interpolates lies within max_distance/2 (+ rounding) of the segment midpoint, so
// d2(centre, mid) > (r + h)^2 proves sphere (centre, r) valid for the whole motion.  h carries a
// relative 1e-6 and an absolute 1e-9 * (largest coordinate magnitude) margin, orders of magnitude
// above the few-ulp rounding of the interpolation; the filter never decides a motion invalid.
static int32_t refresh_filter(oxhip_rrt_batch* b) {
    if (!b->filt_dirty) return OXHIP_OK;
    const uint32_t n = b->dp.n_spheres, dim = b->cfg.dim;
    if (b->cfg.planner == OXHIP_PLANNER_RRT_CONNECT && b->cfg.space == OXHIP_SPACE_REAL_VECTOR) {
        // rrt_connect.hip's motion check looks a state's spheres up (R^2 / R^3, up to 64 spheres, finite bounds): bit j of a cell's
        // mask = sphere j reaches the cell's box (sphere_grid_kernel: squared distance centre - box, the box taken 2^-9 of a cell
        // wider, compared with the sphere's own validity threshold and a relative 1e-9 of slack).  A sphere that is not listed
        // contains no state of the cell: its test would say "no hit".
        b->dp.sph_grid = nullptr;
        bool ok = (dim == 2 || dim == 3) && n >= 1 && n <= 64 && (b->cfg.debug_flags & OXHIP_DEBUG_SE2_NO_SEGMENT_GRID) == 0;
        for (uint32_t k = 0; ok && k < dim; ++k) ok = std::isfinite(b->dp.lo[k]) && std::isfinite(b->dp.hi[k]) && b->dp.hi[k] > b->dp.lo[k];
        if (ok) {
            const uint32_t G = sphere_grid_side(dim);
            const size_t cells = dim == 2 ? (size_t)G * G : (size_t)G * G * G;
            if (b->conn_grid.n != cells) HIP_TRY(b->conn_grid.alloc(cells));
            std::vector<double> f(n);
            for (uint32_t j = 0; j < n; ++j) {
                const double t = sqrt_le_threshold(b->sph_radii[j]);   // valid iff d2 > t
                f[j] = std::isfinite(t) ? (t > 0.0 ? t * (1.0 + 1e-9) : t) : t;
            }
            int32_t st = upload(b->conn_filt, f, b->stream);
            if (st != OXHIP_OK) return st;
            b->dp.sph_grid_G = G;
            launch_sphere_grid(b->dp, b->conn_grid.p, b->conn_filt.p, b->stream);
            b->dp.sph_grid = b->conn_grid.p;
        }
        b->filt_dirty = false;
        return OXHIP_OK;
    }
    double maxabs = 1.0;
    for (uint32_t k = 0; k < 2 * dim; ++k) maxabs = std::fmax(maxabs, std::fabs(b->cfg.bounds[k]));
    for (double v : b->starts) maxabs = std::fmax(maxabs, std::fabs(v));
    for (double v : b->sph_centres) maxabs = std::fmax(maxabs, std::fabs(v));
    const double h = 0.5 * b->cfg.max_distance * (1.0 + 1e-6) + 1e-9 * maxabs;
    std::vector<double> f(n);
    for (uint32_t j = 0; j < n; ++j) {
        const double r = b->sph_radii[j];
        if (std::isnan(r) || std::isinf(r)) f[j] = r < 0.0 ? -1.0 : std::numeric_limits<double>::infinity();
        else if (r < 0.0) f[j] = -1.0;                       // distance > negative radius always holds
        else f[j] = (r + h) * (r + h) * (1.0 + 1e-9);
    }
    int32_t st = upload(b->sph_filt, f, b->stream);
    if (st != OXHIP_OK) return st;
    b->dp.sph_filt = b->sph_filt.p;
    if (b->sph_grid_p && n > 0) launch_sphere_grid(b->dp, b->sph_grid_p, b->sph_filt.p, b->stream);   // (rrt_cells.hip looks the midpoint filter up)
    if (b->star_wired) {   // motion_seq.hpp filters motions of any length: it takes the radii as given and this absolute margin
        if ((st = upload(b->sph_r, b->sph_radii, b->stream)) != OXHIP_OK) return st;
        b->dp.sph_r = b->sph_r.p;
        b->dp.filt_abs = 1e-9 * maxabs;
        b->dp.star_sph_grid = nullptr;
        if (b->star_grid_p && n > 0) {
            // the edge kernel's midpoint filter keeps sphere j when d2(centre, mid) <= (r_j + h)^2 (1 + 1e-9), h = dist / 2 (1 + 1e-6)
            // + filt_abs, dist < search_radius: the grid is built for the largest such ball (either end of h's range when r_j < 0)
            const double h_lo = b->dp.filt_abs, h_hi = 0.5 * b->cfg.search_radius * (1.0 + 1e-6) + b->dp.filt_abs;
            std::vector<double> sf(n);
            for (uint32_t j = 0; j < n; ++j) {
                const double r = b->sph_radii[j];
                const double a = (r + h_lo) * (r + h_lo), c = (r + h_hi) * (r + h_hi);
                sf[j] = (std::isnan(r) || std::isinf(r)) ? std::numeric_limits<double>::infinity()
                                                         : std::fmax(a, c) * (1.0 + 1e-9) * (1.0 + 1e-12);
            }
            if ((st = upload(b->star_filt, sf, b->stream)) != OXHIP_OK) return st;
            launch_sphere_grid(b->dp, b->star_grid_p, b->star_filt.p, b->stream);
            b->dp.star_sph_grid = b->star_grid_p;
        }
    }
    b->filt_dirty = false;
    return OXHIP_OK;
}

static int32_t read_states(oxhip_rrt_batch* b, std::vector<ProblemState>& states);

int32_t oxhip_rrt_batch_set_tree(oxhip_rrt_batch* b, uint32_t problem, const double* states_in, const int32_t* parents_in,
                                 uint32_t n) {
    if (!b || !states_in || !parents_in) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (problem >= b->cfg.n_problems) return fail(OXHIP_ERR_BAD_ARG, "problem index out of range");
    if (n == 0 || n > b->cfg.max_nodes) return fail(OXHIP_ERR_BAD_ARG, "n_nodes must be in 1..max_nodes");
    if (b->cfg.planner != OXHIP_PLANNER_RRT) return fail(OXHIP_ERR_BAD_ARG, "set_tree is for the RRT planner");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    const uint32_t dim = b->cfg.dim, cap = b->dp.cap;
    if (parents_in[0] != -1) return fail(OXHIP_ERR_BAD_ARG, "parents[0] must be -1 (the root)");
    std::vector<double> soa((size_t)dim * n);
    std::vector<uint8_t> skip(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (i > 0 && (parents_in[i] < 0 || (uint32_t)parents_in[i] >= i)) return fail(OXHIP_ERR_BAD_ARG, "parent index must precede its child");
        for (uint32_t k = 0; k < dim; ++k) {
            double v = states_in[(size_t)i * dim + k];
            if (!(std::fabs(v) <= kMaxMagnitude)) return fail(OXHIP_ERR_BAD_ARG, "state not finite or beyond 1e150");
            soa[(size_t)k * n + i] = v;
        }
    }
    // skip flag: a node whose coordinates equal (as values) those of a lower-index node can never be nearest
    {
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        auto key_less = [&](uint32_t a, uint32_t c) {
            for (uint32_t k = 0; k < dim; ++k) {
                double x = soa[(size_t)k * n + a] + 0.0, y = soa[(size_t)k * n + c] + 0.0;  // -0.0 -> +0.0
                if (x < y) return true;
                if (y < x) return false;
            }
            return a < c;
        };
        std::sort(order.begin(), order.end(), key_less);
        for (uint32_t j = 1; j < n; ++j) {
            bool same = true;
            for (uint32_t k = 0; k < dim && same; ++k) same = soa[(size_t)k * n + order[j]] == soa[(size_t)k * n + order[j - 1]];
            if (same) skip[order[j]] = 1;  // order[j-1] has the lower index among equals (ties sort by index)
        }
    }
    HIP_TRY(hipMemcpy2DAsync(b->tree.p + (size_t)problem * dim * cap, (size_t)cap * sizeof(double), soa.data(),
                             (size_t)n * sizeof(double), (size_t)n * sizeof(double), dim, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->parent.p + (size_t)problem * cap, parents_in, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->skip.p + (size_t)problem * cap, skip.data(), n, hipMemcpyHostToDevice, b->stream));
    if (b->shadow_state.p)   // this problem's fl32 shadow starts over
        HIP_TRY(hipMemsetAsync(b->shadow_state.p + 2 * (size_t)problem, 0, 2 * sizeof(uint32_t), b->stream));
    if (b->cell_meta_p) HIP_TRY(hipMemsetAsync(b->cell_meta_p + problem, 0, sizeof(CellMeta), b->stream));   // ... and its cell grid
    b->dp.cells_meta_ok = 0;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    states[problem].n_nodes = n;
    HIP_TRY(hipMemcpyAsync(b->state.p + problem, &states[problem], sizeof(ProblemState), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return OXHIP_OK;
}

static int32_t read_states(oxhip_rrt_batch* b, std::vector<ProblemState>& states) {
    states.resize(b->cfg.n_problems);
    if (!b->h_states) HIP_TRY(hipHostMalloc((void**)&b->h_states, states.size() * sizeof(ProblemState), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(b->h_states, b->state.p, states.size() * sizeof(ProblemState), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    std::memcpy(states.data(), b->h_states, states.size() * sizeof(ProblemState));
    return OXHIP_OK;
}

// Decoupled RRT*: parents and costs of the nodes the RRT kernel just inserted (rrt_star_wire.hip).  A round wires, per
// problem, the longest prefix of its pending nodes whose neighbour lists fit the problem's pool segment; usually one round.
static int32_t wire_new_nodes(oxhip_rrt_batch* b) {
    const uint32_t P = b->cfg.n_problems;
    std::vector<ProblemState> states;
    std::vector<uint32_t> wired(P), take(P), total(P), chunks_used(P);
    for (;;) {
        int32_t st = read_states(b, states);
        if (st != OXHIP_OK) return st;
        HIP_TRY(hipMemcpyAsync(wired.data(), b->wired.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        uint32_t max_pending = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t pend = states[p].n_nodes > wired[p] ? states[p].n_nodes - wired[p] : 0u;
            max_pending = pend > max_pending ? pend : max_pending;
        }
        if (max_pending == 0) return OXHIP_OK;
        uint32_t max_n = 0;
        for (uint32_t p = 0; p < P; ++p) max_n = states[p].n_nodes > max_n ? states[p].n_nodes : max_n;
        launch_star_shadow(b->dp, max_n, b->stream);
        HIP_TRY(hipMemsetAsync(b->chunk_cursor.p, 0, (size_t)P * sizeof(uint32_t), b->stream));
        launch_star_count(b->dp, max_pending, b->stream);
        launch_star_scan(b->dp, b->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(take.data(), b->nbr_take.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipMemcpyAsync(total.data(), b->nbr_total.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        uint32_t max_take = 0, max_total = 0;
        for (uint32_t p = 0; p < P; ++p) {
            max_take = take[p] > max_take ? take[p] : max_take;
            max_total = total[p] > max_total ? total[p] : max_total;
        }
        if (max_take == 0) return fail(OXHIP_ERR_HIP, "RRT* wiring: no node fits the neighbour pool");   // (a list is at most cap <= pool_share long)
        // every pending node of every problem fits this round and the counting pass could keep all it found: the lists are
        // built from its chunks; otherwise (a pool segment or the chunk store ran out) the round searches a second time
        HIP_TRY(hipMemcpyAsync(chunks_used.data(), b->chunk_cursor.p, (size_t)P * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        bool one_pass = (b->dp.dbg_flags & OXHIP_DEBUG_STAR_TWO_PASS) == 0;
        uint32_t max_chunks = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t pend = states[p].n_nodes > wired[p] ? states[p].n_nodes - wired[p] : 0u;
            if (take[p] != pend || chunks_used[p] > b->dp.chunk_share) one_pass = false;
            max_chunks = chunks_used[p] > max_chunks ? chunks_used[p] : max_chunks;
        }
        if (one_pass) launch_star_compact(b->dp, max_chunks, b->stream);
        else launch_star_fill(b->dp, max_take, b->stream);
        // The pairs are checked segment by segment (of the entries) and a segment's nodes -- those whose lists end inside it --
        // are wired on a second stream meanwhile: the wiring kernel is one latency-bound wave per problem and shares the
        // chip with the next segment's throughput-bound edge kernel at almost no cost to either
        const uint32_t segs = (max_total >= 64u * 1024u && (b->dp.dbg_flags & OXHIP_DEBUG_STAR_ONE_SEGMENT) == 0) ? 8u : 1u;
        for (uint32_t sgi = 0; sgi < segs; ++sgi) {
            b->dp.seg_index = sgi;
            b->dp.seg_count = segs;
            launch_star_edges(b->dp, max_total / segs + 2u, b->stream);
            HIP_TRY(hipEventRecord(b->ev_seg[sgi], b->stream));
            HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_seg[sgi], 0));
            launch_star_wire(b->dp, b->stream2);
        }
        HIP_TRY(hipEventRecord(b->ev_join, b->stream2));
        HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
        HIP_TRY(hipGetLastError());
    }
}

int32_t oxhip_rrt_batch_solve(oxhip_rrt_batch* b, uint64_t max_iterations, double timeout_s, uint32_t freeze,
                              int32_t* status_out) {
    if (!b) return fail(OXHIP_ERR_BAD_ARG, "null batch");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");  // rrt.rs:160-163
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    if (b->cfg.planner != OXHIP_PLANNER_RRT && freeze) return fail(OXHIP_ERR_BAD_ARG, "freeze is for the RRT planner");
    if ((st = refresh_filter(b)) != OXHIP_OK) return st;
    if (std::isnan(timeout_s) || timeout_s < 0.0)   // Duration cannot be negative; from_secs_f32 (oxmpl-py rrt.rs:112) panics on both
        return fail(OXHIP_ERR_BAD_ARG, "timeout_s is NaN or negative (0 or +inf = no wall-clock limit)");
    const bool has_timeout = timeout_s > 0.0 && std::isfinite(timeout_s);
    const auto t0 = std::chrono::steady_clock::now();
    // The budget is always cut into bounded launches: the host reads the stop reasons in between, so a problem that can
    // never finish (start enclosed, huge budget) costs one chunk at a time instead of pinning the GPU for 2^40 iterations,
    // and with a timeout the host clock is consulted at the same points (rrt.rs:172-174).  Results do not depend on the
    // cut (a launch resumes from the state array).  A launch's budget stays below 2^31 (32-bit query counters).
    const uint64_t kChunk = has_timeout ? 2048 : 65536;
    const uint64_t chunk = max_iterations < kChunk ? max_iterations : kChunk;
    uint64_t remaining = max_iterations;
    b->last_kernel_ms = 0.0;
    b->last_launches = 0;
    bool timed_out = false;
    std::vector<ProblemState> states;
    while (remaining > 0) {
        uint64_t step = remaining < chunk ? remaining : chunk;
        b->dp.budget = step;
        b->dp.freeze = freeze ? 1 : 0;
        // KERNEL_AUTO = the lane-per-query kernel wherever it exists (R^2 .. R^6, trees that fit its register rows), for frozen
        // and growing launches alike; the stream kernel for everything else
        uint32_t kind = b->kernel_kind;
        b->last_kind = kind;
        HIP_TRY(hipEventRecord(b->ev0, b->stream));
        if (b->cfg.space == OXHIP_SPACE_SE2) launch_rrt_connect_se2(b->dp, b->stream);
        else if (b->cfg.planner == OXHIP_PLANNER_RRT_CONNECT) launch_rrt_connect(b->dp, b->stream);
        else if (b->cfg.planner == OXHIP_PLANNER_RRT_STAR && b->star_wired) {
            // geometry: exactly RRT's loop on the same stream (rrt_star_wire.hip's header); then wire the new nodes
            if (b->star_geo_cells) { launch_rrt_cells(b->dp, b->stream); b->dp.cells_meta_ok = 1; } else launch_rrt_lanes(b->dp, b->stream);
            HIP_TRY(hipGetLastError());
            if ((st = wire_new_nodes(b)) != OXHIP_OK) return st;
        }
        else if (b->cfg.planner == OXHIP_PLANNER_RRT_STAR) launch_rrt_star(b->dp, b->stream);
        else if (kind == OXHIP_KERNEL_CELLS) { launch_rrt_cells(b->dp, b->stream); b->dp.cells_meta_ok = 1; }
        else if (kind == OXHIP_KERNEL_LANES) launch_rrt_lanes(b->dp, b->stream);
        else if (kind == OXHIP_KERNEL_RESIDENT) launch_rrt_resident(b->dp, b->stream);
        else launch_rrt_stream(b->dp, b->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(b->ev1, b->stream));
        if ((st = read_states(b, states)) != OXHIP_OK) return st;   // (the launch's one synchronisation: the stop reasons come with it)
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, b->ev0, b->ev1));
        b->last_kernel_ms += ms;
        b->last_launches++;
        remaining -= step;
        if (remaining == 0) break;
        bool any_running = false;
        for (auto& s : states) if (s.stop_reason == OXHIP_STOP_ITERATIONS) { any_running = true; break; }
        if (!any_running) break;
        if (has_timeout && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            timed_out = true;  // rrt.rs:172-174
            break;
        }
    }
    if (states.empty() && (st = read_states(b, states)) != OXHIP_OK) return st;   // (no launch: max_iterations = 0)
    for (auto& s : states)
        if (s.stop_reason == OXHIP_STOP_INTERNAL) return fail(OXHIP_ERR_HIP, "resident kernel: scanner/resolver hand-off stalled");
    if (timed_out) {
        for (auto& s : states) if (s.stop_reason == OXHIP_STOP_ITERATIONS) s.stop_reason = OXHIP_STOP_TIMEOUT;
        HIP_TRY(hipMemcpyAsync(b->state.p, states.data(), states.size() * sizeof(ProblemState), hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    if (status_out)
        for (uint32_t p = 0; p < b->cfg.n_problems; ++p) {
            const ProblemState& s = states[p];
            status_out[p] = s.goal_node >= 0 ? OXHIP_OK
                          : (s.stop_reason == OXHIP_STOP_TIMEOUT ? OXHIP_ERR_TIMEOUT : OXHIP_ERR_NO_SOLUTION_FOUND);
        }
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_counts(oxhip_rrt_batch* b, uint64_t* iterations, uint32_t* nodes, uint64_t* accepted,
                                   uint64_t* checksum, int32_t* goal_node, int32_t* stop_reason) {
    if (!b) return fail(OXHIP_ERR_BAD_ARG, "null batch");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    if (b->cfg.planner == OXHIP_PLANNER_RRT_STAR) {   // reported checksum = H (iterations) + W (wiring), DESIGN.md section 10
        std::vector<uint64_t> w(b->cfg.n_problems);
        HIP_TRY(hipMemcpyAsync(w.data(), b->wire_chk.p, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        for (uint32_t p = 0; p < b->cfg.n_problems; ++p) states[p].checksum += w[p];
    }
    for (uint32_t p = 0; p < b->cfg.n_problems; ++p) {
        if (iterations) iterations[p] = states[p].iterations;
        if (nodes) nodes[p] = states[p].n_nodes;
        if (accepted) accepted[p] = states[p].accepted;
        if (checksum) checksum[p] = states[p].checksum;
        if (goal_node) goal_node[p] = states[p].goal_node;
        if (stop_reason) stop_reason[p] = states[p].stop_reason;
    }
    return OXHIP_OK;
}

static int32_t fetch_tree(oxhip_rrt_batch* b, uint32_t problem, uint32_t n, std::vector<double>& soa,
                          std::vector<int32_t>& parents, bool goal_tree = false) {
    const uint32_t dim = b->cfg.dim, cap = b->dp.cap;
    const double* tree_base = goal_tree ? b->tree_b.p : b->tree.p;
    const int32_t* parent_base = goal_tree ? b->parent_b.p : b->parent.p;
    soa.resize((size_t)dim * n);
    parents.resize(n);
    // [dim][n] out of [dim][cap]
    HIP_TRY(hipMemcpy2DAsync(soa.data(), (size_t)n * sizeof(double), tree_base + (size_t)problem * dim * cap,
                             (size_t)cap * sizeof(double), (size_t)n * sizeof(double), dim, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(parents.data(), parent_base + (size_t)problem * cap, (size_t)n * sizeof(int32_t),
                           hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_tree(oxhip_rrt_batch* b, uint32_t problem, double* states_out, int32_t* parents_out,
                                 uint32_t cap_nodes, uint32_t* n_nodes) {
    if (!b || !n_nodes) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (problem >= b->cfg.n_problems) return fail(OXHIP_ERR_BAD_ARG, "problem index out of range");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    const uint32_t n = states[problem].n_nodes, dim = b->cfg.dim;
    *n_nodes = n;
    if (n > cap_nodes || (!states_out && !parents_out)) return n > cap_nodes ? fail(OXHIP_ERR_CAPACITY, "tree buffer too small") : OXHIP_OK;
    std::vector<double> soa;
    std::vector<int32_t> par;
    if ((st = fetch_tree(b, problem, n, soa, par)) != OXHIP_OK) return st;
    if (states_out)
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = 0; k < dim; ++k) states_out[(size_t)i * dim + k] = soa[(size_t)k * n + i];
    if (parents_out) std::memcpy(parents_out, par.data(), (size_t)n * sizeof(int32_t));
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_path(oxhip_rrt_batch* b, uint32_t problem, double* states_out, uint32_t cap_states,
                                 uint32_t* len) {
    if (!b || !len) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (problem >= b->cfg.n_problems) return fail(OXHIP_ERR_BAD_ARG, "problem index out of range");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    *len = 0;
    const int32_t goal = states[problem].goal_node;
    if (goal < 0) return OXHIP_OK;
    const uint32_t n = states[problem].n_nodes, dim = b->cfg.dim;
    std::vector<double> soa;
    std::vector<int32_t> par;
    if ((st = fetch_tree(b, problem, n, soa, par)) != OXHIP_OK) return st;
    // reconstruct_path (rrt.rs:118-128): follow parents from the goal node, reverse
    std::vector<uint32_t> chain;
    for (int64_t i = goal; i >= 0; i = par[(size_t)i]) {
        chain.push_back((uint32_t)i);
        if (chain.size() > n) return fail(OXHIP_ERR_HIP, "parent chain is cyclic (corrupt tree)");
    }
    // RRTConnect (rrt_connect.rs:288-304): append the goal-tree chain from the connection node's parent to
    // the goal root (the reversed goal path with its first element, the duplicate connection point, skipped)
    std::vector<double> soa_b;
    std::vector<int32_t> par_b;
    std::vector<uint32_t> chain_b;
    const uint32_t nb = states[problem].n_nodes_b;
    if (b->cfg.planner == OXHIP_PLANNER_RRT_CONNECT && states[problem].goal_node_b >= 0) {
        if ((st = fetch_tree(b, problem, nb, soa_b, par_b, true)) != OXHIP_OK) return st;
        for (int64_t i = par_b[(size_t)states[problem].goal_node_b]; i >= 0; i = par_b[(size_t)i]) {
            chain_b.push_back((uint32_t)i);
            if (chain_b.size() > nb) return fail(OXHIP_ERR_HIP, "goal-tree parent chain is cyclic (corrupt tree)");
        }
    }
    const size_t total = chain.size() + chain_b.size();
    *len = (uint32_t)total;
    if (total > cap_states || !states_out) return total > cap_states ? fail(OXHIP_ERR_CAPACITY, "path buffer too small") : OXHIP_OK;
    for (size_t j = 0; j < chain.size(); ++j) {
        uint32_t i = chain[chain.size() - 1 - j];
        for (uint32_t k = 0; k < dim; ++k) states_out[j * dim + k] = soa[(size_t)k * n + i];
    }
    for (size_t j = 0; j < chain_b.size(); ++j)
        for (uint32_t k = 0; k < dim; ++k) states_out[(chain.size() + j) * dim + k] = soa_b[(size_t)k * nb + chain_b[j]];
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_goal_counts(oxhip_rrt_batch* b, uint32_t* nodes, int32_t* end_node) {
    if (!b) return fail(OXHIP_ERR_BAD_ARG, "null batch");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (b->cfg.planner != OXHIP_PLANNER_RRT_CONNECT) return fail(OXHIP_ERR_BAD_ARG, "not an RRTConnect batch");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    for (uint32_t p = 0; p < b->cfg.n_problems; ++p) {
        if (nodes) nodes[p] = states[p].n_nodes_b;
        if (end_node) end_node[p] = states[p].goal_node_b;
    }
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_goal_tree(oxhip_rrt_batch* b, uint32_t problem, double* states_out, int32_t* parents_out,
                                      uint32_t cap_nodes, uint32_t* n_nodes) {
    if (!b || !n_nodes) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (b->cfg.planner != OXHIP_PLANNER_RRT_CONNECT) return fail(OXHIP_ERR_BAD_ARG, "not an RRTConnect batch");
    if (problem >= b->cfg.n_problems) return fail(OXHIP_ERR_BAD_ARG, "problem index out of range");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    const uint32_t n = states[problem].n_nodes_b, dim = b->cfg.dim;
    *n_nodes = n;
    if (n > cap_nodes || (!states_out && !parents_out)) return n > cap_nodes ? fail(OXHIP_ERR_CAPACITY, "tree buffer too small") : OXHIP_OK;
    std::vector<double> soa;
    std::vector<int32_t> par;
    if ((st = fetch_tree(b, problem, n, soa, par, true)) != OXHIP_OK) return st;
    if (states_out)
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = 0; k < dim; ++k) states_out[(size_t)i * dim + k] = soa[(size_t)k * n + i];
    if (parents_out) std::memcpy(parents_out, par.data(), (size_t)n * sizeof(int32_t));
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_costs(oxhip_rrt_batch* b, uint32_t problem, double* costs, uint32_t cap_nodes, uint32_t* n_nodes) {
    if (!b || !n_nodes) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->is_setup) return fail(OXHIP_ERR_PLANNER_UNINITIALISED, "setup() was not called");
    if (b->cfg.planner != OXHIP_PLANNER_RRT_STAR) return fail(OXHIP_ERR_BAD_ARG, "not an RRT* batch");
    if (problem >= b->cfg.n_problems) return fail(OXHIP_ERR_BAD_ARG, "problem index out of range");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    std::vector<ProblemState> states;
    if ((st = read_states(b, states)) != OXHIP_OK) return st;
    const uint32_t n = states[problem].n_nodes;
    *n_nodes = n;
    if (!costs) return OXHIP_OK;
    if (n > cap_nodes) return fail(OXHIP_ERR_CAPACITY, "cost buffer too small");
    HIP_TRY(hipMemcpyAsync(costs, b->cost.p + (size_t)problem * b->dp.cap, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_last_timing(oxhip_rrt_batch* b, double* kernel_ms, uint32_t* launches, uint32_t* kernel_kind) {
    if (!b) return fail(OXHIP_ERR_BAD_ARG, "null batch");
    if (kernel_ms) *kernel_ms = b->last_kernel_ms;
    if (launches) *launches = b->last_launches;
    if (kernel_kind) *kernel_kind = b->cfg.planner == OXHIP_PLANNER_RRT ? b->last_kind : b->kernel_kind;
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_enable_stamps(oxhip_rrt_batch* b, uint32_t enable) {
    if (!b) return fail(OXHIP_ERR_BAD_ARG, "null batch");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    if (enable) {
        HIP_TRY(b->dbg.alloc(OXHIP_STAMP_WORDS));
        HIP_TRY(hipMemsetAsync(b->dbg.p, 0, OXHIP_STAMP_WORDS * sizeof(uint64_t), b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        b->dp.dbg = b->dbg.p;
    } else {
        b->dp.dbg = nullptr;
    }
    return OXHIP_OK;
}

int32_t oxhip_rrt_batch_get_stamps(oxhip_rrt_batch* b, uint64_t* out, uint32_t cap_words) {
    if (!b || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (!b->dp.dbg) return fail(OXHIP_ERR_BAD_ARG, "stamps are not enabled");
    int32_t st = select_device(b->cfg.device);
    if (st != OXHIP_OK) return st;
    const uint32_t n = cap_words < OXHIP_STAMP_WORDS ? cap_words : OXHIP_STAMP_WORDS;
    if (n) HIP_TRY(hipMemcpyAsync(out, b->dbg.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return OXHIP_OK;
}

// ------------------------------------------------------------------ stand-alone primitives

int32_t oxhip_nn_argmin_batch(int32_t device, uint32_t dim, const double* nodes, const uint32_t* n_nodes,
                              uint32_t n_queries, const double* queries, uint32_t* out_index, double* out_min_dist) {
    if (!nodes || !n_nodes || !queries || !out_index || !out_min_dist) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (dim == 0 || dim > OXHIP_MAX_DIM) return fail(OXHIP_ERR_BAD_ARG, "dim must be in 1..8");
    if (n_queries == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    std::vector<uint64_t> offsets(n_queries);
    uint64_t total = 0;
    for (uint32_t q = 0; q < n_queries; ++q) {
        if (n_nodes[q] == 0) return fail(OXHIP_ERR_BAD_ARG, "every tree needs at least its root (rrt.rs:188 reads tree[0])");
        offsets[q] = total;
        total += n_nodes[q];
    }
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<double> d_nodes, d_q, d_dist;
    DevBuf<uint64_t> d_off;
    DevBuf<uint32_t> d_n, d_idx;
    OX_TRY(to_device(d_nodes, nodes, (size_t)total * dim, ts.s));
    OX_TRY(to_device(d_q, queries, (size_t)n_queries * dim, ts.s));
    OX_TRY(to_device(d_off, offsets.data(), n_queries, ts.s));
    OX_TRY(to_device(d_n, n_nodes, n_queries, ts.s));
    HIP_TRY(d_idx.alloc(n_queries));
    HIP_TRY(d_dist.alloc(n_queries));
    launch_nn_argmin(dim, d_nodes.p, d_off.p, d_n.p, n_queries, d_q.p, d_idx.p, d_dist.p, ts.s);
    HIP_TRY(hipGetLastError());
    OX_TRY(to_host(out_index, d_idx, n_queries, ts.s));
    OX_TRY(to_host(out_min_dist, d_dist, n_queries, ts.s));
    return OXHIP_OK;
}

int32_t oxhip_distance_batch(int32_t device, uint32_t dim, const double* a, const double* b, uint32_t n, double* out) {
    if (!a || !b || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (dim == 0 || dim > OXHIP_MAX_DIM) return fail(OXHIP_ERR_BAD_ARG, "dim must be in 1..8");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<double> da, db, dout;
    OX_TRY(to_device(da, a, (size_t)n * dim, ts.s));
    OX_TRY(to_device(db, b, (size_t)n * dim, ts.s));
    HIP_TRY(dout.alloc(n));
    launch_distance(dim, da.p, db.p, n, dout.p, ts.s);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, n, ts.s);
}

int32_t oxhip_interpolate_batch(int32_t device, uint32_t dim, const double* from, const double* to, const double* t,
                                uint32_t n, double* out) {
    if (!from || !to || !t || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (dim == 0 || dim > OXHIP_MAX_DIM) return fail(OXHIP_ERR_BAD_ARG, "dim must be in 1..8");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<double> da, db, dt, dout;
    OX_TRY(to_device(da, from, (size_t)n * dim, ts.s));
    OX_TRY(to_device(db, to, (size_t)n * dim, ts.s));
    OX_TRY(to_device(dt, t, n, ts.s));
    HIP_TRY(dout.alloc((size_t)n * dim));
    launch_interpolate(dim, da.p, db.p, dt.p, n, dout.p, ts.s);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, (size_t)n * dim, ts.s);
}

int32_t oxhip_rrt_batch_is_valid(oxhip_rrt_batch* b, const double* states, uint32_t n, uint8_t* out) {
    if (!b || !states || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(b->cfg.device));
    DevBuf<double> ds;
    DevBuf<uint8_t> dout;
    OX_TRY(to_device(ds, states, (size_t)n * b->cfg.dim, b->stream));
    HIP_TRY(dout.alloc(n));
    if (b->cfg.space == OXHIP_SPACE_SE2) launch_se2_is_valid(b->dp, ds.p, n, dout.p, b->stream);
    else launch_is_valid(b->dp, ds.p, n, dout.p, b->stream);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, n, b->stream);
}

int32_t oxhip_rrt_batch_check_motion(oxhip_rrt_batch* b, const double* from, const double* to, uint32_t n, uint8_t* out) {
    if (!b || !from || !to || !out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(b->cfg.device));
    DevBuf<double> da, db;
    DevBuf<uint8_t> dout;
    OX_TRY(to_device(da, from, (size_t)n * b->cfg.dim, b->stream));
    OX_TRY(to_device(db, to, (size_t)n * b->cfg.dim, b->stream));
    HIP_TRY(dout.alloc(n));
    if (b->cfg.space == OXHIP_SPACE_SE2) launch_se2_check_motion(b->dp, da.p, db.p, n, dout.p, b->stream);
    else launch_check_motion(b->dp, da.p, db.p, n, dout.p, b->stream);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, n, b->stream);
}

int32_t oxhip_f64_op_batch(int32_t device, uint32_t op, const double* a, const double* b, const double* c, uint32_t n,
                           double* out) {
    if (!a || !out || op > 6) return fail(OXHIP_ERR_BAD_ARG, "bad argument");
    if ((op == 1 || op == 3 || op == 4) && !b) return fail(OXHIP_ERR_BAD_ARG, "operand b required");
    if (op == 3 && !c) return fail(OXHIP_ERR_BAD_ARG, "operand c required");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<double> da, db, dc, dout;
    OX_TRY(to_device(da, a, n, ts.s));
    if (b) OX_TRY(to_device(db, b, n, ts.s));
    if (c) OX_TRY(to_device(dc, c, n, ts.s));
    HIP_TRY(dout.alloc(n));
    launch_f64_op(op, da.p, db.p, dc.p, n, dout.p, ts.s);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, n, ts.s);
}

int32_t oxhip_se2_op_batch(int32_t device, uint32_t op, const double* a, const double* b, const double* t, uint32_t n,
                           double* out) {
    if (!a || !b || !out || op > 1 || (op == 1 && !t)) return fail(OXHIP_ERR_BAD_ARG, "bad argument");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<double> da, db, dt, dout;
    OX_TRY(to_device(da, a, (size_t)3 * n, ts.s));
    OX_TRY(to_device(db, b, (size_t)3 * n, ts.s));
    if (t) OX_TRY(to_device(dt, t, n, ts.s));
    HIP_TRY(dout.alloc((size_t)3 * n));
    launch_se2_op(op, da.p, db.p, dt.p, n, dout.p, ts.s);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, (size_t)3 * n, ts.s);
}

int32_t oxhip_rng_u64_batch(int32_t device, uint64_t seed, uint64_t stream, uint32_t n, uint64_t* out) {
    if (!out) return fail(OXHIP_ERR_BAD_ARG, "null argument");
    if (n == 0) return OXHIP_OK;
    OX_TRY(select_device(device));
    TmpStream ts;
    HIP_TRY(oxhip_stream_acquire(device, &ts.s));
    ts.device = device;
    DevBuf<uint64_t> dout;
    HIP_TRY(dout.alloc(n));
    launch_rng_u64(seed, stream, n, dout.p, ts.s);
    HIP_TRY(hipGetLastError());
    return to_host(out, dout, n, ts.s);
}

}  // extern "C"
