// oxhip_internal.hpp -- launch wrappers shared between the kernel translation units and the
// C-ABI implementation (oxhip_api.hip).  Not installed; the public surface is include/oxmpl_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oxhip {

struct DevParams;

// rrt_stream.hip
void launch_rrt_stream(const DevParams& p, hipStream_t stream);
void launch_nn_argmin(uint32_t dim, const double* nodes, const uint64_t* offsets, const uint32_t* n_nodes,
                      uint32_t n_queries, const double* queries, uint32_t* out_index, double* out_min_dist,
                      hipStream_t stream);
void launch_distance(uint32_t dim, const double* a, const double* b, uint32_t n, double* out, hipStream_t s);
void launch_interpolate(uint32_t dim, const double* from, const double* to, const double* t, uint32_t n,
                        double* out, hipStream_t s);
void launch_is_valid(const DevParams& p, const double* states, uint32_t n, uint8_t* out, hipStream_t s);
void launch_check_motion(const DevParams& p, const double* from, const double* to, uint32_t n, uint8_t* out,
                         hipStream_t s);
void launch_f64_op(uint32_t op, const double* a, const double* b, const double* c, uint32_t n, double* out,
                   hipStream_t s);
void launch_rng_u64(uint64_t seed, uint64_t stream, uint32_t n, uint64_t* out, hipStream_t s);

// rrt_connect.hip: RRTConnect (rrt_connect.rs), one 256-thread workgroup per problem, trees streamed from HBM/L2
void launch_rrt_connect(const DevParams& p, hipStream_t stream);

// rrt_connect_se2.hip: RRTConnect over SE(2) among line segments (BASELINE.json configs[3]), one wave per problem
void launch_rrt_connect_se2(const DevParams& p, hipStream_t stream);
uint32_t seg_grid_side();                                                 // cells along an axis of DevParams::seg_grid
void launch_seg_grid(const DevParams& p, uint16_t* grid, double clearance, hipStream_t stream);   // (re)builds it from segs
void launch_se2_op(uint32_t op, const double* a, const double* b, const double* t, uint32_t n, double* out, hipStream_t s);
void launch_se2_is_valid(const DevParams& p, const double* states, uint32_t n, uint8_t* out, hipStream_t s);
void launch_se2_check_motion(const DevParams& p, const double* from, const double* to, uint32_t n, uint8_t* out, hipStream_t s);

// rrt_star.hip: RRT* (rrt_star.rs), one 256-thread workgroup per problem
void launch_rrt_star(const DevParams& p, hipStream_t stream);

// rrt_star_wire.hip: the wiring stages of the decoupled RRT* (the geometry comes from launch_rrt_lanes)
bool star_wire_supported(uint32_t dim);
void launch_star_shadow(const DevParams& p, uint32_t max_nodes, hipStream_t stream);   // tree32 and its magnitude bound, nodes [0, n)
void launch_star_count(const DevParams& p, uint32_t max_pending, hipStream_t stream);   // nbr_cnt of the nodes [wired, n)
void launch_star_scan(const DevParams& p, hipStream_t stream);                          // nbr_off, nbr_take
void launch_star_fill(const DevParams& p, uint32_t max_take, hipStream_t stream);       // the lists of [wired, wired + take)
void launch_star_edges(const DevParams& p, uint32_t max_total, hipStream_t stream);     // per pair: distance, both motions' validity
void launch_star_compact(const DevParams& p, uint32_t max_chunks, hipStream_t stream);  // the counting pass's chunks -> the lists (instead of launch_star_fill)
void launch_star_wire(const DevParams& p, hipStream_t stream);                          // choose parent / rewire, node by node

// rrt_resident.hip
// true when a register-resident instantiation exists for (dim, cap)
bool resident_supported(uint32_t dim, uint32_t cap);
void launch_rrt_resident(const DevParams& p, hipStream_t stream);

// rrt_lanes.hip: the resident pipeline with a lane-per-query resolver (64 iterations resolved side by side)
bool lanes_supported(uint32_t dim, uint32_t cap);
void launch_rrt_lanes(const DevParams& p, hipStream_t stream);

// rrt_cells.hip: one wave per problem, nearest neighbour through an exact cell grid (R^2, R^3)
bool cells_supported(uint32_t dim, uint32_t cap);
uint32_t cells_level_max(uint32_t dim, uint32_t cap);
uint32_t cells_head_blocks(uint32_t dim, uint32_t cap);
void launch_rrt_cells(const DevParams& p, hipStream_t stream);
uint32_t sphere_grid_side(uint32_t dim);                                  // cells along an axis of DevParams::sph_grid
void launch_sphere_grid(const DevParams& p, uint64_t* grid, const double* filt, hipStream_t stream);   // (re)builds it from sph_c and the filter thresholds

// prm_kernels.hip: PRM roadmap construction / query (prm.rs)
struct PrmState {            // persists in HBM between launches
    uint64_t draws;          // u64 words consumed from the ChaCha12 stream
    uint64_t n_samples;      // sample_uniform calls made (prm.rs:122)
    unsigned long long n_cand;  // in-radius pairs counted by the current pairs launch (beyond cand_cap: not stored)
    uint32_t n_milestones;   // roadmap.len()
    uint32_t n_keys;         // directed edge keys appended so far (2 per undirected edge)
    uint32_t redraw_batches; // sample batches replayed sequentially because rand rejected a draw
    uint32_t pad;
};
struct PrmArgs {
    double* ms;              // milestones, AoS [cap][dim]
    float* ms32;             // fl32(ms), same layout: what the pair search screens with (its i side is read by scalar loads)
    uint32_t cap;
    uint32_t n_target;       // sample until the roadmap holds this many milestones ...
    uint64_t max_samples;    // ... or this many samples were drawn
    uint64_t stream;         // ChaCha12 stream id (the key is DevParams::seed)
    PrmState* state;
    uint2* cand;             // (j, i) with i < j and distance < connection_radius
    uint32_t cand_cap;
    uint64_t* keys;          // (u << shift) | v for every directed edge u -> v, shift = bits of (cap - 1)
};
struct PrmSpec {              // one round of the parallel sampler
    uint64_t pos0;           // stream word of the round's first sample (= PrmState::draws)
    uint32_t m;              // samples drawn this round
    uint32_t pad;
    double* tmp;             // [m][dim] the round's samples
    uint64_t* vbits;         // [ceil(m/64)] validity ballots
    uint32_t* wave_off;      // [ceil(m/64)] per-wave valid counts, then their exclusive prefix
    uint32_t* redraw_flag;   // set when rand's range sampler would have rejected a draw: the round is replayed
    PrmState* result;        // the state after this round (valid when redraw_flag stays 0)
};
struct PrmQuery {
    double start[8], goal_c[8];
    double goal_thr;         // satisfied iff d2 <= goal_thr
};
void launch_prm_sample(const DevParams& p, const PrmArgs& a, hipStream_t s);
// speculative parallel round: draws, scans, compacts; the host commits sp.result when no draw was rejected
void launch_prm_sample_spec(const DevParams& p, const PrmArgs& a, const PrmSpec& sp, uint32_t n0, hipStream_t s);
// thr_row / thr32_row (optional, [cap]): a threshold per row j instead of `thr` (the k-nearest variant's candidate search)
void launch_prm_pairs(const DevParams& p, const PrmArgs& a, uint32_t j0, uint32_t j1, double thr, hipStream_t s, const double* thr_row = nullptr,
                      const float* thr32_row = nullptr);
float prm_screen_threshold(const DevParams& p, double thr);   // the binary32 screen's threshold for a binary64 d2 threshold
// k-nearest variant: candidates -> sortable keys; sorted keys -> each row's k nearest (`sel`, counters[0] pairs; rows whose radius held
// too few in failed_rows, counters[1] of them); the exact search for those rows
void launch_prm_knn_keys(const PrmArgs& a, uint32_t n_cand, uint64_t* keys, hipStream_t s);
void launch_prm_knn_select(const DevParams& p, const PrmArgs& a, const uint64_t* sorted, double* dist, uint32_t n_sorted, uint32_t j0, uint32_t j1,
                           uint32_t k, uint2* sel, uint32_t* counters, uint32_t* failed_rows, hipStream_t s);
void launch_prm_knn_brute(const DevParams& p, const PrmArgs& a, const uint32_t* failed_rows, uint32_t n_failed, uint32_t k, uint2* sel,
                          uint32_t* counters, hipStream_t s);
void launch_prm_edges(const DevParams& p, const PrmArgs& a, uint32_t n_cand, hipStream_t s);
// rocPRIM radix sort of the directed keys; tmp == nullptr queries tmp_bytes
hipError_t prm_sort_keys(void* tmp, size_t& tmp_bytes, uint64_t* in, uint64_t* out, uint32_t n_keys, uint32_t cap,
                         hipStream_t s);
void launch_prm_csr(const uint64_t* sorted, uint32_t n_keys, uint32_t n_nodes, uint32_t cap, uint32_t* offsets,
                    uint32_t* nbrs, hipStream_t s);
void launch_prm_query(const DevParams& p, const PrmArgs& a, uint32_t n, const PrmQuery& q, double thr, uint8_t* flags,
                      uint32_t* start_valid, hipStream_t s);

}  // namespace oxhip
