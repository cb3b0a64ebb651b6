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

// rrt_resident.hip
// true when a register-resident instantiation exists for (dim, cap)
bool resident_supported(uint32_t dim, uint32_t cap);
void launch_rrt_resident(const DevParams& p, hipStream_t stream);

// rrt_pruned.hip: the resident pipeline + launch-time spatial sort and box-pruned scans
bool pruned_supported(uint32_t dim, uint32_t cap);
void launch_rrt_pruned(const DevParams& p, hipStream_t stream);

}  // namespace oxhip
