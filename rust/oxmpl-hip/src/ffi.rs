//! `extern "C"` surface of `liboxmpl_hip.so`, one to one with `include/oxmpl_hip.h`.
//!
//! The layout tables below are data for two checks: `cargo test` compares them with this file's `#[repr(C)]`
//! structs (`offset_of!`), and the repository's CPU test `tests/test_rust_ffi_layout.py` compares them with
//! `offsetof` / `sizeof` as printed by a C program that includes the header, and with the ctypes structures
//! of `oxmpl_amd/capi.py`.  A field added to the header without touching this file fails that test.
#![allow(non_camel_case_types)]

use std::os::raw::c_char;

pub const OXHIP_ABI_VERSION: i32 = 2;
pub const OXHIP_MAX_DIM: usize = 8;

// oxhip_status (mirrors PlanningError, oxmpl/src/base/error.rs:97-108, plus the conditions the reference panics on)
pub const OXHIP_OK: i32 = 0;
pub const OXHIP_ERR_TIMEOUT: i32 = 1;
pub const OXHIP_ERR_NO_SOLUTION_FOUND: i32 = 2;
pub const OXHIP_ERR_PLANNER_UNINITIALISED: i32 = 3;
pub const OXHIP_ERR_INVALID_START_STATE: i32 = 4;
pub const OXHIP_ERR_UNSAMPLED_STATE_SPACE: i32 = 5;
pub const OXHIP_ERR_BAD_ARG: i32 = 16;
pub const OXHIP_ERR_UNBOUNDED: i32 = 17;
pub const OXHIP_ERR_ZERO_VOLUME: i32 = 18;
pub const OXHIP_ERR_CAPACITY: i32 = 19;
pub const OXHIP_ERR_HIP: i32 = 32;
pub const OXHIP_ERR_NO_DEVICE: i32 = 33;

// oxhip_planner_kind / oxhip_kernel_kind / oxhip_space_kind
pub const OXHIP_PLANNER_RRT: u32 = 0;
pub const OXHIP_PLANNER_RRT_CONNECT: u32 = 1;
pub const OXHIP_PLANNER_RRT_STAR: u32 = 2;
pub const OXHIP_KERNEL_AUTO: u32 = 0;
pub const OXHIP_SPACE_REAL_VECTOR: u32 = 0;
pub const OXHIP_SPACE_SE2: u32 = 1;
// oxhip_goal_sampler: what GoalSampleableRegion::sample_goal draws (goal.rs:35-41)
pub const OXHIP_GOAL_SAMPLE_CENTRE: u32 = 0;
pub const OXHIP_GOAL_SAMPLE_UNIFORM_DISC: u32 = 1;

/// `oxhip_rrt_config` (include/oxmpl_hip.h): RRT::new (rrt.rs:75-83) + RealVectorStateSpace::new (rvss.rs:65-100)
/// + the deterministic termination the reference lacks (rrt.rs:226).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OxhipRrtConfig {
    pub struct_size: u32,
    pub dim: u32,
    pub bounds: [f64; 2 * OXHIP_MAX_DIM],
    pub max_distance: f64,
    pub goal_bias: f64,
    pub lvs_fraction: f64,
    pub n_problems: u32,
    pub max_nodes: u32,
    pub stop_at_goal: u32,
    pub kernel: u32,
    pub seed: u64,
    pub first_problem_id: u64,
    pub device: i32,
    pub planner: u32,
    pub search_radius: f64,
    pub space: u32,
    pub goal_sampler: u32,
    pub debug_flags: u32,
    pub star_pool_share: u32,
    pub frozen_split: u32,
    pub reserved: u32,
}

/// (field, byte offset, byte size) of `oxhip_rrt_config`
pub const OXHIP_RRT_CONFIG_LAYOUT: &[(&str, usize, usize)] = &[
    ("struct_size", 0, 4),
    ("dim", 4, 4),
    ("bounds", 8, 128),
    ("max_distance", 136, 8),
    ("goal_bias", 144, 8),
    ("lvs_fraction", 152, 8),
    ("n_problems", 160, 4),
    ("max_nodes", 164, 4),
    ("stop_at_goal", 168, 4),
    ("kernel", 172, 4),
    ("seed", 176, 8),
    ("first_problem_id", 184, 8),
    ("device", 192, 4),
    ("planner", 196, 4),
    ("search_radius", 200, 8),
    ("space", 208, 4),
    ("goal_sampler", 212, 4),
    ("debug_flags", 216, 4),
    ("star_pool_share", 220, 4),
    ("frozen_split", 224, 4),
    ("reserved", 228, 4),
];
pub const OXHIP_RRT_CONFIG_SIZE: usize = 232;

/// `oxhip_prm_config` (include/oxmpl_hip.h): PRM::new(timeout, connection_radius) (prm.rs:70-78) + the space.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OxhipPrmConfig {
    pub struct_size: u32,
    pub dim: u32,
    pub bounds: [f64; 2 * OXHIP_MAX_DIM],
    pub timeout: f64,
    pub connection_radius: f64,
    pub lvs_fraction: f64,
    pub max_milestones: u32,
    pub device: i32,
    pub max_samples: u64,
    pub seed: u64,
    pub stream: u64,
    pub knn_k: u32,
    pub reserved: u32,
}

/// (field, byte offset, byte size) of `oxhip_prm_config`
pub const OXHIP_PRM_CONFIG_LAYOUT: &[(&str, usize, usize)] = &[
    ("struct_size", 0, 4),
    ("dim", 4, 4),
    ("bounds", 8, 128),
    ("timeout", 136, 8),
    ("connection_radius", 144, 8),
    ("lvs_fraction", 152, 8),
    ("max_milestones", 160, 4),
    ("device", 164, 4),
    ("max_samples", 168, 8),
    ("seed", 176, 8),
    ("stream", 184, 8),
    ("knn_k", 192, 4),
    ("reserved", 196, 4),
];
pub const OXHIP_PRM_CONFIG_SIZE: usize = 200;

/// opaque handles (owned by the library; freed with the matching `_destroy`)
#[repr(C)]
pub struct OxhipRrtBatch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct OxhipPrm {
    _private: [u8; 0],
}

extern "C" {
    // ---- library
    pub fn oxhip_abi_version() -> i32;
    pub fn oxhip_status_string(status: i32) -> *const c_char;
    pub fn oxhip_last_error_string() -> *const c_char;
    pub fn oxhip_device_count(count: *mut i32) -> i32;

    // ---- batched tree planners: RRT::new / Planner::setup / Planner::solve (rrt.rs, rrt_connect.rs, rrt_star.rs)
    pub fn oxhip_rrt_batch_create(cfg: *const OxhipRrtConfig, out: *mut *mut OxhipRrtBatch) -> i32;
    pub fn oxhip_rrt_batch_destroy(b: *mut OxhipRrtBatch) -> i32;
    pub fn oxhip_rrt_batch_set_spheres(b: *mut OxhipRrtBatch, centres: *const f64, radii: *const f64, n: u32) -> i32;
    pub fn oxhip_rrt_batch_set_boxes(b: *mut OxhipRrtBatch, lo: *const f64, hi: *const f64, n: u32) -> i32;
    pub fn oxhip_rrt_batch_set_segments(b: *mut OxhipRrtBatch, segments: *const f64, n: u32, clearance: f64) -> i32;
    pub fn oxhip_rrt_batch_setup(b: *mut OxhipRrtBatch, starts: *const f64, goal_centres: *const f64, goal_radii: *const f64) -> i32;
    pub fn oxhip_rrt_batch_set_tree(b: *mut OxhipRrtBatch, problem: u32, states: *const f64, parents: *const i32, n_nodes: u32) -> i32;
    pub fn oxhip_rrt_batch_solve(b: *mut OxhipRrtBatch, max_iterations: u64, timeout_s: f64, freeze: u32, status_out: *mut i32) -> i32;
    pub fn oxhip_rrt_batch_get_counts(
        b: *mut OxhipRrtBatch,
        iterations: *mut u64,
        nodes: *mut u32,
        accepted: *mut u64,
        checksum: *mut u64,
        goal_node: *mut i32,
        stop_reason: *mut i32,
    ) -> i32;
    pub fn oxhip_rrt_batch_get_tree(b: *mut OxhipRrtBatch, problem: u32, states: *mut f64, parents: *mut i32, cap_nodes: u32, n_nodes: *mut u32) -> i32;
    pub fn oxhip_rrt_batch_get_path(b: *mut OxhipRrtBatch, problem: u32, states: *mut f64, cap_states: u32, len: *mut u32) -> i32;
    pub fn oxhip_rrt_batch_get_goal_counts(b: *mut OxhipRrtBatch, nodes: *mut u32, end_node: *mut i32) -> i32;
    pub fn oxhip_rrt_batch_get_goal_tree(b: *mut OxhipRrtBatch, problem: u32, states: *mut f64, parents: *mut i32, cap_nodes: u32, n_nodes: *mut u32) -> i32;
    pub fn oxhip_rrt_batch_get_costs(b: *mut OxhipRrtBatch, problem: u32, costs: *mut f64, cap_nodes: u32, n_nodes: *mut u32) -> i32;
    pub fn oxhip_rrt_batch_last_timing(b: *mut OxhipRrtBatch, kernel_ms: *mut f64, launches: *mut u32, kernel_kind: *mut u32) -> i32;
    pub fn oxhip_rrt_batch_is_valid(b: *mut OxhipRrtBatch, states: *const f64, n: u32, out: *mut u8) -> i32;
    pub fn oxhip_rrt_batch_check_motion(b: *mut OxhipRrtBatch, from: *const f64, to: *const f64, n: u32, out: *mut u8) -> i32;

    // ---- stand-alone pieces of the path (rrt.rs:187-196, rvss.rs:137-155, rvss.rs:161-186)
    pub fn oxhip_nn_argmin_batch(
        device: i32,
        dim: u32,
        nodes: *const f64,
        n_nodes: *const u32,
        n_queries: u32,
        queries: *const f64,
        out_index: *mut u32,
        out_min_dist: *mut f64,
    ) -> i32;
    pub fn oxhip_distance_batch(device: i32, dim: u32, a: *const f64, b: *const f64, n: u32, out: *mut f64) -> i32;
    pub fn oxhip_interpolate_batch(device: i32, dim: u32, from: *const f64, to: *const f64, t: *const f64, n: u32, out: *mut f64) -> i32;

    // ---- PRM (prm.rs)
    pub fn oxhip_prm_create(cfg: *const OxhipPrmConfig, out: *mut *mut OxhipPrm) -> i32;
    pub fn oxhip_prm_destroy(p: *mut OxhipPrm) -> i32;
    pub fn oxhip_prm_set_spheres(p: *mut OxhipPrm, centres: *const f64, radii: *const f64, n: u32) -> i32;
    pub fn oxhip_prm_set_boxes(p: *mut OxhipPrm, lo: *const f64, hi: *const f64, n: u32) -> i32;
    pub fn oxhip_prm_setup(p: *mut OxhipPrm, start: *const f64, goal_centre: *const f64, goal_radius: f64) -> i32;
    pub fn oxhip_prm_set_problem(p: *mut OxhipPrm, start: *const f64, goal_centre: *const f64, goal_radius: f64) -> i32;
    pub fn oxhip_prm_construct_roadmap(p: *mut OxhipPrm) -> i32;
    pub fn oxhip_prm_get_sizes(p: *mut OxhipPrm, n_milestones: *mut u32, n_edge_entries: *mut u64, n_samples: *mut u64) -> i32;
    pub fn oxhip_prm_get_roadmap(p: *mut OxhipPrm, states: *mut f64, cap_nodes: u32, offsets: *mut u64, neighbours: *mut u32, cap_entries: u64) -> i32;
    pub fn oxhip_prm_solve(p: *mut OxhipPrm, timeout_s: f64, path: *mut f64, cap_states: u32, len: *mut u32) -> i32;
    pub fn oxhip_prm_get_query_sets(
        p: *mut OxhipPrm,
        start_connections: *mut u32,
        cap_start: u32,
        n_start: *mut u32,
        goal_indices: *mut u32,
        cap_goal: u32,
        n_goal: *mut u32,
    ) -> i32;
    pub fn oxhip_prm_last_timing(p: *mut OxhipPrm, phase_ms: *mut f64, n_candidates: *mut u64, redraw_batches: *mut u32) -> i32;
    pub fn oxhip_prm_knn_exact_rows(p: *mut OxhipPrm, rows: *mut u32) -> i32;
}

#[cfg(test)]
mod tests {
    use super::*;
    use std::mem::{offset_of, size_of};

    macro_rules! check_layout {
        ($ty:ty, $table:expr, $size:expr, [$($field:ident),* $(,)?]) => {{
            assert_eq!(size_of::<$ty>(), $size);
            let mut i = 0;
            $(
                let (name, off, _size) = $table[i];
                assert_eq!(name, stringify!($field));
                assert_eq!(offset_of!($ty, $field), off, "offset of {}", name);
                i += 1;
            )*
            assert_eq!(i, $table.len());
        }};
    }

    #[test]
    fn repr_c_structs_match_the_layout_tables() {
        check_layout!(OxhipRrtConfig, OXHIP_RRT_CONFIG_LAYOUT, OXHIP_RRT_CONFIG_SIZE,
            [struct_size, dim, bounds, max_distance, goal_bias, lvs_fraction, n_problems, max_nodes, stop_at_goal, kernel,
             seed, first_problem_id, device, planner, search_radius, space, goal_sampler, debug_flags, star_pool_share, frozen_split, reserved]);
        check_layout!(OxhipPrmConfig, OXHIP_PRM_CONFIG_LAYOUT, OXHIP_PRM_CONFIG_SIZE,
            [struct_size, dim, bounds, timeout, connection_radius, lvs_fraction, max_milestones, device, max_samples, seed, stream, knn_k, reserved]);
    }
}
