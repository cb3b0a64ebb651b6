/*
 * oxmpl_hip.h -- C ABI of liboxmpl_hip.so: the MI355X (gfx950) batched RRT hot path.
 *
 * The reference (rossng/oxmpl, Rust) has no C ABI; the boundary of this path is the
 * trait `Planner<S, SP, G>` (oxmpl/src/base/planner.rs:26-60) over `StateSpace`
 * (space.rs:78-145), `StateValidityChecker` (validity.rs:39-48) and `Goal*`
 * (goal.rs:12-41).  The entry points below are what a Rust `extern "C"` block (or
 * cgo / ctypes) binds to put the HIP path behind those traits; each cites the
 * reference item it replaces.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions: plain pointers + sizes, caller owns every host array (copied during
 * the call), all states are f64, AoS `[n][dim]` on this boundary (SoA on the device).
 * Every function returns an oxhip_status; nothing aborts or throws across the ABI.
 * A handle is bound to one HIP device and one stream and is not re-entrant;
 * different handles may be driven from different host threads / processes.
 */
#ifndef OXMPL_HIP_H
#define OXMPL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: get_stamps takes a capacity (ABI 1 wrote 32, later 64 words unasked); kernel kinds 3 / 4 retired; the per-iteration checksum is
 *    the polynomial H <- H P + g (ABI 1 as first shipped chained FNV-1a); oxhip_rrt_config gained goal_sampler / debug_flags */
#define OXHIP_ABI_VERSION 2
#define OXHIP_MAX_DIM 8

/* Status codes.  0-5 mirror `PlanningError` (oxmpl/src/base/error.rs:97-108) in
 * declaration order (+1); 16-18 mirror the sampling/space errors the reference
 * would panic on via unwrap() (rrt.rs:180,183; real_vector_state_space.rs:239-244,
 * 78-83); 32+ are HIP/runtime conditions that have no reference counterpart. */
typedef enum oxhip_status {
    OXHIP_OK = 0,                       /* Ok(Path) / success */
    OXHIP_ERR_TIMEOUT = 1,              /* PlanningError::Timeout            rrt.rs:172-174 */
    OXHIP_ERR_NO_SOLUTION_FOUND = 2,    /* PlanningError::NoSolutionFound    rrt.rs:226 (iteration / node cap) */
    OXHIP_ERR_PLANNER_UNINITIALISED = 3,/* PlanningError::PlannerUninitialised rrt.rs:160-163 */
    OXHIP_ERR_INVALID_START_STATE = 4,  /* PlanningError::InvalidStartState  (unused by RRT, kept for parity of the enum) */
    OXHIP_ERR_UNSAMPLED_STATE_SPACE = 5,/* PlanningError::UnsampledStateSpace (PRM only) */
    OXHIP_ERR_BAD_ARG = 16,             /* null pointer, dim out of range, goal_bias outside [0,1], ... */
    OXHIP_ERR_UNBOUNDED = 17,           /* StateSamplingError::UnboundedDimension  rvss.rs:239-241 */
    OXHIP_ERR_ZERO_VOLUME = 18,         /* StateSamplingError::ZeroVolume / StateSpaceError::InvalidBound */
    OXHIP_ERR_CAPACITY = 19,            /* caller buffer too small; required length is still written */
    OXHIP_ERR_HIP = 32,                 /* a HIP runtime call failed; see oxhip_last_error_string() */
    OXHIP_ERR_NO_DEVICE = 33            /* no gfx950 device visible -- there is NO CPU fallback */
} oxhip_status;

/* why a problem stopped (per-problem, next to its status) */
typedef enum oxhip_stop_reason {
    OXHIP_STOP_NONE = -1,
    OXHIP_STOP_GOAL = 0,        /* goal.is_satisfied(q_new) with stop_at_goal      rrt.rs:220-223 */
    OXHIP_STOP_ITERATIONS = 1,  /* iteration budget of this solve call exhausted */
    OXHIP_STOP_NODES = 2,       /* tree reached max_nodes */
    OXHIP_STOP_TIMEOUT = 3,     /* wall-clock timeout between kernel chunks */
    OXHIP_STOP_INTERNAL = 4     /* kernel-internal hand-off never completed (bug guard); solve returns OXHIP_ERR_HIP */
} oxhip_stop_reason;

typedef enum oxhip_planner_kind {
    OXHIP_PLANNER_RRT = 0,         /* geometric::RRT         oxmpl/src/geometric/planners/rrt.rs */
    OXHIP_PLANNER_RRT_CONNECT = 1, /* geometric::RRTConnect  oxmpl/src/geometric/planners/rrt_connect.rs (stream kernel only) */
    OXHIP_PLANNER_RRT_STAR = 2     /* geometric::RRTStar     oxmpl/src/geometric/planners/rrt_star.rs.  kernel = OXHIP_KERNEL_AUTO /
                                      OXHIP_KERNEL_LANES: the decoupled design (the node positions of an RRT* run are RRT's, so the
                                      lane-per-query kernel grows the tree and rrt_star_wire.hip then chooses parents and rewires:
                                      R^2 .. R^6, trees that fit the register rows); OXHIP_KERNEL_STREAM: rrt_star.hip, everything
                                      in one kernel (any dimension <= 8, any size).  Same results bit for bit. */
} oxhip_planner_kind;

/* State space of a batch.  oxmpl has RealVectorStateSpace, SO2StateSpace and SO3StateSpace; SE(2) is not in the
 * reference (docs/BACKLOG.md:12-14) and is assembled here from the first two (see rrt_connect_se2.hip):
 * state (x, y, theta), distance = 1.0 * d_xy + 0.5 * d_theta, extent = extent_xy + 0.5 * PI. */
typedef enum oxhip_space_kind {
    OXHIP_SPACE_REAL_VECTOR = 0,  /* RealVectorStateSpace(dim)  oxmpl/src/base/spaces/real_vector_state_space.rs */
    OXHIP_SPACE_SE2 = 1           /* R^2 x SO(2): dim must be 3, bounds = (x), (y), (theta: clamped to [-PI, PI]);
                                     planner must be OXHIP_PLANNER_RRT_CONNECT; validity = oxhip_rrt_batch_set_segments */
} oxhip_space_kind;

/* GoalSampleableRegion::sample_goal of the ball goal (goal.rs:35-41; the trait leaves the distribution to the implementor).
 * CENTRE: the ball's centre, no RNG word consumed -- README.md:160-162, oxmpl-js/examples/simple_2d_planning.js:30-33.
 * UNIFORM_DISC: the sampler of the reference's own test fixtures, oxmpl/tests/rrt_rvss_tests.rs:55-66 (dim must be 2):
 *   angle  = rng.random_range(0.0..2.0 * PI)      -- one u64 (52-bit transform)
 *   radius = self.radius * rng.random::<f64>().sqrt()   -- one u64 (53 bits x 2^-53)
 *   (x, y) = (cx + radius * angle.cos(), cy + radius * angle.sin())
 * cos / sin are evaluated by the portable routine ox_sincos (oxmpl_amd/csrc/ox_sincos.hpp: Cody-Waite reduction + fdlibm
 * kernels in unfused binary64; the test suite's CPU checker restates it operation for operation).  It differs from glibc's
 * sin / cos by at most one ulp on [0, 2 PI) (the CPU test suite measures how often), so against a rustc-built oxmpl
 * this mode is within north_star's 1e-6 relative bound rather than bit-exact. */
typedef enum oxhip_goal_sampler {
    OXHIP_GOAL_SAMPLE_CENTRE = 0,
    OXHIP_GOAL_SAMPLE_UNIFORM_DISC = 1
} oxhip_goal_sampler;

/* oxhip_rrt_config.debug_flags (tests only; every switch leaves every result bit-identical) */
typedef enum oxhip_debug_flag {
    OXHIP_DEBUG_PAIR_TO_WHOLE_TREE = 1,   /* rrt_lanes.hip: two-lane near-ties take the whole-tree path like three-way ones */
    OXHIP_DEBUG_AUDIT = 2,                /* rrt_lanes.hip, stamped build: count accepted motions whose end state is invalid (stamps[50]) */
    OXHIP_DEBUG_ALL_WHOLE_TREE = 4,       /* rrt_lanes.hip: no screen verdict is trusted -- every query that is not answered from the memoized
                                             whole-tree answer takes the whole-tree path (with goal_bias > 0 the memo is then hit constantly) */
    OXHIP_DEBUG_ONE_LANE_ROUNDS = 8,      /* rrt_lanes.hip: a round commits one lane; the others are re-resolved against the grown tree */
    OXHIP_DEBUG_SHORT_MEMO = 16,          /* rrt_lanes.hip: the memoized answer expires after 8 inserts instead of (ring - 64) */
    OXHIP_DEBUG_STAR_TWO_PASS = 32,       /* rrt_star_wire.hip: neighbour lists by a second search instead of the counting pass's chunks */
    OXHIP_DEBUG_STAR_ONE_SEGMENT = 64,    /* rrt_star_wire.hip: one edge-check segment, no overlap with the wiring stream */
    OXHIP_DEBUG_SE2_NO_SEGMENT_GRID = 128,/* rrt_connect_se2.hip / rrt_connect.hip: every interpolated state is tested against every segment / sphere (no grid lookup) */
    OXHIP_DEBUG_SE2_SMALL_LDS = 256       /* rrt_connect_se2.hip: the shape for batches larger than the chip (512-node shadows, segments from HBM / L2) whatever the batch size */
} oxhip_debug_flag;

typedef enum oxhip_kernel_kind {
    OXHIP_KERNEL_AUTO = 0,      /* RRT (and the geometry of the decoupled RRT*): OXHIP_KERNEL_CELLS in R^2 / R^3 (trees up to 64,512
                                   nodes; any batch size); OXHIP_KERNEL_LANES in R^4 .. R^6 when the tree fits its register rows;
                                   else streaming */
    OXHIP_KERNEL_STREAM = 1,    /* tree streamed from HBM/L2 SoA arrays every iteration: any dimension <= 8, any tree size */
    OXHIP_KERNEL_RESIDENT = 2,  /* tree held in the workgroup's vector registers as binary64, every node scanned in binary64
                                   (R^2 / R^3, <= 10,240 nodes): no binary32 anywhere -- the cross-check of the screened kernels */
    OXHIP_KERNEL_RETIRED_3 = 3, /* (ABI 1: box-pruned resident scan, an experiment) -- OXHIP_ERR_BAD_ARG since ABI 2 */
    OXHIP_KERNEL_RETIRED_4 = 4, /* (ABI 1: binary32 screen + lane-group resolver, round 1's default) -- OXHIP_ERR_BAD_ARG since ABI 2 */
    OXHIP_KERNEL_LANES = 5,     /* resident + binary32 screen with a lane-per-query resolver: up to 64 iterations are
                                   resolved side by side and committed as the longest prefix that keeps the reference's
                                   sequential semantics (rrt_lanes.hip); same results bit for bit */
    OXHIP_KERNEL_CELLS = 6      /* one WAVE per problem; nearest neighbour through an exact grid of cells (per-cell node lists in
                                   HBM / L2, the 3^D cells around a query first, further shells when they cannot rule out the rest),
                                   lane-per-query as above (rrt_cells.hip; R^2 / R^3, any tree size); same results bit for bit */
} oxhip_kernel_kind;

/* RRT::new(max_distance, goal_bias) (rrt.rs:75-83) + RealVectorStateSpace::new(dim, bounds)
 * (real_vector_state_space.rs:65-100) + the termination the reference lacks (rrt.rs:226). */
typedef struct oxhip_rrt_config {
    uint32_t struct_size;               /* = sizeof(oxhip_rrt_config) */
    uint32_t dim;                       /* RealVectorStateSpace::dimension, 1..OXHIP_MAX_DIM */
    double   bounds[2 * OXHIP_MAX_DIM]; /* (lo,hi) pairs, real_vector_state_space.rs:19 */
    double   max_distance;              /* RRT::max_distance  rrt.rs:55 */
    double   goal_bias;                 /* RRT::goal_bias     rrt.rs:57, must be in [0,1] */
    double   lvs_fraction;              /* longest_valid_segment_fraction, default 0.05 (rvss.rs:98);
                                           clamped like set_longest_valid_segment_fraction (rvss.rs:121-129) */
    uint32_t n_problems;                /* independent planner instances in this batch */
    uint32_t max_nodes;                 /* tree capacity per problem (build-defined node cap) */
    uint32_t stop_at_goal;              /* 1: stop a problem at its first goal node (reference behaviour) */
    uint32_t kernel;                    /* oxhip_kernel_kind */
    uint64_t seed;                      /* RNG key: ChaCha12 key = LE(seed)||0^24, stream id = first_problem_id + p */
    uint64_t first_problem_id;          /* global id of problem 0 of this batch (problem-parallel sharding) */
    int32_t  device;                    /* HIP device ordinal */
    uint32_t planner;                   /* oxhip_planner_kind: 0 = RRT (rrt.rs), 1 = RRTConnect (rrt_connect.rs), 2 = RRT* */
    double   search_radius;             /* RRTStar::search_radius (rrt_star.rs:45): neighbours are the nodes with
                                           distance < search_radius (strict); ignored by the other planners */
    uint32_t space;                     /* oxhip_space_kind */
    uint32_t goal_sampler;              /* oxhip_goal_sampler: what GoalSampleableRegion::sample_goal (goal.rs:35-41) draws */
    uint32_t debug_flags;               /* oxhip_debug_flag bits: test-only switches that force rarely taken code paths; results are
                                           identical by construction.  0 in production (the library reads no environment variable) */
    uint32_t star_pool_share;           /* RRT*, decoupled design: neighbour-list pool entries per problem (16 B each, + 4.5 B of chunk
                                           store); 0 = default: 64 x tree capacity, bounded so that the whole batch stays below 24 GB.
                                           The size bounds memory, never results: lists that do not fit are wired in further rounds */
    uint32_t frozen_split;              /* OXHIP_KERNEL_CELLS, solve(freeze = 1): waves a problem's frozen iterations are divided over
                                           (1 .. 64; 0 = automatic).  Results do not depend on it */
    uint32_t reserved;                  /* 0 */
} oxhip_rrt_config;

typedef struct oxhip_rrt_batch oxhip_rrt_batch;

/* ---- library ---- */
int32_t     oxhip_abi_version(void);
const char* oxhip_status_string(int32_t status);
const char* oxhip_last_error_string(void);              /* thread-local detail of the last OXHIP_ERR_* */
int32_t     oxhip_device_count(int32_t* count);         /* OXHIP_ERR_NO_DEVICE when none */

/* ---- batched planner: replaces RRT::new / Planner::setup / Planner::solve ---- */

/* RRT::new + RealVectorStateSpace::new for n_problems instances; allocates device trees. */
int32_t oxhip_rrt_batch_create(const oxhip_rrt_config* cfg, oxhip_rrt_batch** out);
int32_t oxhip_rrt_batch_destroy(oxhip_rrt_batch* b);

/* Device-describable StateValidityChecker (validity.rs:39-48), shared by all problems:
 * a state is valid iff for every sphere distance(centre, p) > radius  (strict, the shape of
 * README.md:147-150) and it is inside no box: NOT(lo_k <= p_k <= hi_k for all k)
 * (the wall of oxmpl/tests/rrt_rvss_tests.rs:24-36).  Replaces any earlier field. */
int32_t oxhip_rrt_batch_set_spheres(oxhip_rrt_batch* b, const double* centres /*[n][dim]*/,
                                    const double* radii /*[n]*/, uint32_t n);
int32_t oxhip_rrt_batch_set_boxes(oxhip_rrt_batch* b, const double* lo /*[n][dim]*/,
                                  const double* hi /*[n][dim]*/, uint32_t n);

/* SE(2) batches only.  The checker of BASELINE.json configs[3]: a disc robot of radius `clearance` among n line
 * segments (ax, ay, bx, by) -- a state is valid iff its (x, y) is farther than `clearance` from every segment
 * (strict; the heading does not enter).  Replaces any earlier segment set.  Also files the segments in a 256 x 256 lookup grid
 * over the (x, y) bounds on the device (1 MiB; rrt_connect_se2.hip, seg_grid_kernel): the planner's motion checks and
 * oxhip_rrt_batch_check_motion test a state against its cell's segments only -- the same verdicts, by construction. */
int32_t oxhip_rrt_batch_set_segments(oxhip_rrt_batch* b, const double* segments /*[n][4]*/, uint32_t n,
                                     double clearance);

/* Planner::setup (rrt.rs:140-156) for every problem: clears the tree, pushes start_states[0]
 * (validity of the start is NOT checked, as in the reference), resets counters and the RNG
 * stream.  Goal = ball: is_satisfied(s) = distance(s, centre) <= radius
 * (rrt_rvss_tests.rs:45-49); sample_goal() as oxhip_rrt_config.goal_sampler says. */
int32_t oxhip_rrt_batch_setup(oxhip_rrt_batch* b, const double* starts /*[P][dim]*/,
                              const double* goal_centres /*[P][dim]*/, const double* goal_radii /*[P]*/);

/* Warm start: replace problem `problem`'s tree (RRT::tree, rrt.rs:61) by n >= 1 nodes given as AoS
 * states [n][dim] and parent indices (parents[0] = -1).  Allowed after setup(); counters and the RNG
 * stream are left as they are.  Lets a caller continue a tree grown elsewhere (and lets the parity
 * tests plant adversarial trees). */
int32_t oxhip_rrt_batch_set_tree(oxhip_rrt_batch* b, uint32_t problem, const double* states,
                                 const int32_t* parents, uint32_t n_nodes);

/* Planner::solve (rrt.rs:158-227).  Runs every unfinished problem for at most max_iterations
 * further iterations (one iteration = one pass of rrt.rs:170-225).  timeout_s bounds wall time
 * (checked between kernel launches of at most 2048 iterations; 0 or +inf = none; NaN or negative =
 * OXHIP_ERR_BAD_ARG).  The planner mirrors above this ABI map the reference's `solve(Duration::ZERO)`
 * to PlanningError::Timeout themselves (rrt.rs:172-174 fails its first clock check).  Without a timeout
 * the budget is still cut into launches of 65,536 iterations so the host can see every problem stop.
 * freeze != 0 suppresses inserts
 * ("steady" measurement mode: every nearest-neighbour scan sees the same tree).
 * status_out[P] (may be NULL): OXHIP_OK if the problem has a goal node, else
 * OXHIP_ERR_NO_SOLUTION_FOUND / OXHIP_ERR_TIMEOUT.  Calling solve again continues the same
 * trees and RNG streams (the reference's tree also persists across solve calls). */
int32_t oxhip_rrt_batch_solve(oxhip_rrt_batch* b, uint64_t max_iterations, double timeout_s,
                              uint32_t freeze, int32_t* status_out);

/* per-problem counters (any pointer may be NULL) */
int32_t oxhip_rrt_batch_get_counts(oxhip_rrt_batch* b, uint64_t* iterations /*[P]*/,
                                   uint32_t* nodes /*[P]*/, uint64_t* accepted /*[P]*/,
                                   uint64_t* checksum /*[P]*/, int32_t* goal_node /*[P]*/,
                                   int32_t* stop_reason /*[P]*/);

/* RRT::tree of problem p (rrt.rs:61): states AoS [n][dim] and parent indices (-1 = None). */
int32_t oxhip_rrt_batch_get_tree(oxhip_rrt_batch* b, uint32_t problem, double* states,
                                 int32_t* parents, uint32_t cap_nodes, uint32_t* n_nodes);

/* reconstruct_path (rrt.rs:118-128) from the first goal node; len = 0 when unsolved. */
int32_t oxhip_rrt_batch_get_path(oxhip_rrt_batch* b, uint32_t problem, double* states,
                                 uint32_t cap_states, uint32_t* len);

/* RRTConnect only (rrt_connect.rs:58-59: start_tree / goal_tree).  get_tree / get_counts above address the
 * start tree (nodes[], goal_node[] = last start-tree node of the solution); these address the goal tree:
 * its size, the last goal-tree node of the solution (-1 when the start tree reached the goal itself,
 * rrt_connect.rs:271-274) and its nodes.  get_path returns the merged path of rrt_connect.rs:288-304. */
int32_t oxhip_rrt_batch_get_goal_counts(oxhip_rrt_batch* b, uint32_t* nodes /*[P]*/, int32_t* end_node /*[P]*/);
int32_t oxhip_rrt_batch_get_goal_tree(oxhip_rrt_batch* b, uint32_t problem, double* states, int32_t* parents,
                                      uint32_t cap_nodes, uint32_t* n_nodes);

/* RRT* only: Node::cost (rrt_star.rs:26) of every node of problem `problem`, in node order */
int32_t oxhip_rrt_batch_get_costs(oxhip_rrt_batch* b, uint32_t problem, double* costs, uint32_t cap_nodes,
                                  uint32_t* n_nodes);

/* HIP-event time (ms) of the kernels of the last solve call, their launch count, and which
 * kernel ran (oxhip_kernel_kind). */
int32_t oxhip_rrt_batch_last_timing(oxhip_rrt_batch* b, double* kernel_ms, uint32_t* launches,
                                    uint32_t* kernel_kind);

/* Diagnostics: in-kernel counters and cycle stamps of the resident kernels (a separately instantiated
 * diagnostic build; never time that build).  Up to OXHIP_STAMP_WORDS words; get_stamps writes
 * min(cap_words, OXHIP_STAMP_WORDS) of them.  rrt_lanes.hip (workgroup 0 unless noted): [1] resolver wait,
 * [2] work, [3] whole-tree-path cycles; [4] whole-tree-path events, [5] rounds, [6] lanes offered,
 * [7] iterations, [11] memoized answers used, [12] conflict cuts, [15] literal-loop ties,
 * [16+w] / [24+w] scanner wave w wait / work cycles, [32..39] resolver phases, [40..43] fold / conflict
 * trips and exact evaluations, [44..48] batch-wide maxima and sums, [50..53] audit (OXHIP_DEBUG_AUDIT),
 * [54] batch-wide whole-tree events, [55] memo hits, [56] conflict cuts, [57] ring wraps (rounds in which the
 * committed-node ring passed a multiple of its size), [58] two-lane passes. */
#define OXHIP_STAMP_WORDS 64
int32_t oxhip_rrt_batch_enable_stamps(oxhip_rrt_batch* b, uint32_t enable);
int32_t oxhip_rrt_batch_get_stamps(oxhip_rrt_batch* b, uint64_t* out, uint32_t cap_words);

/* ---- stand-alone batched primitives (same device functions as the planner kernels) ---- */

/* nearest-neighbour argmin of rrt.rs:187-196 for Q independent (tree, query) pairs:
 * nodes AoS [sum n_nodes][dim] concatenated, offsets via n_nodes[Q]; queries [Q][dim];
 * out: index (lowest index among post-sqrt ties) and min_dist = distance(nodes[idx], q). */
int32_t oxhip_nn_argmin_batch(int32_t device, uint32_t dim, const double* nodes,
                              const uint32_t* n_nodes, uint32_t n_queries, const double* queries,
                              uint32_t* out_index, double* out_min_dist);

/* RealVectorStateSpace::distance (rvss.rs:137-155) and ::interpolate (rvss.rs:161-186), n pairs */
int32_t oxhip_distance_batch(int32_t device, uint32_t dim, const double* a, const double* b,
                             uint32_t n, double* out);
int32_t oxhip_interpolate_batch(int32_t device, uint32_t dim, const double* from, const double* to,
                                const double* t, uint32_t n, double* out);

/* StateValidityChecker::is_valid for n states and RRT::check_motion (rrt.rs:90-116) for n
 * (from,to) pairs against the batch's current sphere/box field and space resolution. */
int32_t oxhip_rrt_batch_is_valid(oxhip_rrt_batch* b, const double* states, uint32_t n, uint8_t* out);
int32_t oxhip_rrt_batch_check_motion(oxhip_rrt_batch* b, const double* from, const double* to,
                                     uint32_t n, uint8_t* out);

/* device arithmetic self-test hooks: out[i] = op(a[i], b[i]) computed on the GPU.
 * op: 0 sqrt(a), 1 a/b, 2 ceil(a), 3 a + (b - a) * t (t = c[i], unfused), 4 (a-b)*(a-b),
 *     5 / 6: sin(a) / cos(a) as the disc goal sampler evaluates them (ox_sincos; 0 <= a < 2^19 pi/2) */
int32_t oxhip_f64_op_batch(int32_t device, uint32_t op, const double* a, const double* b,
                           const double* c, uint32_t n, double* out);
/* SO(2) / SE(2) arithmetic self-test hooks, n rows of (x, y, theta):
 * op 0: out[i] = (se2_distance(a_i, b_i), so2_normalise(a_i.theta), so2_distance(a_i.theta, b_i.theta))
 * op 1: out[i] = se2_interpolate(a_i, b_i, t[i])        (so2_state_space.rs:97-122, so2_state.rs:33-37) */
int32_t oxhip_se2_op_batch(int32_t device, uint32_t op, const double* a, const double* b, const double* t, uint32_t n,
                           double* out);
/* device RNG self-test: the first n u64 words of the (seed, stream) ChaCha12 stream */
int32_t oxhip_rng_u64_batch(int32_t device, uint64_t seed, uint64_t stream, uint32_t n, uint64_t* out);

/* ---- PRM: replaces geometric::PRM (oxmpl/src/geometric/planners/prm.rs) ----
 *
 * PRM::new(timeout, connection_radius) (prm.rs:70-78) + RealVectorStateSpace::new.  The reference builds
 * its roadmap for `timeout` seconds of wall clock with an OS-seeded RNG; the device path adds the
 * deterministic caps max_milestones / max_samples (checked where the reference reads its clock,
 * prm.rs:118) and a ChaCha12 stream (seed, stream) restarted by setup().  For a given final milestone
 * count the roadmap does not depend on how construction was batched. */
typedef struct oxhip_prm_config {
    uint32_t struct_size;               /* = sizeof(oxhip_prm_config) */
    uint32_t dim;                       /* 1..OXHIP_MAX_DIM */
    double   bounds[2 * OXHIP_MAX_DIM]; /* (lo,hi) pairs */
    double   timeout;                   /* PRM::timeout (prm.rs:50), seconds of construction; read between
                                           device rounds; <= 0 or inf: no wall-clock bound */
    double   connection_radius;         /* PRM::connection_radius (prm.rs:52): edge iff distance < radius (strict) */
    double   lvs_fraction;              /* longest_valid_segment_fraction, default 0.05 */
    uint32_t max_milestones;            /* construct_roadmap stops at this many milestones (build-defined) */
    int32_t  device;                    /* HIP device ordinal */
    uint64_t max_samples;               /* ... or after this many sample_uniform calls; 0 = 4096 * max_milestones + 2^22
                                           (so that a space with no valid state still returns) */
    uint64_t seed;                      /* ChaCha12 key = LE(seed)||0^24 */
    uint64_t stream;                    /* ChaCha12 stream id */
    uint32_t knn_k;                     /* 0: the reference's rule -- a new milestone connects to every earlier one within connection_radius
                                           (prm.rs:131-138).  k > 0: the k-nearest variant (BASELINE.json configs[4]: "all-pairs k-NN"): it connects
                                           to its k nearest earlier milestones, ordered by (distance, index), visited in ascending index order;
                                           check_motion and everything else as prm.rs.  The query's start connections keep the radius rule */
    uint32_t reserved;                  /* 0 */
} oxhip_prm_config;

typedef struct oxhip_prm oxhip_prm;

int32_t oxhip_prm_create(const oxhip_prm_config* cfg, oxhip_prm** out);
int32_t oxhip_prm_destroy(oxhip_prm* prm);
/* the device-describable StateValidityChecker, as for the RRT batch */
int32_t oxhip_prm_set_spheres(oxhip_prm* prm, const double* centres /*[n][dim]*/, const double* radii, uint32_t n);
int32_t oxhip_prm_set_boxes(oxhip_prm* prm, const double* lo /*[n][dim]*/, const double* hi, uint32_t n);
/* Planner::setup (prm.rs:217-225): stores start / ball goal, clears the roadmap, restarts the RNG stream */
int32_t oxhip_prm_setup(oxhip_prm* prm, const double* start, const double* goal_centre, double goal_radius);
/* PRM::set_problem_definition (prm.rs:88-90): new start / goal, roadmap kept (multi-query use) */
int32_t oxhip_prm_set_problem(oxhip_prm* prm, const double* start, const double* goal_centre, double goal_radius);
/* PRM::construct_roadmap (prm.rs:96-154); a no-op when a roadmap exists (prm.rs:106-113) */
int32_t oxhip_prm_construct_roadmap(oxhip_prm* prm);
/* roadmap.len(), sum of edges.len() over all nodes (= 2 x undirected edges), sample_uniform calls made */
int32_t oxhip_prm_get_sizes(oxhip_prm* prm, uint32_t* n_milestones, uint64_t* n_edge_entries, uint64_t* n_samples);
/* PRM::get_roadmap (prm.rs:82-84): states AoS [n][dim]; node i's `edges` = neighbours[offsets[i] .. offsets[i+1]),
 * in the reference's order (ascending).  Any pointer may be NULL. */
int32_t oxhip_prm_get_roadmap(oxhip_prm* prm, double* states, uint32_t cap_nodes, uint64_t* offsets /*[n+1]*/,
                              uint32_t* neighbours, uint64_t cap_entries);
/* Planner::solve (prm.rs:227-307): OXHIP_OK and the path [start, milestones...] (prm.rs:189-208), or
 * OXHIP_ERR_PLANNER_UNINITIALISED / _UNSAMPLED_STATE_SPACE / _INVALID_START_STATE / _NO_SOLUTION_FOUND /
 * _TIMEOUT (graph search, prm.rs:285-287; timeout_s <= 0 or inf: none).  path may be NULL (length only). */
int32_t oxhip_prm_solve(oxhip_prm* prm, double timeout_s, double* path /*[cap_states][dim]*/, uint32_t cap_states,
                        uint32_t* len);
/* start_connections / goal_indices of the last solve (prm.rs:249-264) */
int32_t oxhip_prm_get_query_sets(oxhip_prm* prm, uint32_t* start_connections, uint32_t cap_start, uint32_t* n_start,
                                 uint32_t* goal_indices, uint32_t cap_goal, uint32_t* n_goal);
/* HIP-event times (ms) of the last construct_roadmap / solve: phase_ms[6] = {sampling, all-pairs radius
 * search, edge check_motion, key sort + CSR, query kernel, host BFS}; in-radius pairs examined; sample batches
 * replayed because rand's range sampler rejected a draw */
int32_t oxhip_prm_last_timing(oxhip_prm* prm, double* phase_ms /*[6]*/, uint64_t* n_candidates,
                              uint32_t* redraw_batches);
/* k-nearest variant, last construct_roadmap: rows whose candidate radius held fewer than k earlier milestones and that were
 * searched exactly instead (diagnostic: results do not depend on it) */
int32_t oxhip_prm_knn_exact_rows(oxhip_prm* prm, uint32_t* rows);

#ifdef __cplusplus
}
#endif
#endif /* OXMPL_HIP_H */
