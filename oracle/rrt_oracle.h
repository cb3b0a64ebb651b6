/*
 * oracle/rrt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, IEEE binary64, no FMA contraction) of the
 * reference's RRT hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (oxmpl_amd/, include/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (Rust) cannot be built or imported in this
 * environment and its own tests hold no golden vectors for this path
 * (SURVEY.md section 8c), so this oracle is pinned only by (i) the arithmetic
 * KATs and trees produced by an independent numpy restatement
 * (tests/golden/make_golden.py) and (ii) published ChaCha test vectors for the
 * RNG block function.  Equivalence with a rustc-built oxmpl is unverified.
 *
 * Reference lines followed (relative to /root/reference):
 *   oxmpl/src/geometric/planners/rrt.rs:24-27,75-83,90-128,140-227
 *   oxmpl/src/base/spaces/real_vector_state_space.rs:65-129,137-186,233-253
 *   oxmpl/src/base/states/real_vector_state.rs:5-13
 *   rand 0.9.1 Rng::random_bool / Rng::random_range(f64), rand_chacha 0.9.0
 *   ChaCha12Rng (third-party, absent from the image: restated from the
 *   published algorithm; see DESIGN.md "RNG").
 */
#ifndef OXMPL_RRT_ORACLE_H
#define OXMPL_RRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_DIM 8

/* status codes: mirror include/oxmpl_hip.h (PlanningError, error.rs:97-108) */
enum {
    ORC_SOLVED = 0,
    ORC_TIMEOUT = 1,
    ORC_NO_SOLUTION_FOUND = 2,   /* iteration budget or node cap reached */
    ORC_PLANNER_UNINITIALISED = 3,
    ORC_BAD_ARG = 16,
    ORC_UNBOUNDED = 17,
    ORC_ZERO_VOLUME = 18
};

/* stop reasons reported next to the status */
enum { ORC_STOP_GOAL = 0, ORC_STOP_ITERATIONS = 1, ORC_STOP_NODES = 2, ORC_STOP_TIMEOUT = 3 };

typedef struct orc_rng {
    uint32_t key[8];
    uint64_t counter;
    uint64_t stream;
    uint32_t buf[64];
    uint32_t index;
} orc_rng;

typedef struct orc_rrt orc_rrt;

/* ---- RNG (rand_chacha::ChaCha12Rng + rand 0.9 transforms) ---- */
void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                      uint32_t out[16]);
void orc_rng_seed(orc_rng* r, uint64_t seed, uint64_t stream);
uint64_t orc_rng_next_u64(orc_rng* r);
int orc_random_bool(orc_rng* r, double p);
double orc_random_range(orc_rng* r, double lo, double hi);
uint64_t orc_bernoulli_p_int(double p);

/* ---- RealVectorStateSpace (real_vector_state_space.rs) ---- */
double orc_distance(const double* a, const double* b, uint32_t dim);
void orc_interpolate(const double* from, const double* to, double t, double* out, uint32_t dim);
double orc_maximum_extent(const double* bounds, uint32_t dim);
double orc_clamp_fraction(double fraction);
uint64_t orc_num_steps(double dist, double lvsl);

/* ---- RRT (rrt.rs) ---- */
orc_rrt* orc_rrt_new(uint32_t dim, const double* bounds /*2*dim lo,hi pairs*/, double max_distance,
                     double goal_bias, double lvs_fraction, uint32_t max_nodes, int stop_at_goal,
                     uint64_t seed, uint64_t problem_id, int* status);
void orc_rrt_free(orc_rrt* r);
/* device-describable validity: spheres (valid iff distance(c,p) > r for all) and
 * axis-aligned boxes (invalid iff lo_k <= p_k <= hi_k for all k) */
int orc_rrt_set_spheres(orc_rrt* r, const double* centres, const double* radii, uint32_t n);
int orc_rrt_set_boxes(orc_rrt* r, const double* lo, const double* hi, uint32_t n);
/* Planner::setup (rrt.rs:140-156) with a ball goal whose sample_goal() is the centre */
int orc_rrt_setup(orc_rrt* r, const double* start, const double* goal_centre, double goal_radius);
/* GoalSampleableRegion::sample_goal (goal.rs:35-41): 0 = the centre, no draw (README.md:160-162); 1 = uniform in the disc as the
 * reference's test fixtures sample it (rrt_rvss_tests.rs:55-66; dim must be 2).  Applies to RRT and RRT* (orc_rrts_base). */
int orc_rrt_set_goal_sampler(orc_rrt* r, int mode);
/* sin / cos of the disc sampler: ox_sincos (default: the portable routine the device runs too) or this host's libm */
void orc_set_sincos_libm(int use_libm);
void orc_sincos(double x, double* s, double* c);
/* warm start: replace the tree by n nodes (AoS states, parents[0] = -1); counters / RNG untouched */
int orc_rrt_set_tree(orc_rrt* r, const double* states, const int32_t* parents, uint32_t n);
/* Planner::solve (rrt.rs:158-227) with a deterministic iteration budget.
 * freeze != 0: "steady" mode, inserts suppressed (tree size constant). */
int orc_rrt_solve(orc_rrt* r, uint64_t max_iterations, int freeze, double timeout_s);
uint32_t orc_rrt_num_nodes(const orc_rrt* r);
uint64_t orc_rrt_iterations(const orc_rrt* r);
uint64_t orc_rrt_checksum(const orc_rrt* r);
uint64_t orc_rrt_accepted(const orc_rrt* r);
int32_t orc_rrt_goal_node(const orc_rrt* r);
int32_t orc_rrt_stop_reason(const orc_rrt* r);
void orc_rrt_get_tree(const orc_rrt* r, double* states /*n*dim AoS*/, int32_t* parents);
uint32_t orc_rrt_get_path(const orc_rrt* r, double* out /*cap*dim*/, uint32_t cap);
int orc_rrt_check_motion(const orc_rrt* r, const double* from, const double* to);
int orc_rrt_is_valid(const orc_rrt* r, const double* p);
/* nearest neighbour exactly as rrt.rs:187-196 over an AoS array */
uint32_t orc_nearest(const double* nodes_aos, uint32_t n, uint32_t dim, const double* q, double* min_dist);

/* ---- RRTConnect (oxmpl/src/geometric/planners/rrt_connect.rs:86-309) ----
 * Same construction arguments as orc_rrt_new (max_nodes caps EACH tree).  Returns the status;
 * counters / trees through the getters; which = 0 start tree, 1 goal tree. */
typedef struct orc_rrtc orc_rrtc;
orc_rrtc* orc_rrtc_new(uint32_t dim, const double* bounds, double max_distance, double goal_bias, double lvs_fraction,
                       uint32_t max_nodes, uint64_t seed, uint64_t problem_id, int* status);
void orc_rrtc_free(orc_rrtc* r);
int orc_rrtc_set_spheres(orc_rrtc* r, const double* centres, const double* radii, uint32_t n);
int orc_rrtc_set_boxes(orc_rrtc* r, const double* lo, const double* hi, uint32_t n);
int orc_rrtc_setup(orc_rrtc* r, const double* start, const double* goal_centre, double goal_radius);
int orc_rrtc_solve(orc_rrtc* r, uint64_t max_iterations, double timeout_s);
uint32_t orc_rrtc_num_nodes(const orc_rrtc* r, int which);
uint64_t orc_rrtc_iterations(const orc_rrtc* r);
uint64_t orc_rrtc_checksum(const orc_rrtc* r);
int32_t orc_rrtc_end_node(const orc_rrtc* r, int which);   /* last node of the solution in each tree, -1 = none */
int32_t orc_rrtc_stop_reason(const orc_rrtc* r);
void orc_rrtc_get_tree(const orc_rrtc* r, int which, double* states, int32_t* parents);
uint32_t orc_rrtc_get_path(const orc_rrtc* r, double* out, uint32_t cap);

/* ---- RRT* (oxmpl/src/geometric/planners/rrt_star.rs:39-289) ----
 * Wraps an orc_rrt (tree, space, checker, RNG, goal, counters: use orc_rrts_base() with the orc_rrt_* getters)
 * and adds Node::cost.  The per-iteration checksum also folds in the chosen parent, the new node's cost and
 * the count / index sum of the rewired neighbours. */
typedef struct orc_rrts orc_rrts;
orc_rrts* orc_rrts_new(uint32_t dim, const double* bounds, double max_distance, double goal_bias, double search_radius,
                       double lvs_fraction, uint32_t max_nodes, int stop_at_goal, uint64_t seed, uint64_t problem_id,
                       int* status);
void orc_rrts_free(orc_rrts* r);
int orc_rrts_set_spheres(orc_rrts* r, const double* centres, const double* radii, uint32_t n);
int orc_rrts_set_boxes(orc_rrts* r, const double* lo, const double* hi, uint32_t n);
int orc_rrts_setup(orc_rrts* r, const double* start, const double* goal_centre, double goal_radius);
int orc_rrts_solve(orc_rrts* r, uint64_t max_iterations, double timeout_s);
orc_rrt* orc_rrts_base(orc_rrts* r);
void orc_rrts_get_costs(const orc_rrts* r, double* out /*[n]*/);

/* run many independent problems on `threads` host threads (cpu_baseline leg) */
int orc_rrt_solve_many(orc_rrt** planners, uint32_t n, uint64_t max_iterations, int freeze,
                       uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif
