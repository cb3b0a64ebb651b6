/*
 * oracle/se2_oracle.h -- TEST INFRASTRUCTURE ONLY (see oracle/rrt_oracle.h for the rules).
 *
 * CPU restatement of RRTConnect (oxmpl/src/geometric/planners/rrt_connect.rs:86-309) over SE(2) = R^2 x SO(2)
 * with a segment-soup validity checker: BASELINE.json configs[3].  The reference has NO SE(2) space
 * (docs/BACKLOG.md:12-14); this build defines it from the reference's two component spaces, the way
 * OMPL's SE2StateSpace does:
 *   state      (x, y, theta)
 *   distance   1.0 * RealVectorStateSpace::distance(xy)  (real_vector_state_space.rs:137-155)
 *            + 0.5 * SO2StateSpace::distance(theta)      (so2_state_space.rs:97-101)
 *   interpolate  xy: real_vector_state_space.rs:161-186;  theta: so2_state_space.rs:107-122 (shortest way round,
 *                result normalised to [-PI, PI), so2_state.rs:33-37)
 *   sample     x, y, theta in that order, each rng.random_range(lo..hi) (rvss.rs:245, so2_state_space.rs:164-169)
 *   extent     extent(xy) + 0.5 * PI (so2_state_space.rs:78-80); lvsl = extent * fraction; check_motion as
 *              rrt_connect.rs:166-189 with resolution lvsl * 0.1
 *   SO(2) bounds  default (-PI, PI); given bounds clamped to [-PI, PI]; lo >= hi is InvalidBound (so2_state_space.rs:57-72)
 * Validity (build-defined; the reference leaves the checker to the user): a disc robot of radius `clearance`
 * among n line segments -- valid iff point_segment_distance((x, y), segment) > clearance for every segment.
 * The heading does not enter the predicate (no transcendental function does: device and host libm would not
 * agree on sin/cos to the last bit); it enters distance, steering, interpolation and the goal.
 * Goal: distance(state, target) <= radius with the SE(2) distance; sample_goal() = target, drawing nothing.
 *
 * PARITY UNPINNED: pinned only by the independent numpy restatement tests/golden/make_golden_se2.py.
 */
#ifndef OXMPL_SE2_ORACLE_H
#define OXMPL_SE2_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- component arithmetic ---- */
double orc_so2_normalise(double v);                              /* so2_state.rs:33-37 */
double orc_so2_distance(double a, double b);                     /* so2_state_space.rs:97-101 */
double orc_so2_interpolate(double from, double to, double t);    /* so2_state_space.rs:107-122 */
double orc_se2_distance(const double* a, const double* b);
void orc_se2_interpolate(const double* from, const double* to, double t, double* out);
double orc_se2_extent(const double* bounds_xy /*lo0,hi0,lo1,hi1*/);
double orc_point_segment_distance(double px, double py, const double* seg /*ax,ay,bx,by*/);

/* ---- RRTConnect over SE(2) ---- */
typedef struct orc_se2c orc_se2c;
orc_se2c* orc_se2c_new(const double* bounds_xy, double theta_lo, double theta_hi, double max_distance, double goal_bias,
                       double lvs_fraction, uint32_t max_nodes, uint64_t seed, uint64_t problem_id, int* status);
void orc_se2c_free(orc_se2c* r);
int orc_se2c_set_segments(orc_se2c* r, const double* segs /*[n][4]*/, uint32_t n, double clearance);
int orc_se2c_setup(orc_se2c* r, const double* start /*[3]*/, const double* goal /*[3]*/, double goal_radius);
int orc_se2c_solve(orc_se2c* r, uint64_t max_iterations, double timeout_s);
uint32_t orc_se2c_num_nodes(const orc_se2c* r, int which);
uint64_t orc_se2c_iterations(const orc_se2c* r);
uint64_t orc_se2c_checksum(const orc_se2c* r);
int32_t orc_se2c_end_node(const orc_se2c* r, int which);
int32_t orc_se2c_stop_reason(const orc_se2c* r);
void orc_se2c_get_tree(const orc_se2c* r, int which, double* states /*[n][3]*/, int32_t* parents);
uint32_t orc_se2c_get_path(const orc_se2c* r, double* out /*[cap][3]*/, uint32_t cap);
int orc_se2c_is_valid(const orc_se2c* r, const double* state);
int orc_se2c_check_motion(const orc_se2c* r, const double* from, const double* to);
void orc_se2c_theta_bounds(const orc_se2c* r, double* lo, double* hi);

#ifdef __cplusplus
}
#endif
#endif
