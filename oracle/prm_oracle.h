/*
 * oracle/prm_oracle.h -- TEST INFRASTRUCTURE ONLY (see oracle/rrt_oracle.h for the rules).
 *
 * CPU restatement (plain C, IEEE binary64, no FMA contraction) of the reference's PRM:
 *   oxmpl/src/geometric/planners/prm.rs:22-29    Node { state, edges }
 *   oxmpl/src/geometric/planners/prm.rs:70-90    PRM::new, set_problem_definition
 *   oxmpl/src/geometric/planners/prm.rs:96-154   construct_roadmap
 *   oxmpl/src/geometric/planners/prm.rs:161-187  check_motion
 *   oxmpl/src/geometric/planners/prm.rs:189-208  reconstruct_path
 *   oxmpl/src/geometric/planners/prm.rs:217-307  Planner::setup / Planner::solve (BFS)
 *
 * PARITY UNPINNED, as for the RRT oracle: the reference's PRM tests
 * (oxmpl/tests/prm_rvss_tests.rs) are stochastic property checks and hold no vectors.
 * Pinned by the independent numpy restatement in tests/golden/make_golden.py.
 *
 * Build-defined where the reference has a wall clock or an OS-seeded RNG:
 *   - construct_roadmap's loop (prm.rs:117-120) also stops once `max_milestones` milestones
 *     exist or `max_samples` sample_uniform calls were made (checked where the clock is);
 *   - rand::rng() (prm.rs:115) is the ChaCha12 stream (seed, stream), restarted by setup().
 */
#ifndef OXMPL_PRM_ORACLE_H
#define OXMPL_PRM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_INVALID_START_STATE = 4, ORC_UNSAMPLED_STATE_SPACE = 5 };

typedef struct orc_prm orc_prm;

orc_prm* orc_prm_new(uint32_t dim, const double* bounds /*2*dim lo,hi pairs*/, double timeout_s,
                     double connection_radius, double lvs_fraction, uint64_t seed, uint64_t stream, int* status);
void orc_prm_free(orc_prm* p);
int orc_prm_set_spheres(orc_prm* p, const double* centres, const double* radii, uint32_t n);
int orc_prm_set_boxes(orc_prm* p, const double* lo, const double* hi, uint32_t n);
/* Planner::setup (prm.rs:217-225): stores problem + checker, clears the roadmap */
int orc_prm_setup(orc_prm* p, const double* start, const double* goal_centre, double goal_radius);
/* set_problem_definition (prm.rs:88-90): new start / goal, roadmap kept */
int orc_prm_set_problem(orc_prm* p, const double* start, const double* goal_centre, double goal_radius);
/* construct_roadmap (prm.rs:96-154) */
/* k-nearest variant (BASELINE.json configs[4] says "all-pairs k-NN"; the reference itself connects by radius, prm.rs:131-138):
 * k > 0 connects every new milestone to its k nearest earlier ones by (distance, index); 0 = the reference's rule */
int orc_prm_set_knn(orc_prm* p, uint32_t k);
int orc_prm_construct_roadmap(orc_prm* p, uint32_t max_milestones, uint64_t max_samples);
uint32_t orc_prm_num_milestones(const orc_prm* p);
uint64_t orc_prm_num_edge_entries(const orc_prm* p); /* sum over nodes of edges.len() = 2 x undirected edges */
uint64_t orc_prm_num_samples(const orc_prm* p);
/* get_roadmap (prm.rs:82-84): states AoS [n][dim]; edges as CSR: offsets[n+1], neighbours in each node's
 * `edges` order */
void orc_prm_get_roadmap(const orc_prm* p, double* states, uint64_t* offsets, uint32_t* neighbours);
/* Planner::solve (prm.rs:227-307); the path is kept for orc_prm_get_path */
int orc_prm_solve(orc_prm* p, double timeout_s);
uint32_t orc_prm_get_path(const orc_prm* p, double* out /*cap*dim*/, uint32_t cap);
/* start_connections / goal_indices of the last solve (prm.rs:249-264), for finer-grained parity checks */
uint32_t orc_prm_get_start_connections(const orc_prm* p, uint32_t* out, uint32_t cap);
uint32_t orc_prm_get_goal_indices(const orc_prm* p, uint32_t* out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
