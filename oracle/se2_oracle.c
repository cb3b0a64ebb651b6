/* oracle/se2_oracle.c -- TEST INFRASTRUCTURE ONLY.  See se2_oracle.h. */
#define _POSIX_C_SOURCE 200809L
#include "se2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rrt_oracle.h"

static const double PI = 3.14159265358979323846; /* std::f64::consts::PI */

/* f64::rem_euclid: r = self % rhs; if r < 0.0 { r + rhs.abs() } else { r } */
static double rem_euclid(double a, double b) {
    double r = fmod(a, b);
    return r < 0.0 ? r + fabs(b) : r;
}

double orc_so2_normalise(double v) { return rem_euclid(v + PI, 2.0 * PI) - PI; }

double orc_so2_distance(double a, double b) {
    double diff = a - b;
    diff = rem_euclid(diff + PI, 2.0 * PI) - PI;
    return fabs(diff);
}

double orc_so2_interpolate(double from, double to, double t) {
    double d = orc_so2_normalise(to) - orc_so2_normalise(from);
    if (d > PI) d -= 2.0 * PI;
    else if (d < -PI) d += 2.0 * PI;
    double out = from + d * t;
    return orc_so2_normalise(out);
}

double orc_se2_distance(const double* a, const double* b) {
    double dr = orc_distance(a, b, 2);
    double ds = orc_so2_distance(a[2], b[2]);
    double ws = 0.5 * ds;
    return dr + ws;
}

void orc_se2_interpolate(const double* from, const double* to, double t, double* out) {
    double xy[2];
    orc_interpolate(from, to, t, xy, 2);
    double th = orc_so2_interpolate(from[2], to[2], t);
    out[0] = xy[0];
    out[1] = xy[1];
    out[2] = th;
}

double orc_se2_extent(const double* bounds_xy) {
    double e = orc_maximum_extent(bounds_xy, 2);
    double s = 0.5 * PI;
    return e + s;
}

double orc_point_segment_distance(double px, double py, const double* seg) {
    double abx = seg[2] - seg[0], aby = seg[3] - seg[1];
    double apx = px - seg[0], apy = py - seg[1];
    double l1 = abx * abx, l2 = aby * aby;
    double len2 = l1 + l2;
    double t = 0.0;
    if (len2 > 0.0) {
        double n1 = apx * abx, n2 = apy * aby;
        double num = n1 + n2;
        t = num / len2;
    }
    if (!(t > 0.0)) t = 0.0; /* also NaN */
    if (t > 1.0) t = 1.0;
    double sx = abx * t, sy = aby * t;
    double cx = seg[0] + sx, cy = seg[1] + sy;
    double dx = px - cx, dy = py - cy;
    double q1 = dx * dx, q2 = dy * dy;
    return sqrt(q1 + q2);
}

typedef struct { double s[3]; int64_t parent; } se2_node;

struct orc_se2c {
    double bounds_xy[4], th_lo, th_hi;
    double max_distance, goal_bias, lvs_fraction;
    uint32_t max_nodes;
    orc_rng rng;
    double* segs; uint32_t n_segs; double clearance;
    int is_setup;
    double goal[3], goal_radius;
    se2_node* tree[2]; uint32_t n[2], cap[2];
    uint64_t iterations, checksum;
    int32_t end[2], stop_reason;
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

orc_se2c* orc_se2c_new(const double* bounds_xy, double theta_lo, double theta_hi, double max_distance, double goal_bias,
                       double lvs_fraction, uint32_t max_nodes, uint64_t seed, uint64_t problem_id, int* status) {
    int st = ORC_SOLVED;
    if (max_nodes == 0) st = ORC_BAD_ARG;
    if (!(goal_bias >= 0.0 && goal_bias <= 1.0)) st = ORC_BAD_ARG;
    if (!(max_distance > 0.0) || !isfinite(max_distance)) st = ORC_BAD_ARG;
    for (int k = 0; k < 2 && st == ORC_SOLVED; ++k) {
        double lo = bounds_xy[2 * k], hi = bounds_xy[2 * k + 1];
        if (!isfinite(lo) || !isfinite(hi)) st = ORC_UNBOUNDED;
        else if (lo >= hi) st = ORC_ZERO_VOLUME;
    }
    if (st == ORC_SOLVED && !(theta_lo < theta_hi)) st = ORC_ZERO_VOLUME; /* so2_state_space.rs:59-64 */
    if (st == ORC_SOLVED && !(orc_clamp_fraction(lvs_fraction) > 0.0)) st = ORC_BAD_ARG;
    if (status) *status = st;
    if (st != ORC_SOLVED) return NULL;
    orc_se2c* r = (orc_se2c*)calloc(1, sizeof *r);
    memcpy(r->bounds_xy, bounds_xy, sizeof r->bounds_xy);
    r->th_lo = theta_lo > -PI ? theta_lo : -PI; /* bounds.0.max(-PI), so2_state_space.rs:67 */
    r->th_hi = theta_hi < PI ? theta_hi : PI;   /* bounds.1.min(PI) */
    r->max_distance = max_distance;
    r->goal_bias = goal_bias;
    r->lvs_fraction = orc_clamp_fraction(lvs_fraction);
    r->max_nodes = max_nodes;
    orc_rng_seed(&r->rng, seed, problem_id);
    r->end[0] = r->end[1] = -1;
    r->stop_reason = -1;
    r->checksum = 0xCBF29CE484222325ull;
    return r;
}

void orc_se2c_free(orc_se2c* r) {
    if (!r) return;
    free(r->tree[0]); free(r->tree[1]); free(r->segs);
    free(r);
}

int orc_se2c_set_segments(orc_se2c* r, const double* segs, uint32_t n, double clearance) {
    free(r->segs);
    r->segs = (double*)malloc(sizeof(double) * 4 * (n ? n : 1));
    if (n) memcpy(r->segs, segs, sizeof(double) * 4 * n);
    r->n_segs = n;
    r->clearance = clearance;
    return ORC_SOLVED;
}

static void push(orc_se2c* r, int w, const double* s, int64_t parent) {
    if (r->n[w] == r->cap[w]) {
        r->cap[w] = r->cap[w] ? r->cap[w] * 2 : 4;
        r->tree[w] = (se2_node*)realloc(r->tree[w], sizeof(se2_node) * r->cap[w]);
    }
    memcpy(r->tree[w][r->n[w]].s, s, sizeof(double) * 3);
    r->tree[w][r->n[w]].parent = parent;
    r->n[w]++;
}

int orc_se2c_setup(orc_se2c* r, const double* start, const double* goal, double goal_radius) {
    r->n[0] = r->n[1] = 0;
    memcpy(r->goal, goal, sizeof r->goal);
    r->goal_radius = goal_radius;
    push(r, 0, start, -1);
    push(r, 1, goal, -1); /* goal_tree = [goal.sample_goal()], rrt_connect.rs:218-224 */
    r->is_setup = 1;
    r->iterations = 0;
    r->checksum = 0xCBF29CE484222325ull;
    r->end[0] = r->end[1] = -1;
    r->stop_reason = -1;
    return ORC_SOLVED;
}

int orc_se2c_is_valid(const orc_se2c* r, const double* s) {
    for (uint32_t j = 0; j < r->n_segs; ++j)
        if (!(orc_point_segment_distance(s[0], s[1], r->segs + 4 * (size_t)j) > r->clearance)) return 0;
    return 1;
}

/* rrt_connect.rs:166-189 */
int orc_se2c_check_motion(const orc_se2c* r, const double* from, const double* to) {
    if (!r->is_setup) return 0;
    double dist = orc_se2_distance(from, to);
    double lvsl = orc_se2_extent(r->bounds_xy) * r->lvs_fraction;
    uint64_t num_steps = orc_num_steps(dist, lvsl);
    if (num_steps <= 1) return orc_se2c_is_valid(r, to);
    double interp[3];
    for (uint64_t i = 1; i <= num_steps; ++i) {
        double t = (double)i / (double)num_steps;
        orc_se2_interpolate(from, to, t, interp);
        if (!orc_se2c_is_valid(r, interp)) return 0;
    }
    return 1;
}

/* rrt_connect.rs:121-159 */
static int extend(orc_se2c* r, int w, const double* q_target, uint32_t* nearest_out, double* q_new) {
    const se2_node* tree = r->tree[w];
    uint32_t nearest = 0;
    double min_dist = orc_se2_distance(tree[0].s, q_target);
    for (uint32_t i = 1; i < r->n[w]; ++i) {
        double d = orc_se2_distance(tree[i].s, q_target);
        if (d < min_dist) { min_dist = d; nearest = i; }
    }
    double q_near[3];
    memcpy(q_near, tree[nearest].s, sizeof q_near);
    int result;
    if (min_dist > r->max_distance) {
        double t = r->max_distance / min_dist;
        orc_se2_interpolate(q_near, q_target, t, q_new);
        result = 1;
    } else {
        memcpy(q_new, q_target, sizeof(double) * 3);
        result = 2;
    }
    *nearest_out = nearest;
    if (!orc_se2c_check_motion(r, q_near, q_new)) return 0;
    push(r, w, q_new, (int64_t)nearest);
    return result;
}

static inline uint64_t mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001B3ull; }

/* rrt_connect.rs:227-309 */
int orc_se2c_solve(orc_se2c* r, uint64_t max_iterations, double timeout_s) {
    if (!r->is_setup) return ORC_PLANNER_UNINITIALISED;
    if (r->end[0] >= 0) return ORC_SOLVED;
    double start_time = now_s();
    r->stop_reason = ORC_STOP_ITERATIONS;
    int status = ORC_NO_SOLUTION_FOUND;
    double q_rand[3], qa[3], qb[3];
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if (now_s() - start_time > timeout_s) { r->stop_reason = ORC_STOP_TIMEOUT; status = ORC_TIMEOUT; break; }
        if (r->n[0] >= r->max_nodes || r->n[1] >= r->max_nodes) { r->stop_reason = ORC_STOP_NODES; break; }
        const int grow_start = r->n[0] <= r->n[1];
        if (orc_random_bool(&r->rng, r->goal_bias)) memcpy(q_rand, r->goal, sizeof q_rand);
        else {
            q_rand[0] = orc_random_range(&r->rng, r->bounds_xy[0], r->bounds_xy[1]);
            q_rand[1] = orc_random_range(&r->rng, r->bounds_xy[2], r->bounds_xy[3]);
            q_rand[2] = orc_random_range(&r->rng, r->th_lo, r->th_hi);
        }
        uint32_t near_a = 0, near_b = 0;
        const int wa = grow_start ? 0 : 1, wb = 1 - wa;
        const int ra = extend(r, wa, q_rand, &near_a, qa);
        uint64_t h = mix(r->checksum, (uint64_t)grow_start);
        h = mix(h, (uint64_t)near_a);
        for (int k = 0; k < 3; ++k) { uint64_t b; memcpy(&b, &qa[k], 8); h = mix(h, b); }
        h = mix(h, (uint64_t)ra);
        r->iterations++;
        int done = 0;
        if (ra) {
            const uint32_t idx_a = r->n[wa] - 1;
            if (grow_start && orc_se2_distance(qa, r->goal) <= r->goal_radius) { /* rrt_connect.rs:271-274 */
                r->end[0] = (int32_t)idx_a;
                r->end[1] = -1;
                done = 1;
            } else {
                const int rb = extend(r, wb, qa, &near_b, qb);
                h = mix(h, (uint64_t)near_b);
                for (int k = 0; k < 3; ++k) { uint64_t b; memcpy(&b, &qb[k], 8); h = mix(h, b); }
                h = mix(h, (uint64_t)rb);
                if (rb == 2) {
                    const uint32_t idx_b = r->n[wb] - 1;
                    r->end[wa] = (int32_t)idx_a;
                    r->end[wb] = (int32_t)idx_b;
                    done = 1;
                }
            }
        }
        r->checksum = h;
        if (done) { r->stop_reason = ORC_STOP_GOAL; status = ORC_SOLVED; break; }
    }
    return status;
}

uint32_t orc_se2c_num_nodes(const orc_se2c* r, int which) { return r->n[which ? 1 : 0]; }
uint64_t orc_se2c_iterations(const orc_se2c* r) { return r->iterations; }
uint64_t orc_se2c_checksum(const orc_se2c* r) { return r->checksum; }
int32_t orc_se2c_end_node(const orc_se2c* r, int which) { return r->end[which ? 1 : 0]; }
int32_t orc_se2c_stop_reason(const orc_se2c* r) { return r->stop_reason; }
void orc_se2c_theta_bounds(const orc_se2c* r, double* lo, double* hi) { *lo = r->th_lo; *hi = r->th_hi; }

void orc_se2c_get_tree(const orc_se2c* r, int which, double* states, int32_t* parents) {
    const int w = which ? 1 : 0;
    for (uint32_t i = 0; i < r->n[w]; ++i) {
        memcpy(states + (size_t)i * 3, r->tree[w][i].s, sizeof(double) * 3);
        parents[i] = (int32_t)r->tree[w][i].parent;
    }
}

/* rrt_connect.rs:288-304 */
uint32_t orc_se2c_get_path(const orc_se2c* r, double* out, uint32_t cap) {
    if (r->end[0] < 0) return 0;
    uint32_t la = 0, lb = 0;
    for (int64_t i = r->end[0]; i >= 0; i = r->tree[0][i].parent) ++la;
    if (r->end[1] >= 0) for (int64_t i = r->end[1]; i >= 0; i = r->tree[1][i].parent) ++lb;
    const uint32_t len = la + (lb ? lb - 1 : 0);
    if (len > cap || !out) return len;
    uint32_t pos = la;
    for (int64_t i = r->end[0]; i >= 0; i = r->tree[0][i].parent) { --pos; memcpy(out + (size_t)pos * 3, r->tree[0][i].s, sizeof(double) * 3); }
    pos = la;
    if (r->end[1] >= 0)
        for (int64_t i = r->tree[1][r->end[1]].parent; i >= 0; i = r->tree[1][i].parent) { memcpy(out + (size_t)pos * 3, r->tree[1][i].s, sizeof(double) * 3); ++pos; }
    return len;
}
