/*
 * oracle/rrt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rrt_oracle.h).
 *
 * CPU restatement of oxmpl's RRT::solve hot path.  It deliberately keeps the
 * reference's cost structure (one heap vector per state, a virtual validity
 * call per interpolated state, a clock read per iteration, the extent
 * recomputed by every check_motion) so that it can also serve as the
 * "oxmpl CPU path (C restatement)" baseline in bench.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * PARITY UNPINNED (rrt_oracle.h): no reference-held vector pins this path.
 */
#define _POSIX_C_SOURCE 200809L
#include "rrt_oracle.h"
#include "ox_sincos.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------- */
/* RNG: rand_chacha 0.9.0 ChaCha12Rng (rand's StdRng/ThreadRng core,         */
/* reference call site rrt.rs:167) restated from the published ChaCha        */
/* algorithm: 16-word state = 4 constants, 8 key words, 64-bit block         */
/* counter (words 12,13), 64-bit stream id (words 14,15); the block RNG      */
/* buffers 4 consecutive blocks and next_u64() = two consecutive LE words.   */
/* ------------------------------------------------------------------------- */

static inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define QR(a, b, c, d)            \
    do {                          \
        a += b; d ^= a; d = rotl32(d, 16); \
        c += d; b ^= c; b = rotl32(b, 12); \
        a += b; d ^= a; d = rotl32(d, 8);  \
        c += d; b ^= c; b = rotl32(b, 7);  \
    } while (0)

void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                      uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
    s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
    memcpy(x, s, sizeof x);
    for (int r = 0; r < rounds; r += 2) {
        QR(x[0], x[4], x[8], x[12]);
        QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]);
        QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]);
        QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);
        QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* Build-defined seeding: key = LE bytes of `seed` followed by 24 zero bytes
 * (ChaCha12Rng::from_seed), stream id = problem id (ChaCha12Rng::set_stream). */
void orc_rng_seed(orc_rng* r, uint64_t seed, uint64_t stream) {
    memset(r, 0, sizeof *r);
    r->key[0] = (uint32_t)seed;
    r->key[1] = (uint32_t)(seed >> 32);
    r->counter = 0;
    r->stream = stream;
    r->index = 64; /* empty buffer */
}

uint64_t orc_rng_next_u64(orc_rng* r) {
    if (r->index >= 64) {
        for (int b = 0; b < 4; ++b)
            orc_chacha_block(r->key, r->counter + (uint64_t)b, r->stream, 12, r->buf + 16 * b);
        r->counter += 4;
        r->index = 0;
    }
    uint64_t lo = r->buf[r->index], hi = r->buf[r->index + 1];
    r->index += 2;
    return (hi << 32) | lo;
}

/* rand 0.9 Bernoulli::new: p_int = (p * 2^64) as u64; p == 1.0 -> ALWAYS_TRUE */
uint64_t orc_bernoulli_p_int(double p) {
    if (p == 1.0) return UINT64_MAX;
    double v = p * 18446744073709551616.0; /* 2^64 */
    if (!(v > 0.0)) return 0;              /* Rust `as u64` saturates; NaN -> 0 */
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}

/* Rng::random_bool (rrt.rs:177): ALWAYS_TRUE draws nothing, otherwise one u64 */
int orc_random_bool(orc_rng* r, double p) {
    uint64_t p_int = orc_bernoulli_p_int(p);
    if (p_int == UINT64_MAX) return 1;
    return orc_rng_next_u64(r) < p_int;
}

/* Rng::random_range(lo..hi) for f64 (real_vector_state_space.rs:245):
 * 52 random mantissa bits -> [1,2) -> [0,1); res = v01*scale + lo unfused;
 * accept iff res < hi, else draw again. */
double orc_random_range(orc_rng* r, double lo, double hi) {
    double scale = hi - lo;
    for (;;) {
        uint64_t bits = (orc_rng_next_u64(r) >> 12) | 0x3FF0000000000000ull;
        double v12;
        memcpy(&v12, &bits, sizeof v12);
        double v01 = v12 - 1.0;
        double res = v01 * scale;
        res = res + lo;
        if (res < hi) return res;
    }
}

/* ------------------------------------------------------------------------- */
/* RealVectorStateSpace                                                       */
/* ------------------------------------------------------------------------- */

/* real_vector_state_space.rs:137-155: sequential sum from 0.0 of (a-b)^2, sqrt */
double orc_distance(const double* a, const double* b, uint32_t dim) {
    double acc = 0.0;
    for (uint32_t k = 0; k < dim; ++k) {
        double d = a[k] - b[k];
        double sq = d * d;
        acc = acc + sq;
    }
    return sqrt(acc);
}

/* real_vector_state_space.rs:161-186: from + (to - from) * t, three roundings */
void orc_interpolate(const double* from, const double* to, double t, double* out, uint32_t dim) {
    for (uint32_t k = 0; k < dim; ++k) {
        double diff = to[k] - from[k];
        double scaled = diff * t;
        out[k] = from[k] + scaled;
    }
}

/* real_vector_state_space.rs:103-118 */
double orc_maximum_extent(const double* bounds, uint32_t dim) {
    for (uint32_t k = 0; k < dim; ++k)
        if (!isfinite(bounds[2 * k]) || !isfinite(bounds[2 * k + 1])) return 1.0;
    double acc = 0.0;
    for (uint32_t k = 0; k < dim; ++k) {
        double d = bounds[2 * k + 1] - bounds[2 * k];
        double sq = d * d;
        acc = acc + sq;
    }
    return sqrt(acc);
}

/* real_vector_state_space.rs:121-129 */
double orc_clamp_fraction(double fraction) {
    if (fraction > 0.0 && fraction <= 1.0) return fraction;
    if (fraction <= 0.0) return 0.0;
    return 1.0; /* > 1 and NaN both land in the reference's final else */
}

/* rrt.rs:96-97: (dist / (lvsl * 0.1)).ceil() as usize, Rust saturating cast */
uint64_t orc_num_steps(double dist, double lvsl) {
    double res = lvsl * 0.1;
    double q = dist / res;
    double c = ceil(q);
    if (!(c > 0.0)) return 0; /* NaN, negatives, zero */
    if (c >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)c;
}

/* ------------------------------------------------------------------------- */
/* RRT                                                                        */
/* ------------------------------------------------------------------------- */

typedef struct {
    double* values;  /* RealVectorState { values: Vec<f64> } real_vector_state.rs:5-8 */
    int64_t parent;  /* Option<usize>, -1 = None (rrt.rs:24-27) */
} orc_node;

typedef struct orc_checker {
    int (*is_valid)(const struct orc_checker*, const double*); /* dyn StateValidityChecker */
    uint32_t dim;
    uint32_t n_spheres;
    double* sphere_c;
    double* sphere_r;
    uint32_t n_boxes;
    double* box_lo;
    double* box_hi;
} orc_checker;

struct orc_rrt {
    uint32_t dim;
    double bounds[2 * ORC_MAX_DIM];
    double max_distance;
    double goal_bias;
    double lvs_fraction;
    uint32_t max_nodes;
    int stop_at_goal;
    orc_rng rng;
    orc_checker checker;
    int is_setup;
    double goal_centre[ORC_MAX_DIM];
    int goal_sampler;   /* 0: sample_goal() = centre, no draw; 1: uniform in the disc, rrt_rvss_tests.rs:55-66 */
    double goal_radius;
    orc_node* tree;
    uint32_t n, cap;
    uint64_t iterations;
    uint64_t accepted;
    uint64_t checksum;
    int32_t goal_node;
    int32_t stop_reason;
};

static int field_is_valid(const orc_checker* c, const double* p) {
    for (uint32_t s = 0; s < c->n_spheres; ++s)
        if (!(orc_distance(c->sphere_c + (size_t)s * c->dim, p, c->dim) > c->sphere_r[s])) return 0;
    for (uint32_t b = 0; b < c->n_boxes; ++b) {
        int inside = 1;
        for (uint32_t k = 0; k < c->dim; ++k) {
            double v = p[k];
            if (!(v >= c->box_lo[(size_t)b * c->dim + k] && v <= c->box_hi[(size_t)b * c->dim + k])) {
                inside = 0;
                break;
            }
        }
        if (inside) return 0;
    }
    return 1;
}

orc_rrt* orc_rrt_new(uint32_t dim, const double* bounds, double max_distance, double goal_bias,
                     double lvs_fraction, uint32_t max_nodes, int stop_at_goal, uint64_t seed,
                     uint64_t problem_id, int* status) {
    int st = ORC_SOLVED;
    if (dim == 0 || dim > ORC_MAX_DIM || max_nodes == 0) st = ORC_BAD_ARG;
    if (!(goal_bias >= 0.0 && goal_bias <= 1.0)) st = ORC_BAD_ARG; /* Bernoulli::new error */
    if (!(max_distance > 0.0) || !isfinite(max_distance)) st = ORC_BAD_ARG;
    if (st == ORC_SOLVED)
        for (uint32_t k = 0; k < dim; ++k) {
            double lo = bounds[2 * k], hi = bounds[2 * k + 1];
            if (!isfinite(lo) || !isfinite(hi)) { st = ORC_UNBOUNDED; break; } /* rvss.rs:239-241 */
            if (lo >= hi) { st = ORC_ZERO_VOLUME; break; }                     /* rvss.rs:242-244 */
            if (!isfinite(hi - lo)) { st = ORC_UNBOUNDED; break; }
        }
    if (st == ORC_SOLVED && !(orc_clamp_fraction(lvs_fraction) > 0.0)) st = ORC_BAD_ARG;
    if (status) *status = st;
    if (st != ORC_SOLVED) return NULL;
    orc_rrt* r = (orc_rrt*)calloc(1, sizeof *r);
    r->dim = dim;
    memcpy(r->bounds, bounds, sizeof(double) * 2 * dim);
    r->max_distance = max_distance;
    r->goal_bias = goal_bias;
    r->lvs_fraction = orc_clamp_fraction(lvs_fraction);
    r->max_nodes = max_nodes;
    r->stop_at_goal = stop_at_goal;
    orc_rng_seed(&r->rng, seed, problem_id);
    r->checker.is_valid = field_is_valid;
    r->checker.dim = dim;
    r->goal_node = -1;
    r->stop_reason = -1;
    r->checksum = 0xCBF29CE484222325ull;
    return r;
}

static void clear_tree(orc_rrt* r) {
    for (uint32_t i = 0; i < r->n; ++i) free(r->tree[i].values);
    r->n = 0;
}

void orc_rrt_free(orc_rrt* r) {
    if (!r) return;
    clear_tree(r);
    free(r->tree);
    free(r->checker.sphere_c);
    free(r->checker.sphere_r);
    free(r->checker.box_lo);
    free(r->checker.box_hi);
    free(r);
}

static double* dup_vec(const double* src, size_t n) {
    double* d = (double*)malloc(sizeof(double) * (n ? n : 1));
    if (n) memcpy(d, src, sizeof(double) * n);
    return d;
}

int orc_rrt_set_spheres(orc_rrt* r, const double* centres, const double* radii, uint32_t n) {
    free(r->checker.sphere_c);
    free(r->checker.sphere_r);
    r->checker.sphere_c = dup_vec(centres, (size_t)n * r->dim);
    r->checker.sphere_r = dup_vec(radii, n);
    r->checker.n_spheres = n;
    return ORC_SOLVED;
}

int orc_rrt_set_boxes(orc_rrt* r, const double* lo, const double* hi, uint32_t n) {
    free(r->checker.box_lo);
    free(r->checker.box_hi);
    r->checker.box_lo = dup_vec(lo, (size_t)n * r->dim);
    r->checker.box_hi = dup_vec(hi, (size_t)n * r->dim);
    r->checker.n_boxes = n;
    return ORC_SOLVED;
}

static void push_node(orc_rrt* r, const double* state, int64_t parent) {
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 4; /* Vec growth */
        r->tree = (orc_node*)realloc(r->tree, sizeof(orc_node) * r->cap);
    }
    r->tree[r->n].values = dup_vec(state, r->dim); /* state.clone() */
    r->tree[r->n].parent = parent;
    r->n++;
}

/* rrt.rs:140-156: clear the tree, push start_states[0] with parent None.
 * Start validity is NOT checked (reference behaviour). */
/* GoalSampleableRegion::sample_goal of the ball goal (goal.rs:35-41).  Mode 0: the centre, nothing drawn (README.md:160-162).
 * Mode 1, the reference's own test fixture (oxmpl/tests/rrt_rvss_tests.rs:55-66, R^2):
 *     let angle = rng.random_range(0.0..2.0 * PI);
 *     let radius = self.radius * rng.random::<f64>().sqrt();      -- rand 0.9 StandardUniform f64: (u64 >> 11) * 2^-53
 *     x = target[0] + radius * angle.cos();  y = target[1] + radius * angle.sin();
 * cos / sin through ox_sincos (the routine the device runs too) or, for comparison, this host's libm. */
static int g_sincos_libm = 0;
void orc_set_sincos_libm(int use_libm) { g_sincos_libm = use_libm; }
void orc_sincos(double x, double* s, double* c) {
    if (g_sincos_libm) { *s = sin(x); *c = cos(x); } else ox_sincos(x, s, c);
}
static void sample_goal(orc_rrt* r, double* q) {
    if (r->goal_sampler == 0) {
        memcpy(q, r->goal_centre, sizeof(double) * r->dim);
        return;
    }
    const double two_pi = 2.0 * 3.14159265358979323846;   /* 2.0 * std::f64::consts::PI */
    const double angle = orc_random_range(&r->rng, 0.0, two_pi);
    const double u01 = (double)(orc_rng_next_u64(&r->rng) >> 11) * 0x1p-53;
    const double radius = r->goal_radius * sqrt(u01);
    double sn, cs;
    orc_sincos(angle, &sn, &cs);
    const double rx = radius * cs, ry = radius * sn;
    q[0] = r->goal_centre[0] + rx;
    q[1] = r->goal_centre[1] + ry;
}
int orc_rrt_set_goal_sampler(orc_rrt* r, int mode) {
    if (!r || mode < 0 || mode > 1 || (mode == 1 && r->dim != 2)) return ORC_BAD_ARG;
    r->goal_sampler = mode;
    return 0;
}

int orc_rrt_setup(orc_rrt* r, const double* start, const double* goal_centre, double goal_radius) {
    clear_tree(r);
    memcpy(r->goal_centre, goal_centre, sizeof(double) * r->dim);
    r->goal_radius = goal_radius;
    push_node(r, start, -1);
    r->is_setup = 1;
    r->iterations = 0;
    r->accepted = 0;
    r->checksum = 0xCBF29CE484222325ull;
    r->goal_node = -1;
    r->stop_reason = -1;
    return ORC_SOLVED;
}

int orc_rrt_set_tree(orc_rrt* r, const double* states, const int32_t* parents, uint32_t n) {
    if (!r->is_setup) return ORC_PLANNER_UNINITIALISED;
    if (n == 0 || n > r->max_nodes || parents[0] != -1) return ORC_BAD_ARG;
    clear_tree(r);
    for (uint32_t i = 0; i < n; ++i) push_node(r, states + (size_t)i * r->dim, (int64_t)parents[i]);
    return ORC_SOLVED;
}

/* rrt.rs:90-116 */
static int check_motion(const orc_rrt* r, const double* from, const double* to) {
    if (!r->is_setup) return 0; /* rrt.rs:113-115 */
    const orc_checker* vc = &r->checker;
    double dist = orc_distance(from, to, r->dim);
    /* get_longest_valid_segment_length() recomputes the extent (rvss.rs:251-253) */
    double lvsl = orc_maximum_extent(r->bounds, r->dim) * r->lvs_fraction;
    uint64_t num_steps = orc_num_steps(dist, lvsl);
    if (num_steps <= 1) return vc->is_valid(vc, to);
    double* interp = dup_vec(from, r->dim); /* from.clone() */
    int ok = 1;
    for (uint64_t i = 1; i <= num_steps; ++i) {
        double t = (double)i / (double)num_steps;
        orc_interpolate(from, to, t, interp, r->dim);
        if (!vc->is_valid(vc, interp)) { ok = 0; break; }
    }
    free(interp);
    return ok;
}

int orc_rrt_check_motion(const orc_rrt* r, const double* from, const double* to) {
    return check_motion(r, from, to);
}

int orc_rrt_is_valid(const orc_rrt* r, const double* p) { return r->checker.is_valid(&r->checker, p); }

/* Goal::is_satisfied for the ball goal of the reference's tests
 * (oxmpl/tests/rrt_rvss_tests.rs:45-49): distance(state, target) <= radius */
static int goal_is_satisfied(const orc_rrt* r, const double* state) {
    return orc_distance(state, r->goal_centre, r->dim) <= r->goal_radius;
}

static inline uint64_t fnv_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001B3ull; }

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* rrt.rs:158-227.  Build-defined additions (the reference terminates on wall
 * clock only, rrt.rs:172-174,226): an iteration budget, a node cap checked
 * before any RNG draw of the iteration, `freeze` (inserts suppressed), a
 * running checksum of (nearest, q_new bits, accepted) per iteration. */
int orc_rrt_solve(orc_rrt* r, uint64_t max_iterations, int freeze, double timeout_s) {
    if (!r->is_setup) return ORC_PLANNER_UNINITIALISED; /* rrt.rs:160-163 */
    if (r->stop_at_goal && r->goal_node >= 0) return ORC_SOLVED;
    const uint32_t dim = r->dim;
    double start_time = now_s();
    r->stop_reason = ORC_STOP_ITERATIONS;
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if (now_s() - start_time > timeout_s) { /* rrt.rs:172-174 */
            r->stop_reason = ORC_STOP_TIMEOUT;
            return ORC_TIMEOUT;
        }
        if (!freeze && r->n >= r->max_nodes) {
            r->stop_reason = ORC_STOP_NODES;
            break;
        }
        /* 2. sample (rrt.rs:177-184) */
        double* q_rand = (double*)malloc(sizeof(double) * dim);
        if (orc_random_bool(&r->rng, r->goal_bias)) {
            sample_goal(r, q_rand);
        } else {
            for (uint32_t k = 0; k < dim; ++k) /* rvss.rs:236-246 */
                q_rand[k] = orc_random_range(&r->rng, r->bounds[2 * k], r->bounds[2 * k + 1]);
        }
        /* 3. nearest (rrt.rs:187-196) */
        uint32_t nearest = 0;
        double min_dist = orc_distance(r->tree[0].values, q_rand, dim);
        for (uint32_t i = 1; i < r->n; ++i) {
            double d = orc_distance(r->tree[i].values, q_rand, dim);
            if (d < min_dist) { min_dist = d; nearest = i; }
        }
        const double* q_near = r->tree[nearest].values;
        /* 4. steer (rrt.rs:199-208) */
        double* q_new = dup_vec(q_near, dim);
        if (min_dist > r->max_distance) {
            double t = r->max_distance / min_dist;
            orc_interpolate(q_near, q_rand, t, q_new, dim);
        } else {
            memcpy(q_new, q_rand, sizeof(double) * dim);
        }
        /* 5. motion check (rrt.rs:211) */
        int ok = check_motion(r, q_near, q_new);
        /* checksum (build-defined): digest g of this iteration = FNV-1a fold from the basis of (nearest, bits(q_new), ok);
         * running value H <- H * P + g (mod 2^64) */
        uint64_t g = fnv_mix(0xCBF29CE484222325ull, (uint64_t)nearest);
        for (uint32_t k = 0; k < dim; ++k) {
            uint64_t b;
            memcpy(&b, &q_new[k], sizeof b);
            g = fnv_mix(g, b);
        }
        g = fnv_mix(g, (uint64_t)ok);
        r->checksum = r->checksum * 0x100000001B3ull + g;
        r->iterations++;
        int hit = 0;
        if (ok) {
            r->accepted++;
            if (!freeze) {
                push_node(r, q_new, (int64_t)nearest); /* rrt.rs:213-217 */
                if (goal_is_satisfied(r, q_new)) {     /* rrt.rs:220-223 */
                    if (r->goal_node < 0) r->goal_node = (int32_t)(r->n - 1);
                    hit = 1;
                }
            }
        }
        free(q_new);
        free(q_rand);
        if (hit && r->stop_at_goal) {
            r->stop_reason = ORC_STOP_GOAL;
            return ORC_SOLVED;
        }
    }
    return (r->goal_node >= 0) ? ORC_SOLVED : ORC_NO_SOLUTION_FOUND;
}

uint32_t orc_rrt_num_nodes(const orc_rrt* r) { return r->n; }
uint64_t orc_rrt_iterations(const orc_rrt* r) { return r->iterations; }
uint64_t orc_rrt_checksum(const orc_rrt* r) { return r->checksum; }
uint64_t orc_rrt_accepted(const orc_rrt* r) { return r->accepted; }
int32_t orc_rrt_goal_node(const orc_rrt* r) { return r->goal_node; }
int32_t orc_rrt_stop_reason(const orc_rrt* r) { return r->stop_reason; }

void orc_rrt_get_tree(const orc_rrt* r, double* states, int32_t* parents) {
    for (uint32_t i = 0; i < r->n; ++i) {
        memcpy(states + (size_t)i * r->dim, r->tree[i].values, sizeof(double) * r->dim);
        parents[i] = (int32_t)r->tree[i].parent;
    }
}

/* rrt.rs:118-128: follow parent indices from the goal node, then reverse */
uint32_t orc_rrt_get_path(const orc_rrt* r, double* out, uint32_t cap) {
    if (r->goal_node < 0) return 0;
    uint32_t len = 0;
    for (int64_t i = r->goal_node; i >= 0; i = r->tree[i].parent) ++len;
    if (len > cap) return len;
    uint32_t pos = len;
    for (int64_t i = r->goal_node; i >= 0; i = r->tree[i].parent) {
        --pos;
        memcpy(out + (size_t)pos * r->dim, r->tree[i].values, sizeof(double) * r->dim);
    }
    return len;
}

uint32_t orc_nearest(const double* nodes, uint32_t n, uint32_t dim, const double* q, double* min_dist) {
    uint32_t nearest = 0;
    double best = orc_distance(nodes, q, dim);
    for (uint32_t i = 1; i < n; ++i) {
        double d = orc_distance(nodes + (size_t)i * dim, q, dim);
        if (d < best) { best = d; nearest = i; }
    }
    if (min_dist) *min_dist = best;
    return nearest;
}

/* ------------------------------------------------------------------------- */
/* RRT*: rrt_star.rs:39-289                                                   */
/* ------------------------------------------------------------------------- */

struct orc_rrts {
    orc_rrt* a;            /* tree (state + parent_index), space, checker, rng, goal, counters */
    double* cost;          /* Node::cost (rrt_star.rs:26), parallel to a->tree */
    uint32_t cap_cost;
    double search_radius;  /* rrt_star.rs:45 */
    uint64_t wire;         /* W: the wiring polynomial of the build-defined checksum (see orc_rrts_solve) */
};

orc_rrts* orc_rrts_new(uint32_t dim, const double* bounds, double max_distance, double goal_bias, double search_radius,
                       double lvs_fraction, uint32_t max_nodes, int stop_at_goal, uint64_t seed, uint64_t problem_id,
                       int* status) {
    orc_rrt* a = orc_rrt_new(dim, bounds, max_distance, goal_bias, lvs_fraction, max_nodes, stop_at_goal, seed, problem_id, status);
    if (!a) return NULL;
    orc_rrts* r = (orc_rrts*)calloc(1, sizeof *r);
    r->a = a;
    r->search_radius = search_radius;
    return r;
}

void orc_rrts_free(orc_rrts* r) {
    if (!r) return;
    free(r->cost);
    orc_rrt_free(r->a);
    free(r);
}

int orc_rrts_set_spheres(orc_rrts* r, const double* c, const double* rad, uint32_t n) { return orc_rrt_set_spheres(r->a, c, rad, n); }
int orc_rrts_set_boxes(orc_rrts* r, const double* lo, const double* hi, uint32_t n) { return orc_rrt_set_boxes(r->a, lo, hi, n); }

static void rrts_push_cost(orc_rrts* r, double c) {
    if (r->a->n > r->cap_cost) {
        r->cap_cost = r->cap_cost ? r->cap_cost * 2 : 4;
        while (r->cap_cost < r->a->n) r->cap_cost *= 2;
        r->cost = (double*)realloc(r->cost, sizeof(double) * r->cap_cost);
    }
    r->cost[r->a->n - 1] = c;
}

/* rrt_star.rs:148-168: tree = [Node{start, None, 0.0}] */
int orc_rrts_setup(orc_rrts* r, const double* start, const double* goal_centre, double goal_radius) {
    orc_rrt_setup(r->a, start, goal_centre, goal_radius);
    r->wire = 0;
    rrts_push_cost(r, 0.0);
    return ORC_SOLVED;
}

/* rrt_star.rs:170-289 with the build's iteration budget / node cap next to the wall clock */
int orc_rrts_solve(orc_rrts* r, uint64_t max_iterations, double timeout_s) {
    orc_rrt* a = r->a;
    if (!a->is_setup) return ORC_PLANNER_UNINITIALISED;
    if (a->stop_at_goal && a->goal_node >= 0) return ORC_SOLVED;
    const uint32_t dim = a->dim;
    double start_time = now_s();
    a->stop_reason = ORC_STOP_ITERATIONS;
    uint32_t* neighbours = NULL;
    uint32_t cap_nb = 0;
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if (now_s() - start_time > timeout_s) { a->stop_reason = ORC_STOP_TIMEOUT; free(neighbours); return ORC_TIMEOUT; } /* :175-177 */
        if (a->n >= a->max_nodes) { a->stop_reason = ORC_STOP_NODES; break; }
        /* 2. sample (:180-187) */
        double* q_rand = (double*)malloc(sizeof(double) * dim);
        if (orc_random_bool(&a->rng, a->goal_bias)) {
            sample_goal(a, q_rand);
        } else {
            for (uint32_t k = 0; k < dim; ++k) q_rand[k] = orc_random_range(&a->rng, a->bounds[2 * k], a->bounds[2 * k + 1]);
        }
        /* 3. nearest (:190-200) */
        uint32_t nearest = 0;
        double min_dist = orc_distance(a->tree[0].values, q_rand, dim);
        for (uint32_t i = 1; i < a->n; ++i) {
            double d = orc_distance(a->tree[i].values, q_rand, dim);
            if (d < min_dist) { min_dist = d; nearest = i; }
        }
        const double* q_near = a->tree[nearest].values;
        /* 4. steer (:203-209) */
        double* q_new = dup_vec(q_near, dim);
        if (min_dist > a->max_distance) {
            double t = a->max_distance / min_dist;
            orc_interpolate(q_near, q_rand, t, q_new, dim);
        } else {
            memcpy(q_new, q_rand, sizeof(double) * dim);
        }
        /* 5. check_motion(q_near, q_new) (:212-214) */
        int ok = check_motion(a, q_near, q_new);
        /* Build-defined checksum = H + W (mod 2^64).  H: RRT's iteration polynomial, H <- H P + g with g the FNV-1a fold of
           (nearest, bits of q_new, verdict) -- the geometry of the run, which does not depend on any cost.  W: the wiring
           polynomial, W <- W P + w per inserted node with w the fold of (chosen parent, bits of its cost, number and index
           sum of the rewired neighbours).  Two chains, so that the geometry and the wiring can be computed by different
           kernels; a single wrong parent, cost or rewire still changes the sum. */
        uint64_t H = a->checksum - r->wire;
        {
            uint64_t g = fnv_mix(0xCBF29CE484222325ull, (uint64_t)nearest);
            for (uint32_t k = 0; k < dim; ++k) {
                uint64_t b;
                memcpy(&b, &q_new[k], sizeof b);
                g = fnv_mix(g, b);
            }
            g = fnv_mix(g, (uint64_t)ok);
            H = H * 0x100000001B3ull + g;
        }
        a->checksum = H + r->wire;
        a->iterations++;
        if (!ok) { free(q_new); free(q_rand); continue; }
        a->accepted++;
        /* find_neighbours (:121-131): distance(node.state, tree[i].state) < search_radius, ascending i */
        uint32_t n_nb = 0;
        for (uint32_t i = 0; i < a->n; ++i)
            if (orc_distance(q_new, a->tree[i].values, dim) < r->search_radius) {
                if (n_nb == cap_nb) { cap_nb = cap_nb ? cap_nb * 2 : 16; neighbours = (uint32_t*)realloc(neighbours, sizeof(uint32_t) * cap_nb); }
                neighbours[n_nb++] = i;
            }
        /* 6. choose parent (:225-241); cost(current, neighbour) = neighbour.cost + distance(current, neighbour) (:104-113) */
        uint32_t best_parent = nearest;
        double min_cost = r->cost[nearest] + orc_distance(q_new, a->tree[nearest].values, dim);
        for (uint32_t e = 0; e < n_nb; ++e) {
            uint32_t nb = neighbours[e];
            double c = r->cost[nb] + orc_distance(q_new, a->tree[nb].values, dim);
            if (c < min_cost && check_motion(a, a->tree[nb].values, q_new)) { min_cost = c; best_parent = nb; }
        }
        /* 7. push (:244-250) */
        push_node(a, q_new, (int64_t)best_parent);
        rrts_push_cost(r, min_cost);
        const uint32_t new_idx = a->n - 1;
        /* 8. rewire (:253-282) */
        uint64_t rew_cnt = 0, rew_sum = 0;
        for (uint32_t e = 0; e < n_nb; ++e) {
            uint32_t nb = neighbours[e];
            if (a->tree[new_idx].parent == (int64_t)nb) continue;                                  /* :258-260 */
            double c2 = r->cost[new_idx] + orc_distance(a->tree[nb].values, a->tree[new_idx].values, dim);  /* cost(neighbour, new) */
            if (c2 < r->cost[nb] && check_motion(a, a->tree[new_idx].values, a->tree[nb].values)) {
                a->tree[nb].parent = (int64_t)new_idx;
                r->cost[nb] = c2;
                rew_cnt++;
                rew_sum += nb;
            }
        }
        {
            uint64_t w = fnv_mix(0xCBF29CE484222325ull, (uint64_t)best_parent), b;
            memcpy(&b, &min_cost, sizeof b);
            w = fnv_mix(w, b);
            w = fnv_mix(w, rew_cnt);
            w = fnv_mix(w, rew_sum);
            r->wire = r->wire * 0x100000001B3ull + w;
            a->checksum = H + r->wire;
        }
        /* 9. goal (:285-288) */
        int hit = 0;
        if (goal_is_satisfied(a, q_new)) {
            if (a->goal_node < 0) a->goal_node = (int32_t)new_idx;
            hit = 1;
        }
        free(q_new);
        free(q_rand);
        if (hit && a->stop_at_goal) { a->stop_reason = ORC_STOP_GOAL; free(neighbours); return ORC_SOLVED; }
    }
    free(neighbours);
    return (a->goal_node >= 0) ? ORC_SOLVED : ORC_NO_SOLUTION_FOUND;
}

orc_rrt* orc_rrts_base(orc_rrts* r) { return r->a; }
void orc_rrts_get_costs(const orc_rrts* r, double* out) { memcpy(out, r->cost, sizeof(double) * r->a->n); }

/* ------------------------------------------------------------------------- */
/* RRTConnect: rrt_connect.rs:86-309                                          */
/* ------------------------------------------------------------------------- */

struct orc_rrtc {
    orc_rrt* a;          /* start tree + everything shared (space, checker, rng, goal) */
    orc_node* tree_b;    /* goal tree */
    uint32_t nb, capb;
    int32_t end_a, end_b;
};

orc_rrtc* orc_rrtc_new(uint32_t dim, const double* bounds, double max_distance, double goal_bias, double lvs_fraction,
                       uint32_t max_nodes, uint64_t seed, uint64_t problem_id, int* status) {
    orc_rrt* a = orc_rrt_new(dim, bounds, max_distance, goal_bias, lvs_fraction, max_nodes, 1, seed, problem_id, status);
    if (!a) return NULL;
    orc_rrtc* r = (orc_rrtc*)calloc(1, sizeof *r);
    r->a = a;
    r->end_a = r->end_b = -1;
    return r;
}

static void rrtc_clear_b(orc_rrtc* r) {
    for (uint32_t i = 0; i < r->nb; ++i) free(r->tree_b[i].values);
    r->nb = 0;
}

void orc_rrtc_free(orc_rrtc* r) {
    if (!r) return;
    rrtc_clear_b(r);
    free(r->tree_b);
    orc_rrt_free(r->a);
    free(r);
}

int orc_rrtc_set_spheres(orc_rrtc* r, const double* c, const double* rad, uint32_t n) { return orc_rrt_set_spheres(r->a, c, rad, n); }
int orc_rrtc_set_boxes(orc_rrtc* r, const double* lo, const double* hi, uint32_t n) { return orc_rrt_set_boxes(r->a, lo, hi, n); }

static void push_b(orc_rrtc* r, const double* state, int64_t parent) {
    if (r->nb == r->capb) {
        r->capb = r->capb ? r->capb * 2 : 4;
        r->tree_b = (orc_node*)realloc(r->tree_b, sizeof(orc_node) * r->capb);
    }
    r->tree_b[r->nb].values = dup_vec(state, r->a->dim);
    r->tree_b[r->nb].parent = parent;
    r->nb++;
}

/* rrt_connect.rs:199-225: start tree = [start]; goal tree = [goal.sample_goal()] (the ball goal's sampler
 * returns its centre and draws nothing) */
int orc_rrtc_setup(orc_rrtc* r, const double* start, const double* goal_centre, double goal_radius) {
    orc_rrt_setup(r->a, start, goal_centre, goal_radius);
    rrtc_clear_b(r);
    push_b(r, goal_centre, -1);
    r->end_a = r->end_b = -1;
    return ORC_SOLVED;
}

/* rrt_connect.rs:121-159.  `tree`/`n` select the tree; returns 0 = motion invalid (None),
 * 1 = Advanced, 2 = Reached; *nearest_out and q_new are always filled (for the checksum). */
static int rrtc_extend(orc_rrtc* r, int which, const double* q_target, uint32_t* nearest_out, double* q_new) {
    orc_rrt* a = r->a;
    const uint32_t dim = a->dim;
    orc_node* tree = which ? r->tree_b : a->tree;
    const uint32_t n = which ? r->nb : a->n;
    uint32_t nearest = 0;
    double min_dist = orc_distance(tree[0].values, q_target, dim);
    for (uint32_t i = 1; i < n; ++i) {
        double d = orc_distance(tree[i].values, q_target, dim);
        if (d < min_dist) { min_dist = d; nearest = i; }
    }
    double* q_near = dup_vec(tree[nearest].values, dim); /* q_near.clone() */
    int result;
    if (min_dist > a->max_distance) {
        double t = a->max_distance / min_dist;
        orc_interpolate(q_near, q_target, t, q_new, dim);
        result = 1;
    } else {
        memcpy(q_new, q_target, sizeof(double) * dim);
        result = 2;
    }
    *nearest_out = nearest;
    int ok = check_motion(a, q_near, q_new);
    free(q_near);
    if (!ok) return 0;
    if (which) push_b(r, q_new, (int64_t)nearest);
    else push_node(a, q_new, (int64_t)nearest);
    return result;
}

/* rrt_connect.rs:227-309 with the same build-defined termination as orc_rrt_solve: an iteration budget
 * and a node cap (either tree full, checked before any draw of the iteration). */
int orc_rrtc_solve(orc_rrtc* r, uint64_t max_iterations, double timeout_s) {
    orc_rrt* a = r->a;
    if (!a->is_setup) return ORC_PLANNER_UNINITIALISED;
    if (r->end_a >= 0) return ORC_SOLVED;
    const uint32_t dim = a->dim;
    double start_time = now_s();
    a->stop_reason = ORC_STOP_ITERATIONS;
    double* q_rand = (double*)malloc(sizeof(double) * dim);
    double* q_new_a = (double*)malloc(sizeof(double) * dim);
    double* q_new_b = (double*)malloc(sizeof(double) * dim);
    int status = ORC_NO_SOLUTION_FOUND;
    for (uint64_t it = 0; it < max_iterations; ++it) {
        if (now_s() - start_time > timeout_s) { a->stop_reason = ORC_STOP_TIMEOUT; status = ORC_TIMEOUT; break; }
        if (a->n >= a->max_nodes || r->nb >= a->max_nodes) { a->stop_reason = ORC_STOP_NODES; break; }
        const int grow_start = a->n <= r->nb; /* rrt_connect.rs:249-254 */
        if (orc_random_bool(&a->rng, a->goal_bias)) memcpy(q_rand, a->goal_centre, sizeof(double) * dim);
        else
            for (uint32_t k = 0; k < dim; ++k) q_rand[k] = orc_random_range(&a->rng, a->bounds[2 * k], a->bounds[2 * k + 1]);
        uint32_t near_a = 0, near_b = 0;
        const int ra = rrtc_extend(r, grow_start ? 0 : 1, q_rand, &near_a, q_new_a);
        uint64_t h = fnv_mix(a->checksum, (uint64_t)grow_start);
        h = fnv_mix(h, (uint64_t)near_a);
        for (uint32_t k = 0; k < dim; ++k) { uint64_t b; memcpy(&b, &q_new_a[k], 8); h = fnv_mix(h, b); }
        h = fnv_mix(h, (uint64_t)ra);
        a->iterations++;
        int done = 0;
        if (ra) {
            const uint32_t idx_a = (grow_start ? a->n : r->nb) - 1;
            if (grow_start && goal_is_satisfied(a, q_new_a)) { /* rrt_connect.rs:271-274 */
                r->end_a = (int32_t)idx_a;
                r->end_b = -1;
                done = 1;
            } else {
                const int rb = rrtc_extend(r, grow_start ? 1 : 0, q_new_a, &near_b, q_new_b);
                h = fnv_mix(h, (uint64_t)near_b);
                for (uint32_t k = 0; k < dim; ++k) { uint64_t b; memcpy(&b, &q_new_b[k], 8); h = fnv_mix(h, b); }
                h = fnv_mix(h, (uint64_t)rb);
                if (rb == 2) { /* Reached: rrt_connect.rs:281-305 */
                    const uint32_t idx_b = (grow_start ? r->nb : a->n) - 1;
                    r->end_a = (int32_t)(grow_start ? idx_a : idx_b);
                    r->end_b = (int32_t)(grow_start ? idx_b : idx_a);
                    done = 1;
                }
            }
        }
        a->checksum = h;
        if (done) { a->stop_reason = ORC_STOP_GOAL; status = ORC_SOLVED; break; }
    }
    free(q_rand); free(q_new_a); free(q_new_b);
    return status;
}

uint32_t orc_rrtc_num_nodes(const orc_rrtc* r, int which) { return which ? r->nb : r->a->n; }
uint64_t orc_rrtc_iterations(const orc_rrtc* r) { return r->a->iterations; }
uint64_t orc_rrtc_checksum(const orc_rrtc* r) { return r->a->checksum; }
int32_t orc_rrtc_end_node(const orc_rrtc* r, int which) { return which ? r->end_b : r->end_a; }
int32_t orc_rrtc_stop_reason(const orc_rrtc* r) { return r->a->stop_reason; }

void orc_rrtc_get_tree(const orc_rrtc* r, int which, double* states, int32_t* parents) {
    const orc_node* tree = which ? r->tree_b : r->a->tree;
    const uint32_t n = which ? r->nb : r->a->n, dim = r->a->dim;
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(states + (size_t)i * dim, tree[i].values, sizeof(double) * dim);
        parents[i] = (int32_t)tree[i].parent;
    }
}

/* rrt_connect.rs:288-304: start-tree chain root..end_a, then the goal-tree chain from end_b's parent
 * side reversed (goal path reversed, first element -- the duplicate connection point -- skipped) */
uint32_t orc_rrtc_get_path(const orc_rrtc* r, double* out, uint32_t cap) {
    if (r->end_a < 0) return 0;
    const uint32_t dim = r->a->dim;
    uint32_t la = 0, lb = 0;
    for (int64_t i = r->end_a; i >= 0; i = r->a->tree[i].parent) ++la;
    if (r->end_b >= 0) for (int64_t i = r->end_b; i >= 0; i = r->tree_b[i].parent) ++lb;
    const uint32_t len = la + (lb ? lb - 1 : 0);
    if (len > cap) return len;
    uint32_t pos = la;
    for (int64_t i = r->end_a; i >= 0; i = r->a->tree[i].parent) { --pos; memcpy(out + (size_t)pos * dim, r->a->tree[i].values, sizeof(double) * dim); }
    /* reconstruct_path(goal_tree, end_b) = [goal root .. end_b]; reversed = [end_b .. root]; skip(1) drops end_b */
    pos = la;
    if (r->end_b >= 0)
        for (int64_t i = r->tree_b[r->end_b].parent; i >= 0; i = r->tree_b[i].parent) { memcpy(out + (size_t)pos * dim, r->tree_b[i].values, sizeof(double) * dim); ++pos; }
    return len;
}

/* ------------------------------------------------------------------------- */
/* problem-parallel driver for the CPU baseline (one planner per problem,    */
/* the reference itself is single-threaded per planner)                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    orc_rrt** planners;
    uint32_t n;
    uint64_t max_iterations;
    int freeze;
    uint32_t next; /* shared work index, guarded by mu */
    pthread_mutex_t mu;
} many_job;

static void* many_worker(void* arg) {
    many_job* job = (many_job*)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        uint32_t i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n) break;
        orc_rrt_solve(job->planners[i], job->max_iterations, job->freeze, INFINITY);
    }
    return NULL;
}

int orc_rrt_solve_many(orc_rrt** planners, uint32_t n, uint64_t max_iterations, int freeze,
                       uint32_t threads) {
    many_job job;
    job.planners = planners;
    job.n = n;
    job.max_iterations = max_iterations;
    job.freeze = freeze;
    job.next = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (threads <= 1) {
        many_worker(&job);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
        for (uint32_t t = 0; t < threads; ++t) pthread_create(&th[t], NULL, many_worker, &job);
        for (uint32_t t = 0; t < threads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&job.mu);
    return ORC_SOLVED;
}
