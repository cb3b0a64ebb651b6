/*
 * oracle/prm_oracle.c -- TEST INFRASTRUCTURE ONLY.  See prm_oracle.h.
 *
 * Keeps the reference's data structure on purpose (per-node heap state, per-node growing `edges`
 * vector, queue + parent map for the breadth-first query) so that the CPU timing of this file is a
 * fair stand-in for oxmpl's PRM, and so that the order of every side effect can be read off next to
 * prm.rs.
 */
#define _POSIX_C_SOURCE 200809L
#include "prm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rrt_oracle.h"

typedef struct {
    double* values;   /* state: RealVectorState { values: Vec<f64> } */
    uint32_t* edges;  /* edges: Vec<usize>  prm.rs:28 */
    uint32_t n_edges, cap_edges;
} prm_node;

struct orc_prm {
    uint32_t dim;
    double bounds[2 * ORC_MAX_DIM];
    double timeout;            /* PRM::timeout (prm.rs:50), seconds of roadmap construction */
    double connection_radius;  /* prm.rs:52 */
    uint32_t knn_k;            /* 0: the reference's radius connection; k > 0: the k-nearest variant (BASELINE.json configs[4]) */
    double lvs_fraction;
    uint64_t seed, stream;
    orc_rng rng;
    /* validity field (same predicate as the RRT oracle) */
    uint32_t n_spheres, n_boxes;
    double *sphere_c, *sphere_r, *box_lo, *box_hi;
    int is_setup;
    double start[ORC_MAX_DIM], goal_centre[ORC_MAX_DIM], goal_radius;
    prm_node* roadmap;
    uint32_t n, cap;
    uint64_t n_samples;
    /* last solve */
    double* path;
    uint32_t path_len;
    uint32_t *start_conn, n_start_conn, *goal_idx, n_goal_idx;
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int is_valid(const orc_prm* p, const double* s) {
    for (uint32_t j = 0; j < p->n_spheres; ++j)
        if (!(orc_distance(p->sphere_c + (size_t)j * p->dim, s, p->dim) > p->sphere_r[j])) return 0;
    for (uint32_t b = 0; b < p->n_boxes; ++b) {
        int inside = 1;
        for (uint32_t k = 0; k < p->dim; ++k) {
            double v = s[k];
            if (!(v >= p->box_lo[(size_t)b * p->dim + k] && v <= p->box_hi[(size_t)b * p->dim + k])) { inside = 0; break; }
        }
        if (inside) return 0;
    }
    return 1;
}

orc_prm* orc_prm_new(uint32_t dim, const double* bounds, double timeout_s, double connection_radius,
                     double lvs_fraction, uint64_t seed, uint64_t stream, int* status) {
    int st = ORC_SOLVED;
    if (dim == 0 || dim > ORC_MAX_DIM) st = ORC_BAD_ARG;
    if (st == ORC_SOLVED)
        for (uint32_t k = 0; k < dim; ++k) {
            double lo = bounds[2 * k], hi = bounds[2 * k + 1];
            if (!isfinite(lo) || !isfinite(hi)) { st = ORC_UNBOUNDED; break; } /* rvss.rs:239-241 */
            if (lo >= hi) { st = ORC_ZERO_VOLUME; break; }                     /* rvss.rs:242-244 */
        }
    if (st == ORC_SOLVED && !(orc_clamp_fraction(lvs_fraction) > 0.0)) st = ORC_BAD_ARG;
    if (status) *status = st;
    if (st != ORC_SOLVED) return NULL;
    orc_prm* p = (orc_prm*)calloc(1, sizeof *p);
    p->dim = dim;
    memcpy(p->bounds, bounds, sizeof(double) * 2 * dim);
    p->timeout = timeout_s;
    p->connection_radius = connection_radius;
    p->lvs_fraction = orc_clamp_fraction(lvs_fraction);
    p->seed = seed;
    p->stream = stream;
    orc_rng_seed(&p->rng, seed, stream);
    return p;
}

static void clear_roadmap(orc_prm* p) {
    for (uint32_t i = 0; i < p->n; ++i) { free(p->roadmap[i].values); free(p->roadmap[i].edges); }
    p->n = 0;
}

void orc_prm_free(orc_prm* p) {
    if (!p) return;
    clear_roadmap(p);
    free(p->roadmap);
    free(p->sphere_c); free(p->sphere_r); free(p->box_lo); free(p->box_hi);
    free(p->path); free(p->start_conn); free(p->goal_idx);
    free(p);
}

static double* dup_vec(const double* src, size_t n) {
    double* d = (double*)malloc(sizeof(double) * (n ? n : 1));
    if (n) memcpy(d, src, sizeof(double) * n);
    return d;
}

int orc_prm_set_spheres(orc_prm* p, const double* centres, const double* radii, uint32_t n) {
    free(p->sphere_c); free(p->sphere_r);
    p->sphere_c = dup_vec(centres, (size_t)n * p->dim);
    p->sphere_r = dup_vec(radii, n);
    p->n_spheres = n;
    return ORC_SOLVED;
}

int orc_prm_set_boxes(orc_prm* p, const double* lo, const double* hi, uint32_t n) {
    free(p->box_lo); free(p->box_hi);
    p->box_lo = dup_vec(lo, (size_t)n * p->dim);
    p->box_hi = dup_vec(hi, (size_t)n * p->dim);
    p->n_boxes = n;
    return ORC_SOLVED;
}

int orc_prm_set_problem(orc_prm* p, const double* start, const double* goal_centre, double goal_radius) {
    memcpy(p->start, start, sizeof(double) * p->dim);
    memcpy(p->goal_centre, goal_centre, sizeof(double) * p->dim);
    p->goal_radius = goal_radius;
    return ORC_SOLVED;
}

/* prm.rs:217-225 */
int orc_prm_setup(orc_prm* p, const double* start, const double* goal_centre, double goal_radius) {
    orc_prm_set_problem(p, start, goal_centre, goal_radius);
    clear_roadmap(p);  /* self.roadmap.clear() */
    p->n_samples = 0;
    orc_rng_seed(&p->rng, p->seed, p->stream);
    p->is_setup = 1;
    return ORC_SOLVED;
}

/* prm.rs:161-187 */
static int check_motion(const orc_prm* p, const double* from, const double* to) {
    if (!p->is_setup) return 0;
    double dist = orc_distance(from, to, p->dim);
    double lvsl = orc_maximum_extent(p->bounds, p->dim) * p->lvs_fraction; /* rvss.rs:251-253 */
    uint64_t num_steps = orc_num_steps(dist, lvsl);                         /* prm.rs:167-168 */
    if (num_steps <= 1) return is_valid(p, to);
    double* interp = dup_vec(from, p->dim); /* from.clone() */
    int ok = 1;
    for (uint64_t i = 1; i <= num_steps; ++i) {
        double t = (double)i / (double)num_steps;
        orc_interpolate(from, to, t, interp, p->dim);
        if (!is_valid(p, interp)) { ok = 0; break; }
    }
    free(interp);
    return ok;
}

static void push_edge(prm_node* nd, uint32_t v) {
    if (nd->n_edges == nd->cap_edges) {
        nd->cap_edges = nd->cap_edges ? nd->cap_edges * 2 : 4;
        nd->edges = (uint32_t*)realloc(nd->edges, sizeof(uint32_t) * nd->cap_edges);
    }
    nd->edges[nd->n_edges++] = v;
}

/* prm.rs:96-154 */
int orc_prm_construct_roadmap(orc_prm* p, uint32_t max_milestones, uint64_t max_samples) {
    if (!p->is_setup) return ORC_PLANNER_UNINITIALISED; /* prm.rs:97-104 */
    if (p->n != 0) return ORC_SOLVED;                   /* prm.rs:106-113: already constructed */
    double start_time = now_s();
    double q_rand[ORC_MAX_DIM];
    uint32_t* to_update = NULL;
    uint32_t cap_update = 0;
    double* knn_d = NULL;
    uint32_t* knn_i = NULL;
    uint32_t knn_cap = 0;
    for (;;) {
        if (now_s() - start_time > p->timeout) break;                        /* prm.rs:118-120 */
        if (p->n >= max_milestones || p->n_samples >= max_samples) break;    /* build-defined caps */
        for (uint32_t k = 0; k < p->dim; ++k)                                /* sample_uniform rvss.rs:233-249 */
            q_rand[k] = orc_random_range(&p->rng, p->bounds[2 * k], p->bounds[2 * k + 1]);
        p->n_samples++;
        if (!is_valid(p, q_rand)) continue;                                  /* prm.rs:123 */
        prm_node new_node = {dup_vec(q_rand, p->dim), NULL, 0, 0};
        uint32_t n_update = 0;
        /* k-nearest variant (an extension: the reference connects by radius): the candidates are the k earlier milestones
         * nearest to the new one, by (distance, index) -- strict order, the lower index first among equal distances --, visited
         * in ascending index order like the reference's loop, so every `edges` vector still ends up ascending (prm.rs:143-145). */
        if (p->knn_k > 0 && p->n > 0) {
            const uint32_t k = p->knn_k < p->n ? p->knn_k : p->n;
            if (k > knn_cap) {
                knn_cap = k;
                knn_d = (double*)realloc(knn_d, sizeof(double) * knn_cap);
                knn_i = (uint32_t*)realloc(knn_i, sizeof(uint32_t) * knn_cap);
            }
            uint32_t m = 0;   /* the k best so far, sorted by (distance, index) ascending */
            for (uint32_t i = 0; i < p->n; ++i) {
                const double dist = orc_distance(q_rand, p->roadmap[i].values, p->dim);
                if (m == k && !(dist < knn_d[k - 1])) continue;   /* (equal distance, higher index: not better) */
                uint32_t pos = m < k ? m : k - 1;
                while (pos > 0 && dist < knn_d[pos - 1]) { knn_d[pos] = knn_d[pos - 1]; knn_i[pos] = knn_i[pos - 1]; --pos; }
                knn_d[pos] = dist;
                knn_i[pos] = i;
                if (m < k) ++m;
            }
            /* ascending index order */
            for (uint32_t a = 1; a < m; ++a) {
                uint32_t vi = knn_i[a];
                uint32_t b = a;
                while (b > 0 && knn_i[b - 1] > vi) { knn_i[b] = knn_i[b - 1]; --b; }
                knn_i[b] = vi;
            }
            for (uint32_t a = 0; a < m; ++a) {
                const uint32_t i = knn_i[a];
                if (check_motion(p, q_rand, p->roadmap[i].values)) {
                    push_edge(&new_node, i);
                    if (n_update == cap_update) {
                        cap_update = cap_update ? cap_update * 2 : 16;
                        to_update = (uint32_t*)realloc(to_update, sizeof(uint32_t) * cap_update);
                    }
                    to_update[n_update++] = i;
                }
            }
        }
        for (uint32_t i = 0; p->knn_k == 0 && i < p->n; ++i) {              /* prm.rs:131-138 */
            const double* other = p->roadmap[i].values;
            double dist = orc_distance(q_rand, other, p->dim);
            if (dist < p->connection_radius && check_motion(p, q_rand, other)) {
                push_edge(&new_node, i);
                if (n_update == cap_update) {
                    cap_update = cap_update ? cap_update * 2 : 16;
                    to_update = (uint32_t*)realloc(to_update, sizeof(uint32_t) * cap_update);
                }
                to_update[n_update++] = i;
            }
        }
        uint32_t new_idx = p->n;                                             /* prm.rs:140-141 */
        if (p->n == p->cap) {
            p->cap = p->cap ? p->cap * 2 : 4;
            p->roadmap = (prm_node*)realloc(p->roadmap, sizeof(prm_node) * p->cap);
        }
        p->roadmap[p->n++] = new_node;
        for (uint32_t u = 0; u < n_update; ++u) push_edge(&p->roadmap[to_update[u]], new_idx); /* prm.rs:143-145 */
    }
    free(to_update);
    free(knn_d);
    free(knn_i);
    return ORC_SOLVED;
}

/* k > 0: connect every new milestone to its k nearest earlier ones (by (distance, index)) instead of to every earlier one within
 * the connection radius; 0: the reference's rule.  The query (start connections) keeps the radius rule of prm.rs:249-256. */
int orc_prm_set_knn(orc_prm* p, uint32_t k) {
    if (!p) return ORC_BAD_ARG;
    p->knn_k = k;
    return 0;
}

uint32_t orc_prm_num_milestones(const orc_prm* p) { return p->n; }
uint64_t orc_prm_num_samples(const orc_prm* p) { return p->n_samples; }
uint64_t orc_prm_num_edge_entries(const orc_prm* p) {
    uint64_t e = 0;
    for (uint32_t i = 0; i < p->n; ++i) e += p->roadmap[i].n_edges;
    return e;
}

void orc_prm_get_roadmap(const orc_prm* p, double* states, uint64_t* offsets, uint32_t* neighbours) {
    uint64_t e = 0;
    for (uint32_t i = 0; i < p->n; ++i) {
        if (states) memcpy(states + (size_t)i * p->dim, p->roadmap[i].values, sizeof(double) * p->dim);
        if (offsets) offsets[i] = e;
        if (neighbours) memcpy(neighbours + e, p->roadmap[i].edges, sizeof(uint32_t) * p->roadmap[i].n_edges);
        e += p->roadmap[i].n_edges;
    }
    if (offsets) offsets[p->n] = e;
}

/* prm.rs:227-307 */
int orc_prm_solve(orc_prm* p, double timeout_s) {
    p->path_len = 0;
    p->n_start_conn = p->n_goal_idx = 0;
    if (!p->is_setup) return ORC_PLANNER_UNINITIALISED;        /* prm.rs:229-236 */
    if (p->n == 0) return ORC_UNSAMPLED_STATE_SPACE;           /* prm.rs:239-241 */
    if (!is_valid(p, p->start)) return ORC_INVALID_START_STATE; /* prm.rs:243-246 */
    const uint32_t n = p->n;
    free(p->start_conn); free(p->goal_idx);
    p->start_conn = (uint32_t*)malloc(sizeof(uint32_t) * n);
    p->goal_idx = (uint32_t*)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; ++i)                            /* prm.rs:249-256 */
        if (orc_distance(p->start, p->roadmap[i].values, p->dim) < p->connection_radius &&
            check_motion(p, p->start, p->roadmap[i].values))
            p->start_conn[p->n_start_conn++] = i;
    for (uint32_t i = 0; i < n; ++i)                            /* prm.rs:259-264; goal: distance(state, target) <= radius */
        if (orc_distance(p->roadmap[i].values, p->goal_centre, p->dim) <= p->goal_radius)
            p->goal_idx[p->n_goal_idx++] = i;
    if (p->n_start_conn == 0 || p->n_goal_idx == 0) return ORC_NO_SOLUTION_FOUND; /* prm.rs:266-268 */

    /* prm.rs:271-279: the queue starts as a copy of start_connections and then receives every start
     * connection a second time */
    size_t qcap = (size_t)n + 2 * (size_t)p->n_start_conn + 1, qhead = 0, qtail = 0;
    uint32_t* queue = (uint32_t*)malloc(sizeof(uint32_t) * qcap);
    int64_t* parent = (int64_t*)malloc(sizeof(int64_t) * n);  /* parent_map: -2 absent, -1 Some->None root */
    uint8_t* visited = (uint8_t*)calloc(n, 1);
    uint8_t* is_goal = (uint8_t*)calloc(n, 1);
    for (uint32_t g = 0; g < p->n_goal_idx; ++g) is_goal[p->goal_idx[g]] = 1;
    for (uint32_t i = 0; i < n; ++i) parent[i] = -2;
    for (uint32_t s = 0; s < p->n_start_conn; ++s) queue[qtail++] = p->start_conn[s];
    for (uint32_t s = 0; s < p->n_start_conn; ++s) {
        queue[qtail++] = p->start_conn[s];
        parent[p->start_conn[s]] = -1;
        visited[p->start_conn[s]] = 1;
    }
    int64_t goal_reached = -1;
    int status = ORC_SOLVED;
    double t0 = now_s();
    while (qhead < qtail) {                                     /* prm.rs:284-301 */
        uint32_t cur = queue[qhead++];
        if (now_s() - t0 > timeout_s) { status = ORC_TIMEOUT; break; }
        if (is_goal[cur]) { goal_reached = cur; break; }       /* goal_indices.contains(&current_idx) */
        const prm_node* nd = &p->roadmap[cur];
        for (uint32_t e = 0; e < nd->n_edges; ++e) {
            uint32_t nb = nd->edges[e];
            if (!visited[nb]) {
                visited[nb] = 1;
                parent[nb] = (int64_t)cur;
                queue[qtail++] = nb;
            }
        }
    }
    if (status == ORC_SOLVED && goal_reached < 0) status = ORC_NO_SOLUTION_FOUND; /* prm.rs:304 */
    if (status == ORC_SOLVED) {
        /* prm.rs:189-208: [start] ++ reverse(goal ... root connection) */
        uint32_t len = 1;
        for (int64_t c = goal_reached; c >= 0; c = parent[c]) len++;
        free(p->path);
        p->path = (double*)malloc(sizeof(double) * (size_t)len * p->dim);
        memcpy(p->path, p->start, sizeof(double) * p->dim);
        uint32_t pos = len - 1;
        for (int64_t c = goal_reached; c >= 0; c = parent[c], --pos)
            memcpy(p->path + (size_t)pos * p->dim, p->roadmap[c].values, sizeof(double) * p->dim);
        p->path_len = len;
    }
    free(queue); free(parent); free(visited); free(is_goal);
    return status;
}

uint32_t orc_prm_get_path(const orc_prm* p, double* out, uint32_t cap) {
    if (out && p->path_len <= cap && p->path_len)
        memcpy(out, p->path, sizeof(double) * (size_t)p->path_len * p->dim);
    return p->path_len;
}

uint32_t orc_prm_get_start_connections(const orc_prm* p, uint32_t* out, uint32_t cap) {
    if (out && p->n_start_conn <= cap && p->n_start_conn) memcpy(out, p->start_conn, sizeof(uint32_t) * p->n_start_conn);
    return p->n_start_conn;
}

uint32_t orc_prm_get_goal_indices(const orc_prm* p, uint32_t* out, uint32_t cap) {
    if (out && p->n_goal_idx <= cap && p->n_goal_idx) memcpy(out, p->goal_idx, sizeof(uint32_t) * p->n_goal_idx);
    return p->n_goal_idx;
}
