"""ctypes binding of oracle/liboxmpl_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (oxmpl_amd/) never does.  PARITY UNPINNED: see
oracle/rrt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboxmpl_oracle.so")

SOLVED, TIMEOUT, NO_SOLUTION_FOUND, PLANNER_UNINITIALISED = 0, 1, 2, 3
INVALID_START_STATE, UNSAMPLED_STATE_SPACE = 4, 5
BAD_ARG, UNBOUNDED, ZERO_VOLUME = 16, 17, 18
STOP_GOAL, STOP_ITERATIONS, STOP_NODES, STOP_TIMEOUT = 0, 1, 2, 3


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rrt_oracle.c", "rrt_oracle.h", "prm_oracle.c", "prm_oracle.h", "se2_oracle.c", "se2_oracle.h", "Makefile")]
    if (not force) and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_chacha_block.argtypes = [C.POINTER(C.c_uint32), C.c_uint64, C.c_uint64, C.c_int,
                                       C.POINTER(C.c_uint32)]
        L.orc_rng_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_rng_next_u64.argtypes = [C.c_void_p]
        L.orc_rng_next_u64.restype = C.c_uint64
        L.orc_random_bool.argtypes = [C.c_void_p, C.c_double]
        L.orc_random_range.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_random_range.restype = C.c_double
        L.orc_bernoulli_p_int.argtypes = [C.c_double]
        L.orc_bernoulli_p_int.restype = C.c_uint64
        L.orc_distance.argtypes = [_dp, _dp, C.c_uint32]
        L.orc_distance.restype = C.c_double
        L.orc_interpolate.argtypes = [_dp, _dp, C.c_double, _dp, C.c_uint32]
        L.orc_maximum_extent.argtypes = [_dp, C.c_uint32]
        L.orc_maximum_extent.restype = C.c_double
        L.orc_clamp_fraction.argtypes = [C.c_double]
        L.orc_clamp_fraction.restype = C.c_double
        L.orc_num_steps.argtypes = [C.c_double, C.c_double]
        L.orc_num_steps.restype = C.c_uint64
        L.orc_rrt_new.argtypes = [C.c_uint32, _dp, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                  C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_rrt_new.restype = C.c_void_p
        L.orc_rrt_free.argtypes = [C.c_void_p]
        L.orc_rrt_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrt_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrt_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_rrt_set_goal_sampler.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_sincos_libm.argtypes = [C.c_int]
        L.orc_sincos.argtypes = [C.c_double, _dp, _dp]
        L.orc_rrt_set_tree.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int32), C.c_uint32]
        L.orc_rrt_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_double]
        L.orc_rrt_num_nodes.argtypes = [C.c_void_p]
        L.orc_rrt_num_nodes.restype = C.c_uint32
        for name in ("orc_rrt_iterations", "orc_rrt_checksum", "orc_rrt_accepted"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        L.orc_rrt_goal_node.argtypes = [C.c_void_p]
        L.orc_rrt_goal_node.restype = C.c_int32
        L.orc_rrt_stop_reason.argtypes = [C.c_void_p]
        L.orc_rrt_stop_reason.restype = C.c_int32
        L.orc_rrt_get_tree.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int32)]
        L.orc_rrt_get_path.argtypes = [C.c_void_p, _dp, C.c_uint32]
        L.orc_rrt_get_path.restype = C.c_uint32
        L.orc_rrt_check_motion.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_rrt_is_valid.argtypes = [C.c_void_p, _dp]
        L.orc_nearest.argtypes = [_dp, C.c_uint32, C.c_uint32, _dp, _dp]
        L.orc_nearest.restype = C.c_uint32
        L.orc_rrtc_new.argtypes = [C.c_uint32, _dp, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_int)]
        L.orc_rrtc_new.restype = C.c_void_p
        L.orc_rrtc_free.argtypes = [C.c_void_p]
        L.orc_rrtc_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrtc_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrtc_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_rrtc_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
        L.orc_rrtc_num_nodes.argtypes = [C.c_void_p, C.c_int]
        L.orc_rrtc_num_nodes.restype = C.c_uint32
        for name in ("orc_rrtc_iterations", "orc_rrtc_checksum"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        L.orc_rrtc_end_node.argtypes = [C.c_void_p, C.c_int]
        L.orc_rrtc_end_node.restype = C.c_int32
        L.orc_rrtc_stop_reason.argtypes = [C.c_void_p]
        L.orc_rrtc_stop_reason.restype = C.c_int32
        L.orc_rrtc_get_tree.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int32)]
        L.orc_rrtc_get_path.argtypes = [C.c_void_p, _dp, C.c_uint32]
        L.orc_rrtc_get_path.restype = C.c_uint32
        L.orc_rrt_solve_many.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64, C.c_int,
                                         C.c_uint32]
        L.orc_rrts_new.argtypes = [C.c_uint32, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int,
                                   C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_rrts_new.restype = C.c_void_p
        L.orc_rrts_free.argtypes = [C.c_void_p]
        L.orc_rrts_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrts_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_rrts_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_rrts_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
        L.orc_rrts_base.argtypes = [C.c_void_p]
        L.orc_rrts_base.restype = C.c_void_p
        L.orc_rrts_get_costs.argtypes = [C.c_void_p, _dp]
        L.orc_rrts_get_costs.restype = None
        for name, nargs in (("orc_so2_normalise", 1), ("orc_so2_distance", 2), ("orc_so2_interpolate", 3)):
            getattr(L, name).argtypes = [C.c_double] * nargs
            getattr(L, name).restype = C.c_double
        L.orc_se2_distance.argtypes = [_dp, _dp]
        L.orc_se2_distance.restype = C.c_double
        L.orc_se2_interpolate.argtypes = [_dp, _dp, C.c_double, _dp]
        L.orc_se2_interpolate.restype = None
        L.orc_se2_extent.argtypes = [_dp]
        L.orc_se2_extent.restype = C.c_double
        L.orc_point_segment_distance.argtypes = [C.c_double, C.c_double, _dp]
        L.orc_point_segment_distance.restype = C.c_double
        L.orc_se2c_new.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint64,
                                   C.c_uint64, C.POINTER(C.c_int)]
        L.orc_se2c_new.restype = C.c_void_p
        L.orc_se2c_free.argtypes = [C.c_void_p]
        L.orc_se2c_set_segments.argtypes = [C.c_void_p, _dp, C.c_uint32, C.c_double]
        L.orc_se2c_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_se2c_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
        L.orc_se2c_num_nodes.argtypes = [C.c_void_p, C.c_int]
        L.orc_se2c_num_nodes.restype = C.c_uint32
        for name in ("orc_se2c_iterations", "orc_se2c_checksum"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        L.orc_se2c_end_node.argtypes = [C.c_void_p, C.c_int]
        L.orc_se2c_end_node.restype = C.c_int32
        L.orc_se2c_stop_reason.argtypes = [C.c_void_p]
        L.orc_se2c_stop_reason.restype = C.c_int32
        L.orc_se2c_get_tree.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int32)]
        L.orc_se2c_get_tree.restype = None
        L.orc_se2c_get_path.argtypes = [C.c_void_p, _dp, C.c_uint32]
        L.orc_se2c_get_path.restype = C.c_uint32
        L.orc_se2c_is_valid.argtypes = [C.c_void_p, _dp]
        L.orc_se2c_check_motion.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_se2c_theta_bounds.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_se2c_theta_bounds.restype = None
        _u32p = C.POINTER(C.c_uint32)
        L.orc_prm_new.argtypes = [C.c_uint32, _dp, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                  C.POINTER(C.c_int)]
        L.orc_prm_new.restype = C.c_void_p
        L.orc_prm_free.argtypes = [C.c_void_p]
        L.orc_prm_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_prm_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.orc_prm_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_prm_set_problem.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_prm_construct_roadmap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.orc_prm_set_knn.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_prm_num_milestones.argtypes = [C.c_void_p]
        L.orc_prm_num_milestones.restype = C.c_uint32
        for name in ("orc_prm_num_edge_entries", "orc_prm_num_samples"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        L.orc_prm_get_roadmap.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_uint64), _u32p]
        L.orc_prm_get_roadmap.restype = None
        L.orc_prm_solve.argtypes = [C.c_void_p, C.c_double]
        L.orc_prm_get_path.argtypes = [C.c_void_p, _dp, C.c_uint32]
        L.orc_prm_get_path.restype = C.c_uint32
        for name in ("orc_prm_get_start_connections", "orc_prm_get_goal_indices"):
            getattr(L, name).argtypes = [C.c_void_p, _u32p, C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class Rng:
    """ChaCha12 stream keyed (seed, stream) with rand 0.9's bool / f64-range transforms."""

    def __init__(self, seed, stream):
        self._buf = C.create_string_buffer(512)
        lib().orc_rng_seed(self._buf, seed, stream)

    def next_u64(self):
        return lib().orc_rng_next_u64(self._buf)

    def random_bool(self, p):
        return bool(lib().orc_random_bool(self._buf, p))

    def random_range(self, lo, hi):
        return lib().orc_random_range(self._buf, lo, hi)


def chacha_block(key8, counter, stream, rounds):
    k = (C.c_uint32 * 8)(*key8)
    out = (C.c_uint32 * 16)()
    lib().orc_chacha_block(k, counter, stream, rounds, out)
    return list(out)


def distance(a, b):
    a, pa = _d(a)
    b, pb = _d(b)
    return lib().orc_distance(pa, pb, a.size)


def interpolate(a, b, t):
    a, pa = _d(a)
    b, pb = _d(b)
    out = np.empty_like(a)
    lib().orc_interpolate(pa, pb, t, out.ctypes.data_as(_dp), a.size)
    return out


def maximum_extent(bounds):
    b, pb = _d(np.asarray(bounds, dtype=np.float64).reshape(-1))
    return lib().orc_maximum_extent(pb, b.size // 2)


def num_steps(dist, lvsl):
    return lib().orc_num_steps(dist, lvsl)


def nearest(nodes_aos, q):
    n, pn = _d(nodes_aos)
    qq, pq = _d(q)
    md = C.c_double()
    idx = lib().orc_nearest(pn, n.shape[0], n.shape[1], pq, C.byref(md))
    return idx, md.value


class OracleRRT:
    """One planner instance = one oxmpl RRT<RealVectorState, RealVectorStateSpace, BallGoal>."""

    def __init__(self, dim, bounds, max_distance, goal_bias, lvs_fraction=0.05, max_nodes=10000,
                 stop_at_goal=True, seed=0, problem_id=0):
        self.dim = dim
        b, pb = _d(np.asarray(bounds, dtype=np.float64).reshape(-1))
        st = C.c_int()
        self.h = lib().orc_rrt_new(dim, pb, max_distance, goal_bias, lvs_fraction, max_nodes,
                                   int(stop_at_goal), seed, problem_id, C.byref(st))
        self.create_status = st.value
        if not self.h:
            raise ValueError("orc_rrt_new failed with status %d" % st.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rrt_free(self.h)
            self.h = None

    def set_spheres(self, centres, radii):
        c, pc = _d(np.asarray(centres, dtype=np.float64).reshape(-1, self.dim))
        r, pr = _d(radii)
        lib().orc_rrt_set_spheres(self.h, pc, pr, r.size)

    def set_boxes(self, lo, hi):
        l, pl = _d(np.asarray(lo, dtype=np.float64).reshape(-1, self.dim))
        h, ph = _d(np.asarray(hi, dtype=np.float64).reshape(-1, self.dim))
        lib().orc_rrt_set_boxes(self.h, pl, ph, l.shape[0])

    def setup(self, start, goal_centre, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal_centre)
        return lib().orc_rrt_setup(self.h, ps, pg, goal_radius)

    def set_goal_sampler(self, mode):
        """0: sample_goal() = the centre (no draw); 1: uniform in the disc (rrt_rvss_tests.rs:55-66, R^2).  Also for RRT*."""
        st = lib().orc_rrt_set_goal_sampler(self.h, int(mode))
        if st != 0:
            raise ValueError("orc_rrt_set_goal_sampler failed with status %d" % st)

    def set_tree(self, states, parents):
        s, ps = _d(np.asarray(states, dtype=np.float64).reshape(-1, self.dim))
        par = np.ascontiguousarray(parents, dtype=np.int32)
        return lib().orc_rrt_set_tree(self.h, ps, par.ctypes.data_as(C.POINTER(C.c_int32)), s.shape[0])

    def solve(self, max_iterations, freeze=False, timeout_s=float("inf")):
        return lib().orc_rrt_solve(self.h, max_iterations, int(freeze), timeout_s)

    @property
    def num_nodes(self):
        return lib().orc_rrt_num_nodes(self.h)

    @property
    def iterations(self):
        return lib().orc_rrt_iterations(self.h)

    @property
    def checksum(self):
        return lib().orc_rrt_checksum(self.h)

    @property
    def accepted(self):
        return lib().orc_rrt_accepted(self.h)

    @property
    def goal_node(self):
        return lib().orc_rrt_goal_node(self.h)

    @property
    def stop_reason(self):
        return lib().orc_rrt_stop_reason(self.h)

    def tree(self):
        n = self.num_nodes
        states = np.empty((n, self.dim), dtype=np.float64)
        parents = np.empty(n, dtype=np.int32)
        lib().orc_rrt_get_tree(self.h, states.ctypes.data_as(_dp),
                               parents.ctypes.data_as(C.POINTER(C.c_int32)))
        return states, parents

    def path(self):
        cap = self.num_nodes
        out = np.empty((cap, self.dim), dtype=np.float64)
        ln = lib().orc_rrt_get_path(self.h, out.ctypes.data_as(_dp), cap)
        return out[:ln].copy()

    def check_motion(self, a, b):
        a, pa = _d(a)
        b, pb = _d(b)
        return bool(lib().orc_rrt_check_motion(self.h, pa, pb))

    def is_valid(self, p):
        p, pp = _d(p)
        return bool(lib().orc_rrt_is_valid(self.h, pp))


class OracleRRTConnect:
    """oxmpl RRTConnect<RealVectorState, RealVectorStateSpace, BallGoal> (rrt_connect.rs)"""

    def __init__(self, dim, bounds, max_distance, goal_bias, lvs_fraction=0.05, max_nodes=10000, seed=0, problem_id=0):
        self.dim = dim
        b, pb = _d(np.asarray(bounds, dtype=np.float64).reshape(-1))
        st = C.c_int()
        self.h = lib().orc_rrtc_new(dim, pb, max_distance, goal_bias, lvs_fraction, max_nodes, seed, problem_id, C.byref(st))
        if not self.h:
            raise ValueError("orc_rrtc_new failed with status %d" % st.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rrtc_free(self.h)
            self.h = None

    def set_spheres(self, centres, radii):
        c, pc = _d(np.asarray(centres, dtype=np.float64).reshape(-1, self.dim))
        r, pr = _d(radii)
        lib().orc_rrtc_set_spheres(self.h, pc, pr, r.size)

    def set_boxes(self, lo, hi):
        l, pl = _d(np.asarray(lo, dtype=np.float64).reshape(-1, self.dim))
        h, ph = _d(np.asarray(hi, dtype=np.float64).reshape(-1, self.dim))
        lib().orc_rrtc_set_boxes(self.h, pl, ph, l.shape[0])

    def setup(self, start, goal_centre, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal_centre)
        return lib().orc_rrtc_setup(self.h, ps, pg, goal_radius)

    def solve(self, max_iterations, timeout_s=float("inf")):
        return lib().orc_rrtc_solve(self.h, max_iterations, timeout_s)

    def num_nodes(self, which):
        return lib().orc_rrtc_num_nodes(self.h, which)

    iterations = property(lambda self: lib().orc_rrtc_iterations(self.h))
    checksum = property(lambda self: lib().orc_rrtc_checksum(self.h))
    stop_reason = property(lambda self: lib().orc_rrtc_stop_reason(self.h))

    def end_node(self, which):
        return lib().orc_rrtc_end_node(self.h, which)

    def tree(self, which):
        n = self.num_nodes(which)
        states = np.empty((n, self.dim), dtype=np.float64)
        parents = np.empty(n, dtype=np.int32)
        lib().orc_rrtc_get_tree(self.h, which, states.ctypes.data_as(_dp), parents.ctypes.data_as(C.POINTER(C.c_int32)))
        return states, parents

    def path(self):
        cap = self.num_nodes(0) + self.num_nodes(1)
        out = np.empty((cap, self.dim), dtype=np.float64)
        ln = lib().orc_rrtc_get_path(self.h, out.ctypes.data_as(_dp), cap)
        return out[:ln].copy()


def solve_many(planners, max_iterations, freeze=False, threads=1):
    arr = (C.c_void_p * len(planners))(*[p.h for p in planners])
    return lib().orc_rrt_solve_many(arr, len(planners), max_iterations, int(freeze), threads)


class OraclePRM:
    """oxmpl's PRM (prm.rs) restated on the CPU: construct_roadmap, get_roadmap, solve."""

    def __init__(self, dim, bounds, connection_radius, timeout=float("inf"), lvs_fraction=0.05, seed=0, stream=0):
        self.dim = dim
        b, pb = _d(np.asarray(bounds, dtype=np.float64).reshape(-1))
        st = C.c_int(0)
        self._h = lib().orc_prm_new(dim, pb, timeout, connection_radius, lvs_fraction, seed, stream, C.byref(st))
        self.status = st.value
        if not self._h:
            raise ValueError("orc_prm_new failed with status %d" % st.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_prm_free(self._h)
            self._h = None

    def set_spheres(self, centres, radii):
        c, pc = _d(np.asarray(centres, dtype=np.float64).reshape(-1))
        r, pr = _d(radii)
        lib().orc_prm_set_spheres(self._h, pc, pr, len(r))

    def set_boxes(self, lo, hi):
        l, pl = _d(np.asarray(lo, dtype=np.float64).reshape(-1))
        h, ph = _d(np.asarray(hi, dtype=np.float64).reshape(-1))
        lib().orc_prm_set_boxes(self._h, pl, ph, len(l) // self.dim)

    def setup(self, start, goal_centre, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal_centre)
        return lib().orc_prm_setup(self._h, ps, pg, goal_radius)

    def set_problem(self, start, goal_centre, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal_centre)
        return lib().orc_prm_set_problem(self._h, ps, pg, goal_radius)

    def set_knn(self, k):
        """k > 0: the k-nearest variant (every new milestone connects to its k nearest earlier ones); 0: the reference's radius rule"""
        return lib().orc_prm_set_knn(self._h, int(k))

    def construct_roadmap(self, max_milestones, max_samples=2 ** 62):
        return lib().orc_prm_construct_roadmap(self._h, max_milestones, max_samples)

    @property
    def num_milestones(self):
        return lib().orc_prm_num_milestones(self._h)

    @property
    def num_samples(self):
        return lib().orc_prm_num_samples(self._h)

    def roadmap(self):
        """(states [n][dim], offsets [n+1], neighbours [E]) with each node's `edges` in the reference's order"""
        n = self.num_milestones
        e = lib().orc_prm_num_edge_entries(self._h)
        states = np.zeros((n, self.dim), dtype=np.float64)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        nbrs = np.zeros(max(e, 1), dtype=np.uint32)
        lib().orc_prm_get_roadmap(self._h, states.ctypes.data_as(_dp), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  nbrs.ctypes.data_as(C.POINTER(C.c_uint32)))
        return states, offsets, nbrs[:e]

    def solve(self, timeout_s=float("inf")):
        return lib().orc_prm_solve(self._h, timeout_s)

    def path(self):
        n = lib().orc_prm_get_path(self._h, None, 0)
        out = np.zeros((n, self.dim), dtype=np.float64)
        if n:
            lib().orc_prm_get_path(self._h, out.ctypes.data_as(_dp), n)
        return out

    def start_connections(self):
        n = lib().orc_prm_get_start_connections(self._h, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_prm_get_start_connections(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:n]

    def goal_indices(self):
        n = lib().orc_prm_get_goal_indices(self._h, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_prm_get_goal_indices(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out[:n]


class OracleRRTStar(OracleRRT):
    """oxmpl RRTStar<RealVectorState, RealVectorStateSpace, BallGoal> (rrt_star.rs).  The tree, counters and path
    getters are OracleRRT's, reading the wrapped planner; `costs()` adds Node::cost."""

    def __init__(self, dim, bounds, max_distance, goal_bias, search_radius, lvs_fraction=0.05, max_nodes=10000,
                 stop_at_goal=True, seed=0, problem_id=0):
        self.dim = dim
        b, pb = _d(np.asarray(bounds, dtype=np.float64).reshape(-1))
        st = C.c_int()
        self.hs = lib().orc_rrts_new(dim, pb, max_distance, goal_bias, search_radius, lvs_fraction, max_nodes,
                                     int(bool(stop_at_goal)), seed, problem_id, C.byref(st))
        if not self.hs:
            raise ValueError("orc_rrts_new failed with status %d" % st.value)
        self.h = lib().orc_rrts_base(self.hs)   # borrowed: freed with hs

    def __del__(self):
        if getattr(self, "hs", None):
            lib().orc_rrts_free(self.hs)
            self.hs = None
            self.h = None

    def set_spheres(self, centres, radii):
        c, pc = _d(np.asarray(centres, dtype=np.float64).reshape(-1, self.dim))
        r, pr = _d(radii)
        lib().orc_rrts_set_spheres(self.hs, pc, pr, r.size)

    def set_boxes(self, lo, hi):
        l, pl = _d(np.asarray(lo, dtype=np.float64).reshape(-1, self.dim))
        h, ph = _d(np.asarray(hi, dtype=np.float64).reshape(-1, self.dim))
        lib().orc_rrts_set_boxes(self.hs, pl, ph, l.shape[0])

    def setup(self, start, goal_centre, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal_centre)
        return lib().orc_rrts_setup(self.hs, ps, pg, goal_radius)

    def set_tree(self, states, parents):
        raise NotImplementedError("RRT* trees carry costs; warm starts are not defined")

    def solve(self, max_iterations, timeout_s=float("inf")):
        return lib().orc_rrts_solve(self.hs, max_iterations, timeout_s)

    def costs(self):
        out = np.empty(self.num_nodes, dtype=np.float64)
        lib().orc_rrts_get_costs(self.hs, out.ctypes.data_as(_dp))
        return out


def se2_distance(a, b):
    a, pa = _d(a)
    b, pb = _d(b)
    return lib().orc_se2_distance(pa, pb)


def se2_interpolate(a, b, t):
    a, pa = _d(a)
    b, pb = _d(b)
    out = np.zeros(3)
    lib().orc_se2_interpolate(pa, pb, t, out.ctypes.data_as(_dp))
    return out


def point_segment_distance(px, py, seg):
    s, ps = _d(seg)
    return lib().orc_point_segment_distance(px, py, ps)


class OracleSE2Connect:
    """RRTConnect (rrt_connect.rs) over the build-defined SE(2) space with a segment-soup checker (oracle/se2_oracle.h)"""
    dim = 3

    def __init__(self, bounds_xy, theta_bounds, max_distance, goal_bias, lvs_fraction=0.05, max_nodes=10000, seed=0,
                 problem_id=0):
        b, pb = _d(np.asarray(bounds_xy, dtype=np.float64).reshape(-1))
        st = C.c_int()
        self.h = lib().orc_se2c_new(pb, theta_bounds[0], theta_bounds[1], max_distance, goal_bias, lvs_fraction, max_nodes,
                                    seed, problem_id, C.byref(st))
        self.create_status = st.value
        if not self.h:
            raise ValueError("orc_se2c_new failed with status %d" % st.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_se2c_free(self.h)
            self.h = None

    def set_segments(self, segs, clearance):
        s, ps = _d(np.asarray(segs, dtype=np.float64).reshape(-1, 4))
        lib().orc_se2c_set_segments(self.h, ps, s.shape[0], clearance)

    def setup(self, start, goal, goal_radius):
        s, ps = _d(start)
        g, pg = _d(goal)
        return lib().orc_se2c_setup(self.h, ps, pg, goal_radius)

    def solve(self, max_iterations, timeout_s=float("inf")):
        return lib().orc_se2c_solve(self.h, max_iterations, timeout_s)

    def num_nodes(self, which):
        return lib().orc_se2c_num_nodes(self.h, which)

    iterations = property(lambda self: lib().orc_se2c_iterations(self.h))
    checksum = property(lambda self: lib().orc_se2c_checksum(self.h))
    stop_reason = property(lambda self: lib().orc_se2c_stop_reason(self.h))

    def end_node(self, which):
        return lib().orc_se2c_end_node(self.h, which)

    def tree(self, which):
        n = self.num_nodes(which)
        states = np.empty((n, 3), dtype=np.float64)
        parents = np.empty(n, dtype=np.int32)
        lib().orc_se2c_get_tree(self.h, which, states.ctypes.data_as(_dp), parents.ctypes.data_as(C.POINTER(C.c_int32)))
        return states, parents

    def path(self):
        n = lib().orc_se2c_get_path(self.h, None, 0)
        out = np.empty((n, 3), dtype=np.float64)
        if n:
            lib().orc_se2c_get_path(self.h, out.ctypes.data_as(_dp), n)
        return out

    def is_valid(self, s):
        a, pa = _d(s)
        return bool(lib().orc_se2c_is_valid(self.h, pa))

    def check_motion(self, a, b):
        a, pa = _d(a)
        b, pb = _d(b)
        return bool(lib().orc_se2c_check_motion(self.h, pa, pb))


def sincos(x):
    """(sin, cos) as the disc goal sampler computes them (ox_sincos, or libm after set_sincos_libm(True))"""
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def set_sincos_libm(use_libm):
    lib().orc_set_sincos_libm(int(bool(use_libm)))
