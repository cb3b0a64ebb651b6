"""ctypes binding of include/oxmpl_hip.h (the C ABI of liboxmpl_hip.so).

Mirrors the header one to one; numpy arrays in, numpy arrays out.  Nothing here computes:
every call goes into the HIP library, and a missing library or GPU raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# OXMPL_HIP_LIB: development override (kernel tuning experiments load alternative builds)
_LIB = os.environ.get("OXMPL_HIP_LIB") or os.path.join(_PKG, "lib", "liboxmpl_hip.so")

MAX_DIM = 8
OK, ERR_TIMEOUT, ERR_NO_SOLUTION_FOUND, ERR_PLANNER_UNINITIALISED = 0, 1, 2, 3
ERR_INVALID_START_STATE, ERR_UNSAMPLED_STATE_SPACE = 4, 5
ERR_BAD_ARG, ERR_UNBOUNDED, ERR_ZERO_VOLUME, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE = 16, 17, 18, 19, 32, 33
STOP_NONE, STOP_GOAL, STOP_ITERATIONS, STOP_NODES, STOP_TIMEOUT = -1, 0, 1, 2, 3
KERNEL_AUTO, KERNEL_STREAM, KERNEL_RESIDENT, KERNEL_LANES, KERNEL_CELLS = 0, 1, 2, 5, 6   # (3 and 4 were retired with ABI version 2)
GOAL_SAMPLE_CENTRE, GOAL_SAMPLE_UNIFORM_DISC = 0, 1
# oxhip_debug_flag: test-only switches that force rarely taken code paths (results identical by construction)
DEBUG_PAIR_TO_WHOLE_TREE, DEBUG_AUDIT, DEBUG_ALL_WHOLE_TREE, DEBUG_ONE_LANE_ROUNDS, DEBUG_SHORT_MEMO = 1, 2, 4, 8, 16
DEBUG_STAR_TWO_PASS, DEBUG_STAR_ONE_SEGMENT = 32, 64
DEBUG_SE2_NO_SEGMENT_GRID = 128
DEBUG_SE2_SMALL_LDS = 256
ABI_VERSION = 2
STAMP_WORDS = 64
PLANNER_RRT, PLANNER_RRT_CONNECT, PLANNER_RRT_STAR = 0, 1, 2
SPACE_REAL_VECTOR, SPACE_SE2 = 0, 1

# every symbol include/oxmpl_hip.h declares (tests check the library exports them all)
EXPORTS = [
    "oxhip_abi_version", "oxhip_status_string", "oxhip_last_error_string", "oxhip_device_count",
    "oxhip_rrt_batch_create", "oxhip_rrt_batch_destroy", "oxhip_rrt_batch_set_spheres",
    "oxhip_rrt_batch_set_boxes", "oxhip_rrt_batch_set_segments", "oxhip_rrt_batch_setup", "oxhip_rrt_batch_set_tree", "oxhip_rrt_batch_solve",
    "oxhip_rrt_batch_get_counts", "oxhip_rrt_batch_get_tree", "oxhip_rrt_batch_get_path",
    "oxhip_rrt_batch_get_goal_counts", "oxhip_rrt_batch_get_goal_tree", "oxhip_rrt_batch_get_costs",
    "oxhip_rrt_batch_last_timing", "oxhip_rrt_batch_enable_stamps", "oxhip_rrt_batch_get_stamps",
    "oxhip_nn_argmin_batch", "oxhip_distance_batch",
    "oxhip_interpolate_batch", "oxhip_rrt_batch_is_valid", "oxhip_rrt_batch_check_motion",
    "oxhip_f64_op_batch", "oxhip_se2_op_batch", "oxhip_rng_u64_batch",
    "oxhip_prm_create", "oxhip_prm_destroy", "oxhip_prm_set_spheres", "oxhip_prm_set_boxes", "oxhip_prm_setup",
    "oxhip_prm_set_problem", "oxhip_prm_construct_roadmap", "oxhip_prm_get_sizes", "oxhip_prm_get_roadmap",
    "oxhip_prm_solve", "oxhip_prm_get_query_sets", "oxhip_prm_last_timing", "oxhip_prm_knn_exact_rows",
]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("dim", C.c_uint32), ("bounds", C.c_double * (2 * MAX_DIM)),
        ("max_distance", C.c_double), ("goal_bias", C.c_double), ("lvs_fraction", C.c_double),
        ("n_problems", C.c_uint32), ("max_nodes", C.c_uint32), ("stop_at_goal", C.c_uint32),
        ("kernel", C.c_uint32), ("seed", C.c_uint64), ("first_problem_id", C.c_uint64),
        ("device", C.c_int32), ("planner", C.c_uint32), ("search_radius", C.c_double),
        ("space", C.c_uint32), ("goal_sampler", C.c_uint32), ("debug_flags", C.c_uint32), ("star_pool_share", C.c_uint32),
        ("frozen_split", C.c_uint32), ("reserved", C.c_uint32),
    ]


class PrmConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("dim", C.c_uint32), ("bounds", C.c_double * (2 * MAX_DIM)),
        ("timeout", C.c_double), ("connection_radius", C.c_double), ("lvs_fraction", C.c_double),
        ("max_milestones", C.c_uint32), ("device", C.c_int32), ("max_samples", C.c_uint64),
        ("seed", C.c_uint64), ("stream", C.c_uint64), ("knn_k", C.c_uint32), ("reserved", C.c_uint32),
    ]


class OxhipError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("oxhip status %d (%s)%s" % (status, status_string(status), (": " + detail) if detail else ""))


def library_path():
    return _LIB


def build_library(force=False):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree (oxmpl_amd/lib)."""
    csrc = os.path.join(_PKG, "csrc")
    cmd = ["make", "-s", "-C", csrc, "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


_lib = None
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def lib():
    """Load liboxmpl_hip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C oxmpl_amd/csrc). There is no CPU fallback." % _LIB)
        L = C.CDLL(_LIB)
        L.oxhip_abi_version.restype = C.c_int32
        L.oxhip_status_string.restype = C.c_char_p
        L.oxhip_status_string.argtypes = [C.c_int32]
        L.oxhip_last_error_string.restype = C.c_char_p
        L.oxhip_device_count.argtypes = [_i32p]
        L.oxhip_rrt_batch_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.oxhip_rrt_batch_destroy.argtypes = [C.c_void_p]
        L.oxhip_rrt_batch_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.oxhip_rrt_batch_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.oxhip_rrt_batch_set_segments.argtypes = [C.c_void_p, _dp, C.c_uint32, C.c_double]
        L.oxhip_se2_op_batch.argtypes = [C.c_int32, C.c_uint32, _dp, _dp, _dp, C.c_uint32, _dp]
        L.oxhip_rrt_batch_setup.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.oxhip_rrt_batch_set_tree.argtypes = [C.c_void_p, C.c_uint32, _dp, _i32p, C.c_uint32]
        L.oxhip_rrt_batch_solve.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_uint32, _i32p]
        L.oxhip_rrt_batch_get_counts.argtypes = [C.c_void_p, _u64p, _u32p, _u64p, _u64p, _i32p, _i32p]
        L.oxhip_rrt_batch_get_tree.argtypes = [C.c_void_p, C.c_uint32, _dp, _i32p, C.c_uint32, _u32p]
        L.oxhip_rrt_batch_get_path.argtypes = [C.c_void_p, C.c_uint32, _dp, C.c_uint32, _u32p]
        L.oxhip_rrt_batch_get_goal_counts.argtypes = [C.c_void_p, _u32p, _i32p]
        L.oxhip_rrt_batch_get_goal_tree.argtypes = [C.c_void_p, C.c_uint32, _dp, _i32p, C.c_uint32, _u32p]
        L.oxhip_rrt_batch_get_costs.argtypes = [C.c_void_p, C.c_uint32, _dp, C.c_uint32, _u32p]
        L.oxhip_rrt_batch_last_timing.argtypes = [C.c_void_p, _dp, _u32p, _u32p]
        L.oxhip_rrt_batch_enable_stamps.argtypes = [C.c_void_p, C.c_uint32]
        L.oxhip_rrt_batch_get_stamps.argtypes = [C.c_void_p, _u64p, C.c_uint32]
        L.oxhip_nn_argmin_batch.argtypes = [C.c_int32, C.c_uint32, _dp, _u32p, C.c_uint32, _dp, _u32p, _dp]
        L.oxhip_distance_batch.argtypes = [C.c_int32, C.c_uint32, _dp, _dp, C.c_uint32, _dp]
        L.oxhip_interpolate_batch.argtypes = [C.c_int32, C.c_uint32, _dp, _dp, _dp, C.c_uint32, _dp]
        L.oxhip_rrt_batch_is_valid.argtypes = [C.c_void_p, _dp, C.c_uint32, _u8p]
        L.oxhip_rrt_batch_check_motion.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32, _u8p]
        L.oxhip_f64_op_batch.argtypes = [C.c_int32, C.c_uint32, _dp, _dp, _dp, C.c_uint32, _dp]
        L.oxhip_rng_u64_batch.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, _u64p]
        L.oxhip_prm_create.argtypes = [C.POINTER(PrmConfig), C.POINTER(C.c_void_p)]
        L.oxhip_prm_destroy.argtypes = [C.c_void_p]
        L.oxhip_prm_set_spheres.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.oxhip_prm_set_boxes.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32]
        L.oxhip_prm_setup.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.oxhip_prm_set_problem.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.oxhip_prm_construct_roadmap.argtypes = [C.c_void_p]
        L.oxhip_prm_get_sizes.argtypes = [C.c_void_p, _u32p, _u64p, _u64p]
        L.oxhip_prm_get_roadmap.argtypes = [C.c_void_p, _dp, C.c_uint32, _u64p, _u32p, C.c_uint64]
        L.oxhip_prm_solve.argtypes = [C.c_void_p, C.c_double, _dp, C.c_uint32, _u32p]
        L.oxhip_prm_get_query_sets.argtypes = [C.c_void_p, _u32p, C.c_uint32, _u32p, _u32p, C.c_uint32, _u32p]
        L.oxhip_prm_last_timing.argtypes = [C.c_void_p, _dp, _u64p, _u32p]
        L.oxhip_prm_knn_exact_rows.argtypes = [C.c_void_p, _u32p]
        for name in EXPORTS:
            if name not in ("oxhip_status_string", "oxhip_last_error_string"):
                getattr(L, name).restype = C.c_int32
        _lib = L
    return _lib


def status_string(status):
    try:
        return lib().oxhip_status_string(status).decode()
    except OSError:
        return "?"


def _check(status):
    if status != OK:
        raise OxhipError(status, lib().oxhip_last_error_string().decode())


def device_count():
    n = C.c_int32()
    _check(lib().oxhip_device_count(C.byref(n)))
    return n.value


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


class RRTBatch:
    """P independent oxmpl RRT planners (RealVectorStateSpace, ball goal, sphere/box validity)
    grown on one GPU.  Thin wrapper of the oxhip_rrt_batch_* entry points."""

    def __init__(self, dim, bounds, max_distance, goal_bias, n_problems, max_nodes=10000,
                 lvs_fraction=0.05, stop_at_goal=True, seed=0, first_problem_id=0, device=0,
                 kernel=KERNEL_AUTO, planner=PLANNER_RRT, search_radius=0.0, space=SPACE_REAL_VECTOR,
                 goal_sampler=GOAL_SAMPLE_CENTRE, debug_flags=0, star_pool_share=0, frozen_split=0):
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.dim = dim
        b = _f64(bounds).reshape(-1)
        if b.size != 2 * dim:
            raise OxhipError(ERR_BAD_ARG, "bounds must hold dim (lo,hi) pairs")  # StateSpaceError::DimensionMismatch
        for i, v in enumerate(b[:2 * MAX_DIM]):  # dim > MAX_DIM is rejected by the library
            cfg.bounds[i] = v
        cfg.max_distance, cfg.goal_bias, cfg.lvs_fraction = max_distance, goal_bias, lvs_fraction
        cfg.n_problems, cfg.max_nodes = n_problems, max_nodes
        cfg.stop_at_goal, cfg.kernel = int(bool(stop_at_goal)), kernel
        cfg.seed, cfg.first_problem_id, cfg.device = seed, first_problem_id, device
        cfg.planner = planner
        cfg.search_radius = search_radius
        cfg.space = space
        cfg.goal_sampler, cfg.debug_flags, cfg.star_pool_share = goal_sampler, debug_flags, star_pool_share
        cfg.frozen_split = frozen_split
        self.planner = planner
        self.dim, self.n_problems, self.max_nodes = dim, n_problems, max_nodes
        self._h = C.c_void_p()
        _check(lib().oxhip_rrt_batch_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().oxhip_rrt_batch_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_spheres(self, centres, radii):
        r = _f64(radii).reshape(-1)
        c = _f64(centres, (r.size, self.dim))
        _check(lib().oxhip_rrt_batch_set_spheres(self._h, _p(c), _p(r), r.size))

    def set_boxes(self, lo, hi):
        lo = _f64(lo).reshape(-1, self.dim)
        hi = _f64(hi, lo.shape)
        _check(lib().oxhip_rrt_batch_set_boxes(self._h, _p(lo), _p(hi), lo.shape[0]))

    def set_segments(self, segments, clearance):
        """SE(2) batches: the segment-soup checker (disc robot of radius `clearance`)"""
        s = _f64(segments).reshape(-1, 4)
        _check(lib().oxhip_rrt_batch_set_segments(self._h, _p(s), s.shape[0], clearance))

    def setup(self, starts, goal_centres, goal_radii):
        P = self.n_problems
        s = np.ascontiguousarray(np.broadcast_to(_f64(starts).reshape(-1, self.dim), (P, self.dim)))
        g = np.ascontiguousarray(np.broadcast_to(_f64(goal_centres).reshape(-1, self.dim), (P, self.dim)))
        r = np.ascontiguousarray(np.broadcast_to(_f64(goal_radii).reshape(-1), (P,)))
        _check(lib().oxhip_rrt_batch_setup(self._h, _p(s), _p(g), _p(r)))

    def set_tree(self, problem, states, parents):
        s = _f64(states).reshape(-1, self.dim)
        par = np.ascontiguousarray(parents, dtype=np.int32)
        _check(lib().oxhip_rrt_batch_set_tree(self._h, problem, _p(s), _p(par, _i32p), s.shape[0]))

    def solve(self, max_iterations, timeout_s=0.0, freeze=False):
        st = np.empty(self.n_problems, dtype=np.int32)
        _check(lib().oxhip_rrt_batch_solve(self._h, int(max_iterations), float(timeout_s), int(bool(freeze)),
                                           _p(st, _i32p)))
        return st

    def counts(self):
        P = self.n_problems
        out = dict(iterations=np.empty(P, np.uint64), nodes=np.empty(P, np.uint32), accepted=np.empty(P, np.uint64),
                   checksum=np.empty(P, np.uint64), goal_node=np.empty(P, np.int32), stop_reason=np.empty(P, np.int32))
        _check(lib().oxhip_rrt_batch_get_counts(self._h, _p(out["iterations"], _u64p), _p(out["nodes"], _u32p),
                                                _p(out["accepted"], _u64p), _p(out["checksum"], _u64p),
                                                _p(out["goal_node"], _i32p), _p(out["stop_reason"], _i32p)))
        return out

    def tree(self, problem):
        n = C.c_uint32()
        st = lib().oxhip_rrt_batch_get_tree(self._h, problem, None, None, 0, C.byref(n))
        if st not in (OK, ERR_CAPACITY):
            _check(st)
        states = np.empty((n.value, self.dim), dtype=np.float64)
        parents = np.empty(n.value, dtype=np.int32)
        _check(lib().oxhip_rrt_batch_get_tree(self._h, problem, _p(states), _p(parents, _i32p), n.value, C.byref(n)))
        return states, parents

    def costs(self, problem):
        """RRT*: Node::cost of every node (rrt_star.rs:26)"""
        n = C.c_uint32()
        _check(lib().oxhip_rrt_batch_get_costs(self._h, problem, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        _check(lib().oxhip_rrt_batch_get_costs(self._h, problem, _p(out), n.value, C.byref(n)))
        return out

    def goal_counts(self):
        """RRTConnect: goal-tree sizes and the last goal-tree node of each solution"""
        P = self.n_problems
        nodes, end = np.empty(P, np.uint32), np.empty(P, np.int32)
        _check(lib().oxhip_rrt_batch_get_goal_counts(self._h, _p(nodes, _u32p), _p(end, _i32p)))
        return dict(nodes=nodes, end_node=end)

    def goal_tree(self, problem):
        n = C.c_uint32()
        st = lib().oxhip_rrt_batch_get_goal_tree(self._h, problem, None, None, 0, C.byref(n))
        if st not in (OK, ERR_CAPACITY):
            _check(st)
        states = np.empty((n.value, self.dim), dtype=np.float64)
        parents = np.empty(n.value, dtype=np.int32)
        _check(lib().oxhip_rrt_batch_get_goal_tree(self._h, problem, _p(states), _p(parents, _i32p), n.value, C.byref(n)))
        return states, parents

    def path(self, problem):
        ln = C.c_uint32()
        st = lib().oxhip_rrt_batch_get_path(self._h, problem, None, 0, C.byref(ln))
        if st not in (OK, ERR_CAPACITY):
            _check(st)
        out = np.empty((ln.value, self.dim), dtype=np.float64)
        if ln.value:
            _check(lib().oxhip_rrt_batch_get_path(self._h, problem, _p(out), ln.value, C.byref(ln)))
        return out

    def last_timing(self):
        ms, launches, kind = C.c_double(), C.c_uint32(), C.c_uint32()
        _check(lib().oxhip_rrt_batch_last_timing(self._h, C.byref(ms), C.byref(launches), C.byref(kind)))
        return dict(kernel_ms=ms.value, launches=launches.value, kernel=kind.value)

    def enable_stamps(self, enable=True):
        _check(lib().oxhip_rrt_batch_enable_stamps(self._h, int(bool(enable))))

    def stamps(self):
        out = np.zeros(STAMP_WORDS, dtype=np.uint64)
        _check(lib().oxhip_rrt_batch_get_stamps(self._h, _p(out, _u64p), STAMP_WORDS))
        return out

    def is_valid(self, states):
        s = _f64(states).reshape(-1, self.dim)
        out = np.empty(s.shape[0], dtype=np.uint8)
        _check(lib().oxhip_rrt_batch_is_valid(self._h, _p(s), s.shape[0], _p(out, _u8p)))
        return out.astype(bool)

    def check_motion(self, frm, to):
        a = _f64(frm).reshape(-1, self.dim)
        b = _f64(to, a.shape)
        out = np.empty(a.shape[0], dtype=np.uint8)
        _check(lib().oxhip_rrt_batch_check_motion(self._h, _p(a), _p(b), a.shape[0], _p(out, _u8p)))
        return out.astype(bool)


def nn_argmin_batch(trees, queries, device=0):
    """trees: list of [n_i, dim] arrays; queries: [Q, dim] -> (index[Q], min_dist[Q])  (rrt.rs:187-196)"""
    q = _f64(queries)
    dim = q.shape[1]
    n = np.array([t.shape[0] for t in trees], dtype=np.uint32)
    nodes = _f64(np.concatenate([_f64(t).reshape(-1, dim) for t in trees], axis=0))
    idx = np.empty(len(trees), dtype=np.uint32)
    md = np.empty(len(trees), dtype=np.float64)
    _check(lib().oxhip_nn_argmin_batch(device, dim, _p(nodes), _p(n, _u32p), len(trees), _p(q), _p(idx, _u32p), _p(md)))
    return idx, md


def distance_batch(a, b, device=0):
    a = _f64(a)
    b = _f64(b, a.shape)
    out = np.empty(a.shape[0], dtype=np.float64)
    _check(lib().oxhip_distance_batch(device, a.shape[1], _p(a), _p(b), a.shape[0], _p(out)))
    return out


def interpolate_batch(frm, to, t, device=0):
    a = _f64(frm)
    b = _f64(to, a.shape)
    t = _f64(t).reshape(-1)
    out = np.empty_like(a)
    _check(lib().oxhip_interpolate_batch(device, a.shape[1], _p(a), _p(b), _p(t), a.shape[0], _p(out)))
    return out


def f64_op_batch(op, a, b=None, c=None, device=0):
    a = _f64(a).reshape(-1)
    b = None if b is None else _f64(b, a.shape)
    c = None if c is None else _f64(c, a.shape)
    out = np.empty_like(a)
    _check(lib().oxhip_f64_op_batch(device, op, _p(a), None if b is None else _p(b), None if c is None else _p(c),
                                    a.size, _p(out)))
    return out


def rng_u64_batch(seed, stream, n, device=0):
    out = np.empty(n, dtype=np.uint64)
    _check(lib().oxhip_rng_u64_batch(device, seed, stream, n, _p(out, _u64p)))
    return out


def se2_op_batch(op, a, b, t=None, device=0):
    """op 0: rows (se2_distance, so2_normalise(a.theta), so2_distance(a.theta, b.theta)); op 1: se2_interpolate"""
    a, b = _f64(a).reshape(-1, 3), _f64(b).reshape(-1, 3)
    out = np.empty_like(a)
    tt = None if t is None else _f64(t).reshape(-1)
    _check(lib().oxhip_se2_op_batch(device, op, _p(a), _p(b), None if tt is None else _p(tt), a.shape[0], _p(out)))
    return out


class PRMRoadmap:
    """oxmpl's PRM (prm.rs) on one GPU: roadmap construction and queries.  Thin wrapper of oxhip_prm_*."""

    def __init__(self, dim, bounds, connection_radius, max_milestones, timeout=0.0, lvs_fraction=0.05,
                 max_samples=0, seed=0, stream=0, device=0, knn_k=0):
        cfg = PrmConfig()
        cfg.struct_size = C.sizeof(PrmConfig)
        cfg.dim = dim
        b = _f64(bounds).reshape(-1)
        if b.size != 2 * dim:
            raise OxhipError(ERR_BAD_ARG, "bounds must hold dim (lo,hi) pairs")
        for i, v in enumerate(b[:2 * MAX_DIM]):
            cfg.bounds[i] = v
        cfg.timeout, cfg.connection_radius, cfg.lvs_fraction = timeout, connection_radius, lvs_fraction
        cfg.max_milestones, cfg.device, cfg.max_samples = max_milestones, device, max_samples
        cfg.seed, cfg.stream = seed, stream
        cfg.knn_k = knn_k   # 0: radius connection (the reference); k > 0: connect to the k nearest earlier milestones
        self.dim = dim
        self._h = C.c_void_p()
        _check(lib().oxhip_prm_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().oxhip_prm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_spheres(self, centres, radii):
        r = _f64(radii).reshape(-1)
        c = _f64(centres, (r.size, self.dim))
        _check(lib().oxhip_prm_set_spheres(self._h, _p(c), _p(r), r.size))

    def set_boxes(self, lo, hi):
        lo = _f64(lo).reshape(-1, self.dim)
        hi = _f64(hi, lo.shape)
        _check(lib().oxhip_prm_set_boxes(self._h, _p(lo), _p(hi), lo.shape[0]))

    def setup(self, start, goal_centre, goal_radius):
        s, g = _f64(start, (self.dim,)), _f64(goal_centre, (self.dim,))
        _check(lib().oxhip_prm_setup(self._h, _p(s), _p(g), goal_radius))

    def set_problem(self, start, goal_centre, goal_radius):
        s, g = _f64(start, (self.dim,)), _f64(goal_centre, (self.dim,))
        _check(lib().oxhip_prm_set_problem(self._h, _p(s), _p(g), goal_radius))

    def construct_roadmap(self):
        _check(lib().oxhip_prm_construct_roadmap(self._h))

    def sizes(self):
        """(milestones, edge entries = 2 x undirected edges, samples drawn)"""
        n, e, s = C.c_uint32(), C.c_uint64(), C.c_uint64()
        _check(lib().oxhip_prm_get_sizes(self._h, C.byref(n), C.byref(e), C.byref(s)))
        return n.value, e.value, s.value

    def roadmap(self):
        """(states [n][dim], offsets [n+1] u64, neighbours [E] u32)"""
        n, e, _ = self.sizes()
        states = np.zeros((n, self.dim), dtype=np.float64)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        nbrs = np.zeros(max(e, 1), dtype=np.uint32)
        _check(lib().oxhip_prm_get_roadmap(self._h, _p(states), n, _p(offsets, _u64p), _p(nbrs, _u32p), e))
        return states, offsets, nbrs[:e]

    def solve(self, timeout_s=0.0):
        """Planner::solve: returns (status, path [len][dim]); the status codes mirror PlanningError"""
        ln = C.c_uint32()
        st = lib().oxhip_prm_solve(self._h, timeout_s, None, 0, C.byref(ln))
        if st != OK:
            if st in (ERR_TIMEOUT, ERR_NO_SOLUTION_FOUND, ERR_PLANNER_UNINITIALISED, ERR_INVALID_START_STATE,
                      ERR_UNSAMPLED_STATE_SPACE):
                return st, np.zeros((0, self.dim))
            _check(st)
        path = np.zeros((ln.value, self.dim), dtype=np.float64)
        _check(lib().oxhip_prm_solve(self._h, timeout_s, _p(path), ln.value, C.byref(ln)))
        return OK, path

    def query_sets(self):
        ns, ng = C.c_uint32(), C.c_uint32()
        _check(lib().oxhip_prm_get_query_sets(self._h, None, 0, C.byref(ns), None, 0, C.byref(ng)))
        sc = np.zeros(max(ns.value, 1), dtype=np.uint32)
        gi = np.zeros(max(ng.value, 1), dtype=np.uint32)
        _check(lib().oxhip_prm_get_query_sets(self._h, _p(sc, _u32p), ns.value, C.byref(ns), _p(gi, _u32p), ng.value,
                                              C.byref(ng)))
        return sc[:ns.value], gi[:ng.value]

    def knn_exact_rows(self):
        """k-nearest variant: rows of the last construct_roadmap that were searched exactly (their candidate radius fell short)"""
        r = C.c_uint32()
        _check(lib().oxhip_prm_knn_exact_rows(self._h, C.byref(r)))
        return r.value

    def last_timing(self):
        """dict(phase_ms=[sample, pairs, edges, sort+csr, query, bfs], candidates, redraw_batches)"""
        ms = np.zeros(6, dtype=np.float64)
        c, r = C.c_uint64(), C.c_uint32()
        _check(lib().oxhip_prm_last_timing(self._h, _p(ms), C.byref(c), C.byref(r)))
        return dict(phase_ms=[float(v) for v in ms], candidates=c.value, redraw_batches=r.value)
