"""Shared helpers for the parity tests (test-side only; may use the oracle)."""
import struct

import numpy as np


def unhex(h):
    return struct.unpack("<d", struct.pack("<Q", int(h, 16)))[0]


def hexf(v):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", float(v)))[0]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def params_spheres(params):
    """golden 'params' -> (centres[n,dim], radii[n]); config2 stores hex strings."""
    cs, rs = [], []
    for c, r in params["spheres"]:
        cs.append([unhex(v) if isinstance(v, str) else v for v in c])
        rs.append(unhex(r) if isinstance(r, str) else r)
    dim = params["dim"]
    return np.array(cs, dtype=np.float64).reshape(-1, dim), np.array(rs, dtype=np.float64)


def params_boxes(params):
    dim = params["dim"]
    lo = np.array([b[0] for b in params["boxes"]], dtype=np.float64).reshape(-1, dim)
    hi = np.array([b[1] for b in params["boxes"]], dtype=np.float64).reshape(-1, dim)
    return lo, hi


def is_path_valid(path, bounds, lvs_fraction, is_valid, extent_fn, num_steps_fn, interp_fn, dist_fn):
    """The reference's own validator (oxmpl/tests/rrt_rvss_tests.rs:72-107): every vertex valid,
    every edge valid when discretised at lvsl (10x coarser than the planner's own check)."""
    lvsl = extent_fn(bounds) * lvs_fraction
    for i in range(len(path) - 1):
        a, b = path[i], path[i + 1]
        if not is_valid(a):
            return False
        if i + 1 == len(path) - 1 and not is_valid(b):
            return False
        d = dist_fn(a, b)
        n = int(np.ceil(d / lvsl))
        if n > 1:
            for j in range(1, n + 1):
                if not is_valid(interp_fn(a, b, float(j) / float(n))):
                    return False
    return True
