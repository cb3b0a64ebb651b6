"""Test-side helpers for the PRM parity tests (may use the oracle)."""
import numpy as np

from helpers import params_spheres, params_boxes

FNV_P = 0x100000001B3
MASK = (1 << 64) - 1
STATUS_NAME = {0: "solved", 1: "timeout", 2: "no_solution", 3: "uninitialised", 4: "invalid_start", 5: "unsampled"}


def csr_checksum(offsets, nbrs):
    """FNV-1a over (degree, neighbours...) per node, as tests/golden/make_golden_prm.py"""
    h = 0xCBF29CE484222325
    for i in range(len(offsets) - 1):
        a, b = int(offsets[i]), int(offsets[i + 1])
        h = ((h ^ (b - a)) * FNV_P) & MASK
        for v in nbrs[a:b]:
            h = ((h ^ int(v)) * FNV_P) & MASK
    return h


def states_checksum(states):
    h = 0xCBF29CE484222325
    for v in np.ascontiguousarray(states, dtype=np.float64).view(np.uint64).ravel():
        h = ((h ^ int(v)) * FNV_P) & MASK
    return h


def make_oracle_prm(P, **kw):
    from oracle import oracle_py as orc
    o = orc.OraclePRM(P["dim"], P["bounds"], P["radius"], lvs_fraction=P["fraction"], seed=P["seed"],
                      stream=P["stream"], **kw)
    if P["spheres"]:
        c, r = params_spheres(P)
        o.set_spheres(c, r)
    if P["boxes"]:
        lo, hi = params_boxes(P)
        o.set_boxes(lo, hi)
    if P.get("knn_k"):
        o.set_knn(P["knn_k"])
    return o
