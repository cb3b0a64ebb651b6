"""CPU: oracle/prm_oracle.c against tests/golden/prm_golden.json (independent numpy restatement of
oxmpl's PRM, tests/golden/make_golden_prm.py).  PARITY UNPINNED against oxmpl itself: the reference's
PRM tests (oxmpl/tests/prm_rvss_tests.rs) hold properties only, which test_prm_reference_properties
re-asserts on the oracle's output."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from helpers import unhex, bits, params_spheres, params_boxes, is_path_valid
from prm_helpers import csr_checksum, states_checksum, make_oracle_prm, STATUS_NAME


@pytest.fixture(scope="module")
def prm_knn_golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prm_knn_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("key", ["wall", "r3", "r6", "sample_cap", "knn:wall_k6", "knn:r3_k10", "knn:r6_k8", "knn:wall_k1"])
def test_prm_oracle_matches_numpy_restatement(prm_golden, prm_knn_golden, key):
    """radius connection (the reference, prm.rs:131-138) and the k-nearest variant (tests/golden/make_golden_prm_knn.py)"""
    if key.startswith("knn:"):
        prm_golden, key = prm_knn_golden, key[4:]
    P, R = prm_golden[key]["params"], prm_golden[key]["run"]
    q0 = R["queries"][0]
    o = make_oracle_prm(P)
    assert o.solve() == orc.PLANNER_UNINITIALISED                       # prm.rs:229-236
    assert o.construct_roadmap(10) == orc.PLANNER_UNINITIALISED         # prm.rs:97-104
    o.setup(q0["start"], q0["goal_c"], q0["goal_r"])
    assert o.solve() == orc.UNSAMPLED_STATE_SPACE                       # prm.rs:239-241
    assert o.construct_roadmap(P["max_milestones"], P["max_samples"]) == orc.SOLVED
    assert o.num_milestones == R["n"] and o.num_samples == R["n_samples"]
    states, off, nbrs = o.roadmap()
    assert len(nbrs) == R["edge_entries"]
    assert "%016x" % states_checksum(states) == R["states_checksum"]
    assert "%016x" % csr_checksum(off, nbrs) == R["csr_checksum"]
    head = np.array([[unhex(v) for v in row] for row in R["states_head"]])
    assert np.array_equal(bits(states[:len(head)]), bits(head))
    for i, want in enumerate(R["edges_head"]):
        assert list(nbrs[int(off[i]):int(off[i + 1])]) == want
    # every node's `edges` comes out ascending (lower neighbours at insertion, higher ones as they arrive)
    for i in range(o.num_milestones):
        seg = nbrs[int(off[i]):int(off[i + 1])]
        assert np.all(seg[1:] > seg[:-1])
    # a second construct_roadmap is a no-op (prm.rs:106-113)
    assert o.construct_roadmap(P["max_milestones"] + 100, P["max_samples"]) == orc.SOLVED
    assert o.num_milestones == R["n"]
    for q in R["queries"]:
        o.set_problem(q["start"], q["goal_c"], q["goal_r"])
        st = o.solve()
        assert STATUS_NAME[st] == q["status"]
        if q["status"] not in ("invalid_start",):
            assert list(o.start_connections()) == q["start_connections"]
            assert list(o.goal_indices()) == q["goal_indices"]
        want = np.array([[unhex(v) for v in row] for row in q["path"]]).reshape(-1, P["dim"])
        got = o.path()
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


def test_prm_reference_properties(prm_golden):
    """the assertions of oxmpl/tests/prm_rvss_tests.rs:162-204 on the wall scene"""
    P, R = prm_golden["wall"]["params"], prm_golden["wall"]["run"]
    q = R["queries"][0]
    o = make_oracle_prm(P)
    o.setup(q["start"], q["goal_c"], q["goal_r"])
    o.construct_roadmap(P["max_milestones"])
    assert o.num_milestones > 0
    assert o.solve() == orc.SOLVED
    path = o.path()
    assert len(path) > 0
    assert orc.distance(path[0], q["start"]) < 1e-9
    assert orc.distance(path[-1], q["goal_c"]) <= q["goal_r"]
    lo, hi = params_boxes(P)

    def valid(p):
        return not any(all(lo[b][k] <= p[k] <= hi[b][k] for k in range(2)) for b in range(len(lo)))

    assert is_path_valid(path, [tuple(b) for b in P["bounds"]], P["fraction"], valid, orc.maximum_extent,
                         orc.num_steps, orc.interpolate, orc.distance)
    # setup() clears the roadmap (prm.rs:224) and restarts the build-defined RNG stream
    o.setup(q["start"], q["goal_c"], q["goal_r"])
    assert o.num_milestones == 0 and o.solve() == orc.UNSAMPLED_STATE_SPACE
    o.construct_roadmap(100)
    s2, _, _ = o.roadmap()
    head = np.array([[unhex(v) for v in row] for row in R["states_head"]])
    assert np.array_equal(bits(s2[:48]), bits(head))


def test_prm_oracle_rejects_bad_spaces():
    with pytest.raises(ValueError):
        orc.OraclePRM(2, [(0.0, float("inf")), (0.0, 1.0)], 0.5)
    with pytest.raises(ValueError):
        orc.OraclePRM(2, [(1.0, 1.0), (0.0, 1.0)], 0.5)
    with pytest.raises(ValueError):
        orc.OraclePRM(0, [], 0.5)
