"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/oxmpl_hip.h
declares, and validates arguments the way the reference's constructors / rand would.
No compute call is made without a GPU (there is no CPU fallback to call)."""
import ctypes as C
import math
import os
import re

import pytest

from oxmpl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    capi.build_library()
    return capi.lib()


def test_header_and_binding_export_the_same_symbols(L):
    header = open(os.path.join(ROOT, "include", "oxmpl_hip.h")).read()
    declared = set(re.findall(r"\b(oxhip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.oxhip_abi_version() == capi.ABI_VERSION == 2
    assert C.sizeof(capi.Config) == 232  # layout of oxhip_rrt_config on the ABI
    assert C.sizeof(capi.PrmConfig) == 200  # oxhip_prm_config


def test_status_strings_cover_planning_error(L):
    # PlanningError variants in declaration order (oxmpl/src/base/error.rs:97-108)
    for code in (capi.ERR_TIMEOUT, capi.ERR_NO_SOLUTION_FOUND, capi.ERR_PLANNER_UNINITIALISED,
                 capi.ERR_INVALID_START_STATE, capi.ERR_UNSAMPLED_STATE_SPACE, capi.ERR_UNBOUNDED,
                 capi.ERR_ZERO_VOLUME, capi.ERR_NO_DEVICE):
        assert capi.status_string(code) not in ("", "unknown status")


def _create(**kw):
    args = dict(dim=2, bounds=[(0.0, 10.0)] * 2, max_distance=0.5, goal_bias=0.05, n_problems=1, max_nodes=100)
    args.update(kw)
    return capi.RRTBatch(**args)


@pytest.mark.parametrize("kw,code", [
    (dict(dim=0, bounds=[]), capi.ERR_BAD_ARG),
    (dict(dim=9, bounds=[(0.0, 1.0)] * 9), capi.ERR_BAD_ARG),
    (dict(bounds=[(0.0, math.inf), (0.0, 1.0)]), capi.ERR_UNBOUNDED),    # rvss.rs:239-241
    (dict(bounds=[(-math.inf, math.inf)] * 2), capi.ERR_UNBOUNDED),
    (dict(bounds=[(1.0, 1.0), (0.0, 1.0)]), capi.ERR_ZERO_VOLUME),       # rvss.rs:78-83,242-244
    (dict(bounds=[(2.0, 1.0), (0.0, 1.0)]), capi.ERR_ZERO_VOLUME),
    (dict(goal_bias=1.5), capi.ERR_BAD_ARG),                             # rand Bernoulli::new
    (dict(goal_bias=-0.1), capi.ERR_BAD_ARG),
    (dict(goal_bias=float("nan")), capi.ERR_BAD_ARG),
    (dict(max_distance=0.0), capi.ERR_BAD_ARG),
    (dict(lvs_fraction=0.0), capi.ERR_BAD_ARG),                          # check_motion would never end
    (dict(lvs_fraction=-3.0), capi.ERR_BAD_ARG),
    (dict(n_problems=0), capi.ERR_BAD_ARG),
    (dict(max_nodes=0), capi.ERR_BAD_ARG),
    (dict(bounds=[(0.0, 1.0)] * 3), capi.ERR_BAD_ARG),                   # StateSpaceError::DimensionMismatch
])
def test_create_rejects_what_the_reference_rejects(L, kw, code):
    with pytest.raises(capi.OxhipError) as ei:
        _create(**kw)
    assert ei.value.status == code


def test_no_device_is_a_loud_error_not_a_fallback(L):
    n = C.c_int32()
    if L.oxhip_device_count(C.byref(n)) == capi.OK:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.OxhipError) as ei:
        _create()
    assert ei.value.status == capi.ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(capi.OxhipError) as ei:
        capi.distance_batch(np.zeros((1, 2)), np.ones((1, 2)))
    assert ei.value.status == capi.ERR_NO_DEVICE
    with pytest.raises(capi.OxhipError) as ei:
        capi.nn_argmin_batch([np.zeros((3, 2))], np.ones((1, 2)))
    assert ei.value.status == capi.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under oxmpl_amd/ or include/ may reference it."""
    for base in ("oxmpl_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_scenarios_match_golden_sphere_field(golden):
    import numpy as np
    from oxmpl_amd import scenarios
    from helpers import params_spheres, bits
    c, r = params_spheres(golden["config2"]["params"])
    sc = scenarios.config2()
    assert np.array_equal(bits(c), bits(sc["spheres"][0])) and np.array_equal(bits(r), bits(sc["spheres"][1]))
    assert len(r) == 64


def test_null_and_range_arguments_are_bad_arg_not_crashes(L):
    """nothing aborts across the ABI: null handles / pointers come back as OXHIP_ERR_BAD_ARG"""
    n = C.c_uint32()
    assert L.oxhip_rrt_batch_create(None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_set_spheres(None, None, None, 0) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_set_boxes(None, None, None, 0) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_setup(None, None, None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_solve(None, 10, 0.0, 0, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_get_counts(None, None, None, None, None, None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_get_tree(None, 0, None, None, 0, C.byref(n)) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_get_path(None, 0, None, 0, C.byref(n)) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_get_goal_counts(None, None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_last_timing(None, None, None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_rrt_batch_destroy(None) == capi.OK
    assert L.oxhip_nn_argmin_batch(0, 3, None, None, 1, None, None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_distance_batch(0, 0, None, None, 1, None) == capi.ERR_BAD_ARG
    assert L.oxhip_f64_op_batch(0, 9, None, None, None, 1, None) == capi.ERR_BAD_ARG
    assert L.oxhip_device_count(None) == capi.ERR_BAD_ARG
    assert b"null" in L.oxhip_last_error_string() or L.oxhip_last_error_string()
    with pytest.raises(capi.OxhipError) as ei:   # unknown planner / kernel kinds
        capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.1, 0.0, 1, 10, planner=7)
    assert ei.value.status == capi.ERR_BAD_ARG
    with pytest.raises(capi.OxhipError) as ei:
        capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.1, 0.0, 1, 10, kernel=9)
    assert ei.value.status == capi.ERR_BAD_ARG
    with pytest.raises(capi.OxhipError) as ei:   # RRTConnect has no resident kernel
        capi.RRTBatch(2, [(0.0, 1.0)] * 2, 0.1, 0.0, 1, 10, kernel=capi.KERNEL_RESIDENT, planner=capi.PLANNER_RRT_CONNECT)
    assert ei.value.status == capi.ERR_BAD_ARG


@pytest.mark.parametrize("kw,code", [
    (dict(dim=0, bounds=[]), capi.ERR_BAD_ARG),
    (dict(bounds=[(0.0, math.inf), (0.0, 1.0)]), capi.ERR_UNBOUNDED),    # rvss.rs:239-241
    (dict(bounds=[(1.0, 1.0), (0.0, 1.0)]), capi.ERR_ZERO_VOLUME),       # rvss.rs:78-83,242-244
    (dict(max_milestones=0), capi.ERR_BAD_ARG),
    (dict(connection_radius=float("nan")), capi.ERR_BAD_ARG),
    (dict(lvs_fraction=0.0), capi.ERR_BAD_ARG),                          # check_motion would never end
    (dict(connection_radius=1e9), capi.ERR_BAD_ARG),                     # > 1e6 validity checks per edge
    (dict(), capi.ERR_NO_DEVICE),                                        # valid arguments: only the GPU is missing
])
def test_prm_create_validates_like_the_reference(L, kw, code):
    import torch
    if code == capi.ERR_NO_DEVICE and torch.cuda.is_available():
        pytest.skip("a GPU is present")
    args = dict(dim=2, bounds=[(0.0, 10.0)] * 2, connection_radius=0.5, max_milestones=100)
    args.update(kw)
    with pytest.raises(capi.OxhipError) as ei:
        capi.PRMRoadmap(**args)
    assert ei.value.status == code


def test_prm_null_arguments_are_rejected(L):
    assert L.oxhip_prm_create(None, None) == capi.ERR_BAD_ARG
    assert L.oxhip_prm_destroy(None) == capi.OK
    for fn in (L.oxhip_prm_construct_roadmap,):
        assert fn(None) == capi.ERR_BAD_ARG
    assert L.oxhip_prm_solve(None, 0.0, None, 0, None) == capi.ERR_BAD_ARG
    assert L.oxhip_prm_get_sizes(None, None, None, None) == capi.ERR_BAD_ARG
