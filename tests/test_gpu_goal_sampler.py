"""-m gpu: OXHIP_GOAL_SAMPLE_UNIFORM_DISC -- the goal sampler of the reference's own test fixtures
(oxmpl/tests/rrt_rvss_tests.rs:55-66, oxmpl-py/tests/test_rrt_rvss.py:19-25): angle = random_range(0..2 PI), radius = r sqrt(random f64),
(x, y) = centre + radius (cos, sin).  The device's sin / cos (ox_sincos.hpp) equal the oracle's restatement bit for bit, and the planners
that sample through it equal the oracle on the reference's wall scene and on the README scene, for every kernel that supports the mode.
Against a rustc-built oxmpl (libm's sin / cos) the mode is within 1e-6 relative, not bit-exact: tests/test_oracle_golden.py measures it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oxmpl_amd import capi, scenarios  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from helpers import bits  # noqa: E402

KERNELS = {"stream": capi.KERNEL_STREAM, "lanes": capi.KERNEL_LANES, "cells": capi.KERNEL_CELLS, "auto": capi.KERNEL_AUTO}


def test_device_sincos_equals_the_oracle_restatement():
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(400000) * 2.0 * np.pi, np.arange(9) * (np.pi / 4), np.nextafter(np.arange(1, 9) * (np.pi / 4), 0.0),
                        np.nextafter(np.arange(1, 9) * (np.pi / 4), 10.0), [0.0, 5e-324, 1e-300, 2.0 ** -27, 2.0 ** -28, 1e-9, 6.283185307179586,
                                                                              np.nextafter(6.283185307179586, 0.0)]])
    want = np.array([orc.sincos(v) for v in x])
    assert np.array_equal(bits(capi.f64_op_batch(5, x)), bits(want[:, 0]))
    assert np.array_equal(bits(capi.f64_op_batch(6, x)), bits(want[:, 1]))


def _oracle(sc, seed, pid, max_nodes, stop, star_radius=None):
    if star_radius is None:
        o = orc.OracleRRT(2, sc["bounds"], sc["max_distance"], sc["goal_bias"], sc["lvs_fraction"], max_nodes, stop, seed, pid)
    else:
        o = orc.OracleRRTStar(2, sc["bounds"], sc["max_distance"], sc["goal_bias"], star_radius, sc["lvs_fraction"], max_nodes, stop, seed, pid)
    o.set_goal_sampler(1)
    if sc["spheres"] is not None:
        o.set_spheres(*sc["spheres"])
    if sc["boxes"] is not None:
        o.set_boxes(*sc["boxes"])
    o.setup(sc["start"], sc["goal_centre"], sc["goal_radius"])
    return o


def _same(g, p, o, c):
    assert int(c["nodes"][p]) == o.num_nodes and int(c["iterations"][p]) == o.iterations and int(c["checksum"][p]) == o.checksum
    assert int(c["goal_node"][p]) == o.goal_node and int(c["accepted"][p]) == o.accepted
    gs, gp = g.tree(p)
    os_, op = o.tree()
    assert np.array_equal(gp, op) and np.array_equal(bits(gs), bits(os_))
    assert np.array_equal(bits(g.path(p)), bits(o.path()))


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("scene,goal_bias", [("wall", 0.05), ("wall", 0.5), ("wall", 1.0), ("config1", 0.05), ("config1", 0.3)])
def test_rrt_with_the_disc_goal_sampler(scene, goal_bias, kernel):
    sc = dict(scenarios.wall() if scene == "wall" else scenarios.config1(), goal_bias=goal_bias)
    P = 6
    # to the goal (the reference's behaviour) ...
    g = scenarios.make_batch(sc, P, 10000, True, 11, 3, 0, KERNELS[kernel], goal_sampler=capi.GOAL_SAMPLE_UNIFORM_DISC)
    g.solve(700)
    g.solve(10 ** 6)
    pl = [_oracle(sc, 11, 3 + p, 10000, True) for p in range(P)]
    orc.solve_many(pl, 10 ** 6 + 700, threads=6)
    c = g.counts()
    for p in range(P):
        _same(g, p, pl[p], c)
        if goal_bias < 1.0:
            assert pl[p].goal_node >= 0
    g.close()
    # ... and past it, then frozen
    g = scenarios.make_batch(sc, P, 3000, False, 12, 40, 0, KERNELS[kernel], goal_sampler=capi.GOAL_SAMPLE_UNIFORM_DISC, frozen_split=3)
    g.solve(2500)
    g.solve(300, freeze=True)
    pl = [_oracle(sc, 12, 40 + p, 3000, False) for p in range(P)]
    orc.solve_many(pl, 2500, threads=6)
    orc.solve_many(pl, 300, freeze=True, threads=6)
    c = g.counts()
    for p in range(P):
        _same(g, p, pl[p], c)
    g.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_CELLS, capi.KERNEL_LANES, capi.KERNEL_STREAM], ids=["decoupled_cells", "decoupled_lanes", "one_kernel"])
def test_rrt_star_with_the_disc_goal_sampler(kernel):
    sc = dict(scenarios.wall(), goal_bias=0.2)
    g = scenarios.make_batch(sc, 4, 1500, False, 21, 5, 0, kernel, capi.PLANNER_RRT_STAR, 0.8, goal_sampler=capi.GOAL_SAMPLE_UNIFORM_DISC)
    g.solve(1200)
    c = g.counts()
    for p in range(4):
        o = _oracle(sc, 21, 5 + p, 1500, False, star_radius=0.8)
        o.solve(1200)
        _same(g, p, o, c)
        assert np.array_equal(bits(g.costs(p)), bits(o.costs()))
    g.close()


def test_the_mode_is_refused_where_it_is_not_built():
    sc = scenarios.wall()
    for kw in (dict(kernel=capi.KERNEL_RESIDENT), dict(planner=capi.PLANNER_RRT_CONNECT)):
        with pytest.raises(capi.OxhipError) as ei:
            capi.RRTBatch(2, sc["bounds"], 0.5, 0.05, 1, goal_sampler=capi.GOAL_SAMPLE_UNIFORM_DISC, **kw)
        assert ei.value.status == capi.ERR_BAD_ARG
    with pytest.raises(capi.OxhipError):
        capi.RRTBatch(3, [(0, 1)] * 3, 0.5, 0.05, 1, goal_sampler=capi.GOAL_SAMPLE_UNIFORM_DISC)
