"""Mirror of `oxmpl_py.geometric` (reference: oxmpl-py/src/geometric/rrt.rs:30-136) for the GPU path.

    planner = RRT(max_distance=0.5, goal_bias=0.05, problem_definition=problem_def)
    planner.setup(SphereBoxValidityChecker(...))
    path = planner.solve(timeout_secs=5.0)      # -> Path; raises Exception(str) like the reference

`RRTBatch` (oxmpl_amd.capi) is the batched form the benchmark uses; this class is the one-problem
drop-in for users of the reference's Python API.
"""
import numpy as np

from . import capi
from .base import Path, ProblemDefinition, RealVectorState, SphereBoxValidityChecker

_MESSAGES = {  # Display strings of PlanningError (oxmpl/src/base/error.rs:110-136)
    capi.ERR_TIMEOUT: "No solution found within timeout.",
    capi.ERR_NO_SOLUTION_FOUND: "No solution found.",
    capi.ERR_PLANNER_UNINITIALISED: "<Planner>.setup() was not called, thus Planner is uninitialised.",
    capi.ERR_INVALID_START_STATE: "Start state is not valid in the current StateSpace.",
    capi.ERR_UNSAMPLED_STATE_SPACE: "StateSpace is not sampled. Either Tree or Roadmap is empty.",
}


class RRT:
    """One-problem drop-in for `oxmpl_py.geometric.RRT`.

    Difference from the reference, on purpose: the tree lives in a device buffer sized at setup(), so it is capped at
    `max_nodes` (default 10,000).  The reference has no node cap (rrt.rs:170-226 grows until the wall clock runs out);
    here a problem that fills its tree before the timeout ends with "No solution found." (PlanningError::NoSolutionFound,
    declared but unused by the reference's RRT) instead of "No solution found within timeout.".  Pass a larger `max_nodes`
    for narrow-passage problems.  `solve(0)` is Timeout, as upstream."""
    _PLANNER = capi.PLANNER_RRT

    def __init__(self, max_distance, goal_bias, problem_definition, max_nodes=10000, seed=0, problem_id=0, device=0,
                 goal_sampler=capi.GOAL_SAMPLE_CENTRE):
        """goal_sampler: what `goal.sample_goal()` draws on the device -- capi.GOAL_SAMPLE_CENTRE (the goal's `target`, as the
        README's goal, no random draw) or capi.GOAL_SAMPLE_UNIFORM_DISC (uniform in the disc as the CircularGoal of
        oxmpl-py/tests/test_rrt_rvss.py:19-25 and oxmpl/tests/rrt_rvss_tests.rs:55-66 samples it; 2-D spaces)"""
        if not isinstance(problem_definition, ProblemDefinition):
            raise TypeError("problem_definition must be a ProblemDefinition")
        self.max_distance, self.goal_bias = float(max_distance), float(goal_bias)
        self._pd = problem_definition
        self._opts = dict(max_nodes=max_nodes, seed=seed, first_problem_id=problem_id, device=device, goal_sampler=goal_sampler)
        self._batch = None

    def setup(self, validity_checker):
        """Planner::setup (rrt.rs:140-156).  The reference takes a Python callable here; the GPU path
        takes a SphereBoxValidityChecker (see oxmpl_amd.base)."""
        if not isinstance(validity_checker, SphereBoxValidityChecker):
            raise TypeError("the GPU path cannot call a Python function per interpolated state; "
                            "describe the obstacles with oxmpl_amd.base.SphereBoxValidityChecker")
        pd = self._pd
        if self._batch is not None:
            self._batch.close()
        try:
            b = capi.RRTBatch(pd.space.dimension, pd.space.bounds, self.max_distance, self.goal_bias, 1,
                              lvs_fraction=pd.space.longest_valid_segment_fraction, stop_at_goal=True,
                              planner=self._PLANNER, search_radius=getattr(self, "search_radius", 0.0), **self._opts)
        except capi.OxhipError as e:
            if e.status in (capi.ERR_UNBOUNDED, capi.ERR_ZERO_VOLUME, capi.ERR_BAD_ARG):
                raise ValueError(str(e)) from None
            raise
        if validity_checker.spheres:
            b.set_spheres([c for c, _ in validity_checker.spheres], [r for _, r in validity_checker.spheres])
        if validity_checker.boxes:
            b.set_boxes([lo for lo, _ in validity_checker.boxes], [hi for _, hi in validity_checker.boxes])
        b.setup(pd.start_state.values, pd.goal.target.values, float(pd.goal.radius))
        self._batch = b
        self._checker = validity_checker

    def solve(self, timeout_secs):
        """Planner::solve (rrt.rs:158-227); errors surface as Exception(message) like the reference
        (oxmpl-py/src/geometric/rrt.rs:117)."""
        if self._batch is None:
            raise Exception(_MESSAGES[capi.ERR_PLANNER_UNINITIALISED])
        timeout_secs = float(timeout_secs)
        if timeout_secs != timeout_secs or timeout_secs < 0.0:
            # Duration::from_secs_f32 (oxmpl-py/src/geometric/rrt.rs:111) panics on NaN / negative values
            raise ValueError("timeout_secs must be a non-negative number")
        if timeout_secs == 0.0:
            # rrt.rs:172-174: `start_time.elapsed() > timeout` already holds at the first check
            raise Exception(_MESSAGES[capi.ERR_TIMEOUT])
        st = self._batch.solve(1 << 40, timeout_s=timeout_secs)
        if st[0] != capi.OK:
            raise Exception(_MESSAGES.get(int(st[0]), capi.status_string(int(st[0]))))
        return Path([RealVectorState(row) for row in self._batch.path(0)])

    def is_state_valid(self, state):
        """the checker's predicate, evaluated on the device (the reference calls the user's callable)"""
        return bool(self._batch.is_valid(np.array([state.values]))[0])

    @property
    def num_nodes(self):
        return int(self._batch.counts()["nodes"][0])


class RRTConnect(RRT):
    """oxmpl_py.geometric.RRTConnect (oxmpl-py/src/geometric/rrt_connect.rs; planner:
    oxmpl/src/geometric/planners/rrt_connect.rs): same constructor / setup / solve surface as RRT,
    two trees grown towards each other; `max_nodes` caps each tree."""
    _PLANNER = capi.PLANNER_RRT_CONNECT

    @property
    def num_nodes(self):
        return int(self._batch.counts()["nodes"][0]) + int(self._batch.goal_counts()["nodes"][0])


class RRTStar(RRT):
    """oxmpl_py.geometric.RRTStar (oxmpl-py/src/geometric/rrt_star.rs:30-140; planner:
    oxmpl/src/geometric/planners/rrt_star.rs): RRTStar(max_distance, goal_bias, search_radius, problem_definition)."""
    _PLANNER = capi.PLANNER_RRT_STAR

    def __init__(self, max_distance, goal_bias, search_radius, problem_definition, **opts):
        super().__init__(max_distance, goal_bias, problem_definition, **opts)
        self.search_radius = float(search_radius)

    def path_cost(self):
        """Node::cost of the goal node (rrt_star.rs:26)"""
        g = int(self._batch.counts()["goal_node"][0])
        return float(self._batch.costs(0)[g]) if g >= 0 else float("inf")


class PRM:
    """oxmpl_py.geometric.PRM (oxmpl-py/src/geometric/prm.rs:30-206; planner: oxmpl/src/geometric/planners/prm.rs).

        planner = PRM(timeout=5.0, connection_radius=0.5, problem_definition=problem_def)
        planner.setup(SphereBoxValidityChecker(...))
        planner.construct_roadmap()
        path = planner.solve(timeout_secs=5.0)

    The reference samples for `timeout` seconds of wall clock; the device path also stops at `max_milestones`
    (a GPU fills five seconds with tens of millions of milestones).  `seed` / `stream` key the sampler."""

    def __init__(self, timeout, connection_radius, problem_definition, max_milestones=16384, max_samples=0, seed=0,
                 stream=0, device=0):
        if not isinstance(problem_definition, ProblemDefinition):
            raise TypeError("problem_definition must be a ProblemDefinition")
        self.timeout, self.connection_radius = float(timeout), float(connection_radius)
        self._pd = problem_definition
        self._opts = dict(max_milestones=max_milestones, max_samples=max_samples, seed=seed, stream=stream, device=device)
        self._prm = None

    def setup(self, validity_checker):
        """Planner::setup (prm.rs:217-225): stores problem and checker, clears the roadmap"""
        if not isinstance(validity_checker, SphereBoxValidityChecker):
            raise TypeError("the GPU path cannot call a Python function per interpolated state; "
                            "describe the obstacles with oxmpl_amd.base.SphereBoxValidityChecker")
        pd = self._pd
        if self._prm is not None:
            self._prm.close()
        try:
            g = capi.PRMRoadmap(pd.space.dimension, pd.space.bounds, self.connection_radius, timeout=self.timeout,
                                lvs_fraction=pd.space.longest_valid_segment_fraction, **self._opts)
        except capi.OxhipError as e:
            if e.status in (capi.ERR_UNBOUNDED, capi.ERR_ZERO_VOLUME, capi.ERR_BAD_ARG):
                raise ValueError(str(e)) from None
            raise
        if validity_checker.spheres:
            g.set_spheres([c for c, _ in validity_checker.spheres], [r for _, r in validity_checker.spheres])
        if validity_checker.boxes:
            g.set_boxes([lo for lo, _ in validity_checker.boxes], [hi for _, hi in validity_checker.boxes])
        g.setup(pd.start_state.values, pd.goal.target.values, float(pd.goal.radius))
        self._prm = g

    def set_problem_definition(self, problem_definition):
        """PRM::set_problem_definition (prm.rs:88-90): new start / goal on the roadmap already built"""
        if not isinstance(problem_definition, ProblemDefinition):
            raise TypeError("problem_definition must be a ProblemDefinition")
        self._pd = problem_definition
        if self._prm is not None:
            self._prm.set_problem(problem_definition.start_state.values, problem_definition.goal.target.values,
                                  float(problem_definition.goal.radius))

    def construct_roadmap(self):
        if self._prm is None:
            raise Exception(_MESSAGES[capi.ERR_PLANNER_UNINITIALISED])
        self._prm.construct_roadmap()

    def solve(self, timeout_secs):
        """Planner::solve (prm.rs:227-307); errors surface as Exception(message) like the reference
        (oxmpl-py/src/geometric/prm.rs:120-145)"""
        if self._prm is None:
            raise Exception(_MESSAGES[capi.ERR_PLANNER_UNINITIALISED])
        st, path = self._prm.solve(float(timeout_secs))
        if st != capi.OK:
            raise Exception(_MESSAGES.get(int(st), capi.status_string(int(st))))
        return Path([RealVectorState(row) for row in path])

    def get_roadmap(self):
        """PRM::get_roadmap (prm.rs:82-84) as (states, offsets, neighbours): node i's edges are
        neighbours[offsets[i]:offsets[i+1]]"""
        return self._prm.roadmap()

    @property
    def num_milestones(self):
        return 0 if self._prm is None else self._prm.sizes()[0]
