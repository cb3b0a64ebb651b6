"""Mirror of `oxmpl_py.base` (reference: oxmpl-py/src/base/*.rs) for the GPU RRT path.

Same class names, constructor signatures and error types as the reference's PyO3 classes
(`RealVectorState`, `RealVectorStateSpace`, `ProblemDefinition.from_real_vector`, `Path`), so the
reference's Python test (oxmpl-py/tests/test_rrt_rvss.py) ports by changing the import line and
replacing its per-state Python callback by a descriptor object: a Python callable evaluated once
per interpolated state under the GIL (oxmpl-py/src/base/state_validity_checker.rs:30-48) is the
opposite of a batched device path, and there is deliberately no CPU fallback that would run it.

Everything that computes (`distance`, validity) is evaluated by the HIP library.
"""
import math

import numpy as np

from . import capi


class RealVectorState:
    """oxmpl_py.base.RealVectorState (oxmpl-py/src/base/real_vector_state.rs:17-40)"""

    def __init__(self, values):
        self._values = [float(v) for v in values]

    @property
    def values(self):
        return list(self._values)

    def __repr__(self):
        return "<RealVectorState values=%r>" % (self._values,)


class RealVectorStateSpace:
    """oxmpl_py.base.RealVectorStateSpace (oxmpl-py/src/base/real_vector_state_space.rs:15-57).
    Construction errors are ValueError with the reference's messages (error.rs:40-52)."""

    def __init__(self, dimension, bounds=None):
        dimension = int(dimension)
        if bounds is not None:
            bounds = [(float(lo), float(hi)) for lo, hi in bounds]
            if len(bounds) != dimension:
                raise ValueError("provided bounds length (%d) does not match specified dimension (%d)."
                                 % (len(bounds), dimension))
            for lo, hi in bounds:
                if lo >= hi:
                    raise ValueError("Lower bound %s is greater than upper bound %s." % (lo, hi))
        else:
            if dimension == 0:
                raise ValueError("Cannot create 0-dimensional unbounded space.")
            bounds = [(-math.inf, math.inf)] * dimension
        self.dimension = dimension
        self.bounds = bounds
        self.longest_valid_segment_fraction = 0.05

    def distance(self, state1, state2):
        a = np.array([state1.values], dtype=np.float64)
        b = np.array([state2.values], dtype=np.float64)
        return float(capi.distance_batch(a, b)[0])

    def get_maximum_extent(self):
        if any((not math.isfinite(lo)) or (not math.isfinite(hi)) for lo, hi in self.bounds):
            return 1.0
        # sqrt of the sequential sum of squared widths (real_vector_state_space.rs:103-118), on the device
        z = np.zeros((1, self.dimension))
        w = np.array([[hi - lo for lo, hi in self.bounds]], dtype=np.float64)
        return float(capi.distance_batch(w, z)[0])

    def set_longest_valid_segment_fraction(self, fraction):
        if 0.0 < fraction <= 1.0:
            self.longest_valid_segment_fraction = float(fraction)
        elif fraction <= 0.0:
            self.longest_valid_segment_fraction = 0.0
        else:
            self.longest_valid_segment_fraction = 1.0


class SphereBoxValidityChecker:
    """Device-describable StateValidityChecker: a state is valid iff it lies strictly outside every
    sphere (distance(centre, p) > radius, the README's predicate) and inside no box (faces
    inclusive, the wall of the reference's tests).  Pass it to RRT.setup() where the reference takes
    a Python callable."""

    def __init__(self, spheres=(), boxes=()):
        self.spheres = [([float(v) for v in c], float(r)) for c, r in spheres]
        self.boxes = [([float(v) for v in lo], [float(v) for v in hi]) for lo, hi in boxes]


class ProblemDefinition:
    """oxmpl_py.base.ProblemDefinition (oxmpl-py/src/base/problem_definition.rs:43-75)"""

    def __init__(self, space, start_state, goal):
        self.space, self.start_state, self.goal = space, start_state, goal

    @staticmethod
    def from_real_vector(space, start_state, goal):
        """`goal` is any object with a `target` (RealVectorState) and a `radius`: the reference's
        CircularGoal classes qualify as they are.  is_satisfied(s) = distance(s, target) <= radius;
        the device goal sampler returns `target` (README.md:160-162), `goal.sample_goal` is not called."""
        if not isinstance(space, RealVectorStateSpace):
            raise TypeError("space must be a RealVectorStateSpace")
        if not hasattr(goal, "target") or not hasattr(goal, "radius"):
            raise TypeError("the GPU path needs a ball goal: an object with `target` and `radius` attributes")
        if len(start_state.values) != space.dimension or len(goal.target.values) != space.dimension:
            raise ValueError("state dimension does not match the space")
        return ProblemDefinition(space, start_state, goal)


class Path:
    """oxmpl_py.base.Path (oxmpl-py/src/base/path.rs:27-99)"""

    def __init__(self, states):
        self._states = list(states)

    @staticmethod
    def from_real_vector_states(states):
        return Path(states)

    @property
    def states(self):
        return list(self._states)

    def __len__(self):
        return len(self._states)

    def __repr__(self):
        return "<Path with %d states>" % len(self._states)
