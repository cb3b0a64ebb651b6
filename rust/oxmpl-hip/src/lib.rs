//! `oxmpl-hip`: oxmpl's planners on an AMD MI355X, behind oxmpl's own `Planner` trait.
//!
//! ```ignore
//! let checker = Arc::new(SphereField { spheres: vec![(vec![0.0, 0.0], 2.0)] });      // impl DeviceValidityChecker
//! let mut planner = HipRRT::new(0.5, 0.05, checker.clone());                         // RRT::new(0.5, 0.05)
//! planner.setup(problem_def, checker);                                               // Planner::setup
//! let path = planner.solve(Duration::from_secs(5))?;                                 // Planner::solve
//! ```
//!
//! # How the validity checker crosses the boundary
//!
//! `Planner::setup` (oxmpl/src/base/planner.rs:42-46) hands the planner an `Arc<dyn StateValidityChecker<S>>`.
//! That trait (oxmpl/src/base/validity.rs:39-48) has one method, `is_valid`, and no `Any` supertrait, so nothing
//! can be recovered from the trait object but verdicts on single states -- and a GPU cannot call back into Rust
//! per interpolated state.  The obstacle *description* therefore enters through the constructor, typed:
//! `HipRRT::new(max_distance, goal_bias, checker: Arc<D>)` with `D: DeviceValidityChecker`.  `setup` keeps its
//! trait signature; the `dyn` checker it receives is used for what it can do -- it is probed on the start
//! state and on a deterministic set of states inside the bounds, and `setup` panics if its verdicts differ from
//! the description's, so the two cannot silently drift apart.  There is no CPU fallback in this crate.
//!
//! The goal is not erased by the trait (`G` is a type parameter of `Planner`), so `G: DeviceGoal` is enough.
//!
//! Source only: the repository's build image has no Rust toolchain.

pub mod ffi;

use std::ffi::CStr;
use std::ptr;
use std::sync::Arc;
use std::time::Duration;

use oxmpl::base::error::PlanningError;
use oxmpl::base::goal::GoalSampleableRegion;
use oxmpl::base::planner::{Path, Planner};
use oxmpl::base::problem_definition::ProblemDefinition;
use oxmpl::base::space::{RealVectorStateSpace, StateSpace};
use oxmpl::base::state::RealVectorState;
use oxmpl::base::validity::StateValidityChecker;

type Pd<G> = ProblemDefinition<RealVectorState, RealVectorStateSpace, G>;

/// A `StateValidityChecker` that can describe itself to the device: a state is valid iff it lies strictly
/// outside every sphere (`distance(centre, p) > radius`, the predicate of README.md:147-150) and inside no
/// axis-aligned box (faces inclusive, the wall of oxmpl/tests/rrt_rvss_tests.rs:24-36).  `is_valid` of the
/// implementor must be exactly that predicate; `HipRRT::setup` spot-checks it.
pub trait DeviceValidityChecker: StateValidityChecker<RealVectorState> {
    /// `(centre, radius)` pairs
    fn spheres(&self) -> Vec<(Vec<f64>, f64)> {
        Vec::new()
    }
    /// `(lower corner, upper corner)` pairs
    fn boxes(&self) -> Vec<(Vec<f64>, Vec<f64>)> {
        Vec::new()
    }
}

/// A goal the device understands: `is_satisfied(s) = distance(s, centre) <= radius`.  The device samples the
/// region at its centre (no RNG draw), which is what README.md:160-162 does; goals that sample elsewhere plan
/// correctly but draw a different random stream than the CPU planner would.
pub trait DeviceGoal: GoalSampleableRegion<RealVectorState> {
    fn ball(&self) -> (Vec<f64>, f64);
    /// what `sample_goal` draws on the device (default: the centre, no random word -- README.md:160-162)
    fn sampling(&self) -> GoalSampling {
        GoalSampling::Centre
    }
}

/// What `sample_goal` draws on the device (`oxhip_goal_sampler`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum GoalSampling {
    /// the ball's centre, no random word consumed (`README.md:160-162`)
    Centre,
    /// uniform in the disc, as the `CircularGoalRegion` of `oxmpl/tests/rrt_rvss_tests.rs:55-66` samples it:
    /// `random_range(0.0..2.0 * PI)`, then `random::<f64>().sqrt()`; 2-D spaces.  The device evaluates cos / sin with its own
    /// portable routine (within one ulp of libm): results are within 1e-6 relative of the CPU path, not bit-identical.
    UniformDisc,
}

/// Deterministic knobs the reference does not have (its only stop is the wall clock, rrt.rs:172-174, and its RNG
/// is OS-seeded, rrt.rs:167).
#[derive(Clone, Copy, Debug)]
pub struct HipOptions {
    /// tree capacity; a full tree ends `solve` with `NoSolutionFound`
    pub max_nodes: u32,
    /// ChaCha12 key (LE(seed) || 0^24) and stream id of this planner's random stream
    pub seed: u64,
    pub stream: u64,
    /// HIP device ordinal
    pub device: i32,
}

impl Default for HipOptions {
    fn default() -> Self {
        HipOptions { max_nodes: 10_000, seed: 0, stream: 0, device: 0 }
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::oxhip_last_error_string()).to_string_lossy().into_owned() }
}

fn to_planning_error(status: i32) -> PlanningError {
    match status {
        ffi::OXHIP_ERR_TIMEOUT => PlanningError::Timeout,
        ffi::OXHIP_ERR_PLANNER_UNINITIALISED => PlanningError::PlannerUninitialised,
        ffi::OXHIP_ERR_INVALID_START_STATE => PlanningError::InvalidStartState,
        ffi::OXHIP_ERR_UNSAMPLED_STATE_SPACE => PlanningError::UnsampledStateSpace,
        _ => PlanningError::NoSolutionFound,
    }
}

/// `longest_valid_segment_fraction` is a private field of `RealVectorStateSpace` (rvss.rs:25); its public trace is
/// `get_longest_valid_segment_length() = extent * fraction` (rvss.rs:251-253).  The C ABI takes the fraction and
/// repeats that product, so look for the double whose product with the extent reproduces the length bit for bit
/// (the quotient, or one of its neighbours).
fn fraction_of(space: &RealVectorStateSpace) -> f64 {
    let extent = space.get_maximum_extent();
    let length = space.get_longest_valid_segment_length();
    if !(extent > 0.0) {
        return 0.05;
    }
    let q = length / extent;
    for step in [0i64, 1, -1, 2, -2] {
        let cand = f64::from_bits((q.to_bits() as i64 + step) as u64);
        if extent * cand == length {
            return cand;
        }
    }
    q
}

fn flatten(rows: &[Vec<f64>], dim: usize, what: &str) -> Vec<f64> {
    let mut out = Vec::with_capacity(rows.len() * dim);
    for r in rows {
        assert_eq!(r.len(), dim, "{} has the wrong dimension", what);
        out.extend_from_slice(r);
    }
    out
}

/// the description's own verdict, on the host: used only to cross-check the `dyn` checker in `setup`
fn described_valid(spheres: &[(Vec<f64>, f64)], boxes: &[(Vec<f64>, Vec<f64>)], p: &[f64]) -> bool {
    for (c, r) in spheres {
        let mut acc = 0.0;
        for (a, b) in c.iter().zip(p) {
            acc += (a - b) * (a - b);
        }
        if !(acc.sqrt() > *r) {
            return false;
        }
    }
    for (lo, hi) in boxes {
        if lo.iter().zip(hi).zip(p).all(|((l, h), x)| l <= x && x <= h) {
            return false;
        }
    }
    true
}

/// which of the reference's tree planners a `HipRRT` stands for
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum TreePlanner {
    /// `geometric::RRT` (rrt.rs)
    Rrt,
    /// `geometric::RRTConnect` (rrt_connect.rs)
    RrtConnect,
    /// `geometric::RRTStar` (rrt_star.rs) with its `search_radius`
    RrtStar { search_radius: f64 },
}

/// GPU twin of `oxmpl::geometric::RRT` (and, through `TreePlanner`, of `RRTConnect` / `RRTStar`): same public
/// parameters (rrt.rs:53-57), same trait, one planning problem per instance.  Batches of thousands of problems
/// skip the trait and drive `ffi::oxhip_rrt_batch_*` with `n_problems = P` (what bench.py does through ctypes).
pub struct HipRRT<D: DeviceValidityChecker, G: DeviceGoal> {
    pub max_distance: f64,
    pub goal_bias: f64,
    pub planner: TreePlanner,
    pub options: HipOptions,
    checker: Arc<D>,
    problem_def: Option<Arc<Pd<G>>>,
    batch: *mut ffi::OxhipRrtBatch,
}

impl<D: DeviceValidityChecker, G: DeviceGoal> HipRRT<D, G> {
    /// `RRT::new(max_distance, goal_bias)` (rrt.rs:75-83) plus the typed obstacle description (see the crate docs)
    pub fn new(max_distance: f64, goal_bias: f64, checker: Arc<D>) -> Self {
        HipRRT {
            max_distance,
            goal_bias,
            planner: TreePlanner::Rrt,
            options: HipOptions::default(),
            checker,
            problem_def: None,
            batch: ptr::null_mut(),
        }
    }

    /// `RRTStar::new(max_distance, goal_bias, search_radius)` (rrt_star.rs:65-77)
    pub fn new_star(max_distance: f64, goal_bias: f64, search_radius: f64, checker: Arc<D>) -> Self {
        let mut p = Self::new(max_distance, goal_bias, checker);
        p.planner = TreePlanner::RrtStar { search_radius };
        p
    }

    /// `RRTConnect::new(max_distance, goal_bias)` (rrt_connect.rs:86-95)
    pub fn new_connect(max_distance: f64, goal_bias: f64, checker: Arc<D>) -> Self {
        let mut p = Self::new(max_distance, goal_bias, checker);
        p.planner = TreePlanner::RrtConnect;
        p
    }

    pub fn with_options(mut self, options: HipOptions) -> Self {
        self.options = options;
        self
    }

    fn release(&mut self) {
        if !self.batch.is_null() {
            unsafe { ffi::oxhip_rrt_batch_destroy(self.batch) };
            self.batch = ptr::null_mut();
        }
    }

    /// the tree as (states, parent indices): `RRT::tree` (rrt.rs:61)
    pub fn tree(&self) -> Option<(Vec<RealVectorState>, Vec<i32>)> {
        let pd = self.problem_def.as_ref()?;
        if self.batch.is_null() {
            return None;
        }
        let dim = pd.space.dimension;
        let mut n = 0u32;
        unsafe { ffi::oxhip_rrt_batch_get_tree(self.batch, 0, ptr::null_mut(), ptr::null_mut(), 0, &mut n) };
        let mut flat = vec![0.0f64; n as usize * dim];
        let mut parents = vec![0i32; n as usize];
        let st = unsafe { ffi::oxhip_rrt_batch_get_tree(self.batch, 0, flat.as_mut_ptr(), parents.as_mut_ptr(), n, &mut n) };
        if st != ffi::OXHIP_OK {
            return None;
        }
        let states = flat.chunks(dim).map(|c| RealVectorState::new(c.to_vec())).collect();
        Some((states, parents))
    }
}

impl<D: DeviceValidityChecker, G: DeviceGoal> Drop for HipRRT<D, G> {
    fn drop(&mut self) {
        self.release();
    }
}

impl<D: DeviceValidityChecker, G: DeviceGoal> Planner<RealVectorState, RealVectorStateSpace, G> for HipRRT<D, G> {
    /// `Planner::setup` (rrt.rs:140-156): stores the problem, clears the tree, pushes `start_states[0]`.
    ///
    /// Panics where the reference would (its `unwrap()` of the sampler's error at rrt.rs:183 for unbounded or
    /// zero-volume spaces) and when `validity_checker` disagrees with the description given to `new`.
    fn setup(&mut self, problem_def: Arc<Pd<G>>, validity_checker: Arc<dyn StateValidityChecker<RealVectorState>>) {
        self.release();
        let space = &problem_def.space;
        let dim = space.dimension;
        assert!(dim >= 1 && dim <= ffi::OXHIP_MAX_DIM, "HipRRT supports 1..=8 dimensions");
        let start = &problem_def.start_states[0];
        assert_eq!(start.values.len(), dim);

        let spheres = self.checker.spheres();
        let boxes = self.checker.boxes();

        // the trait object is all `setup` gets: make sure it is the checker that was described
        let mut probes: Vec<Vec<f64>> = vec![start.values.clone()];
        let mut lcg = 0x9E37_79B9_7F4A_7C15u64;
        for _ in 0..64 {
            let mut p = Vec::with_capacity(dim);
            for (lo, hi) in &space.bounds {
                lcg = lcg.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
                let u = (lcg >> 11) as f64 / (1u64 << 53) as f64;
                p.push(if lo.is_finite() && hi.is_finite() { lo + (hi - lo) * u } else { u });
            }
            probes.push(p);
        }
        for p in &probes {
            let state = RealVectorState::new(p.clone());
            assert_eq!(
                validity_checker.is_valid(&state),
                described_valid(&spheres, &boxes, p),
                "the StateValidityChecker given to setup() is not the one described to HipRRT::new (state {:?})",
                p
            );
        }

        let mut cfg: ffi::OxhipRrtConfig = unsafe { std::mem::zeroed() };
        cfg.struct_size = std::mem::size_of::<ffi::OxhipRrtConfig>() as u32;
        cfg.dim = dim as u32;
        for (k, (lo, hi)) in space.bounds.iter().enumerate() {
            cfg.bounds[2 * k] = *lo;
            cfg.bounds[2 * k + 1] = *hi;
        }
        cfg.max_distance = self.max_distance;
        cfg.goal_bias = self.goal_bias;
        cfg.lvs_fraction = fraction_of(space);
        cfg.n_problems = 1;
        cfg.max_nodes = self.options.max_nodes;
        cfg.stop_at_goal = 1;
        cfg.kernel = ffi::OXHIP_KERNEL_AUTO;
        cfg.seed = self.options.seed;
        cfg.first_problem_id = self.options.stream;
        cfg.device = self.options.device;
        cfg.space = ffi::OXHIP_SPACE_REAL_VECTOR;
        cfg.goal_sampler = match problem_def.goal.sampling() {
            GoalSampling::Centre => ffi::OXHIP_GOAL_SAMPLE_CENTRE,
            GoalSampling::UniformDisc => ffi::OXHIP_GOAL_SAMPLE_UNIFORM_DISC,
        };
        match self.planner {
            TreePlanner::Rrt => cfg.planner = ffi::OXHIP_PLANNER_RRT,
            TreePlanner::RrtConnect => cfg.planner = ffi::OXHIP_PLANNER_RRT_CONNECT,
            TreePlanner::RrtStar { search_radius } => {
                cfg.planner = ffi::OXHIP_PLANNER_RRT_STAR;
                cfg.search_radius = search_radius;
            }
        }
        let mut batch: *mut ffi::OxhipRrtBatch = ptr::null_mut();
        let st = unsafe { ffi::oxhip_rrt_batch_create(&cfg, &mut batch) };
        if st != ffi::OXHIP_OK {
            // UNBOUNDED / ZERO_VOLUME: the reference panics on the same condition at its first sample (rrt.rs:183)
            panic!("oxhip_rrt_batch_create failed with status {}: {}", st, last_error());
        }
        self.batch = batch;

        if !spheres.is_empty() {
            let centres = flatten(&spheres.iter().map(|(c, _)| c.clone()).collect::<Vec<_>>(), dim, "a sphere centre");
            let radii: Vec<f64> = spheres.iter().map(|(_, r)| *r).collect();
            let st = unsafe { ffi::oxhip_rrt_batch_set_spheres(self.batch, centres.as_ptr(), radii.as_ptr(), radii.len() as u32) };
            assert_eq!(st, ffi::OXHIP_OK, "set_spheres: {}", last_error());
        }
        if !boxes.is_empty() {
            let lo = flatten(&boxes.iter().map(|(l, _)| l.clone()).collect::<Vec<_>>(), dim, "a box corner");
            let hi = flatten(&boxes.iter().map(|(_, h)| h.clone()).collect::<Vec<_>>(), dim, "a box corner");
            let st = unsafe { ffi::oxhip_rrt_batch_set_boxes(self.batch, lo.as_ptr(), hi.as_ptr(), boxes.len() as u32) };
            assert_eq!(st, ffi::OXHIP_OK, "set_boxes: {}", last_error());
        }
        let (centre, radius) = problem_def.goal.ball();
        assert_eq!(centre.len(), dim);
        let st = unsafe { ffi::oxhip_rrt_batch_setup(self.batch, start.values.as_ptr(), centre.as_ptr(), &radius) };
        assert_eq!(st, ffi::OXHIP_OK, "setup: {}", last_error());
        self.problem_def = Some(problem_def);
    }

    /// `Planner::solve` (rrt.rs:158-227)
    fn solve(&mut self, timeout: Duration) -> Result<Path<RealVectorState>, PlanningError> {
        let pd = match (&self.problem_def, self.batch.is_null()) {
            (Some(pd), false) => pd.clone(),
            _ => return Err(PlanningError::PlannerUninitialised), // rrt.rs:160-163
        };
        if timeout.is_zero() {
            return Err(PlanningError::Timeout); // rrt.rs:172-174: elapsed() > 0 at the first check
        }
        let mut status = ffi::OXHIP_ERR_NO_SOLUTION_FOUND;
        let rc = unsafe { ffi::oxhip_rrt_batch_solve(self.batch, 1u64 << 40, timeout.as_secs_f64(), 0, &mut status) };
        if rc != ffi::OXHIP_OK {
            return Err(to_planning_error(rc));
        }
        if status != ffi::OXHIP_OK {
            return Err(to_planning_error(status));
        }
        let dim = pd.space.dimension;
        let mut len = 0u32;
        unsafe { ffi::oxhip_rrt_batch_get_path(self.batch, 0, ptr::null_mut(), 0, &mut len) };
        let mut flat = vec![0.0f64; len as usize * dim];
        let rc = unsafe { ffi::oxhip_rrt_batch_get_path(self.batch, 0, flat.as_mut_ptr(), len, &mut len) };
        if rc != ffi::OXHIP_OK {
            return Err(PlanningError::NoSolutionFound);
        }
        Ok(Path(flat.chunks(dim).map(|c| RealVectorState::new(c.to_vec())).collect()))
    }
}

/// GPU twin of `oxmpl::geometric::PRM` (prm.rs:48-57): same public fields, `construct_roadmap`,
/// `set_problem_definition`, and the `Planner` trait for the query.
pub struct HipPRM<D: DeviceValidityChecker, G: DeviceGoal> {
    pub timeout: f64,
    pub connection_radius: f64,
    /// construction also stops at this many milestones (a GPU fills seconds of wall clock with far too many)
    pub max_milestones: u32,
    pub options: HipOptions,
    checker: Arc<D>,
    problem_def: Option<Arc<Pd<G>>>,
    prm: *mut ffi::OxhipPrm,
}

impl<D: DeviceValidityChecker, G: DeviceGoal> HipPRM<D, G> {
    /// `PRM::new(timeout, connection_radius)` (prm.rs:70-78) plus the typed obstacle description
    pub fn new(timeout: f64, connection_radius: f64, checker: Arc<D>) -> Self {
        HipPRM {
            timeout,
            connection_radius,
            max_milestones: 16_384,
            options: HipOptions::default(),
            checker,
            problem_def: None,
            prm: ptr::null_mut(),
        }
    }

    fn release(&mut self) {
        if !self.prm.is_null() {
            unsafe { ffi::oxhip_prm_destroy(self.prm) };
            self.prm = ptr::null_mut();
        }
    }

    /// `PRM::construct_roadmap` (prm.rs:96-154)
    pub fn construct_roadmap(&mut self) -> Result<(), PlanningError> {
        if self.prm.is_null() {
            return Err(PlanningError::PlannerUninitialised);
        }
        match unsafe { ffi::oxhip_prm_construct_roadmap(self.prm) } {
            ffi::OXHIP_OK => Ok(()),
            st => Err(to_planning_error(st)),
        }
    }

    /// `PRM::set_problem_definition` (prm.rs:88-90): a new start / goal on the roadmap already built
    pub fn set_problem_definition(&mut self, pd: Arc<Pd<G>>) {
        if !self.prm.is_null() {
            let (centre, radius) = pd.goal.ball();
            let st = unsafe { ffi::oxhip_prm_set_problem(self.prm, pd.start_states[0].values.as_ptr(), centre.as_ptr(), radius) };
            assert_eq!(st, ffi::OXHIP_OK, "set_problem: {}", last_error());
        }
        self.problem_def = Some(pd);
    }

    /// `PRM::get_roadmap` (prm.rs:82-84) as (states, per-node edge lists in the reference's order)
    pub fn get_roadmap(&self) -> Vec<(RealVectorState, Vec<usize>)> {
        let (pd, prm) = match (&self.problem_def, self.prm.is_null()) {
            (Some(pd), false) => (pd, self.prm),
            _ => return Vec::new(),
        };
        let dim = pd.space.dimension;
        let (mut n, mut entries, mut samples) = (0u32, 0u64, 0u64);
        unsafe { ffi::oxhip_prm_get_sizes(prm, &mut n, &mut entries, &mut samples) };
        let mut states = vec![0.0f64; n as usize * dim];
        let mut offsets = vec![0u64; n as usize + 1];
        let mut nbrs = vec![0u32; entries as usize];
        let st = unsafe { ffi::oxhip_prm_get_roadmap(prm, states.as_mut_ptr(), n, offsets.as_mut_ptr(), nbrs.as_mut_ptr(), entries) };
        if st != ffi::OXHIP_OK {
            return Vec::new();
        }
        (0..n as usize)
            .map(|i| {
                let edges = nbrs[offsets[i] as usize..offsets[i + 1] as usize].iter().map(|&v| v as usize).collect();
                (RealVectorState::new(states[i * dim..(i + 1) * dim].to_vec()), edges)
            })
            .collect()
    }
}

impl<D: DeviceValidityChecker, G: DeviceGoal> Drop for HipPRM<D, G> {
    fn drop(&mut self) {
        self.release();
    }
}

impl<D: DeviceValidityChecker, G: DeviceGoal> Planner<RealVectorState, RealVectorStateSpace, G> for HipPRM<D, G> {
    /// `Planner::setup` (prm.rs:217-225): stores problem and checker, clears the roadmap
    fn setup(&mut self, problem_def: Arc<Pd<G>>, validity_checker: Arc<dyn StateValidityChecker<RealVectorState>>) {
        self.release();
        let space = &problem_def.space;
        let dim = space.dimension;
        assert!(dim >= 1 && dim <= ffi::OXHIP_MAX_DIM, "HipPRM supports 1..=8 dimensions");
        let start = &problem_def.start_states[0];
        let spheres = self.checker.spheres();
        let boxes = self.checker.boxes();
        assert_eq!(
            validity_checker.is_valid(start),
            described_valid(&spheres, &boxes, &start.values),
            "the StateValidityChecker given to setup() is not the one described to HipPRM::new"
        );
        let mut cfg: ffi::OxhipPrmConfig = unsafe { std::mem::zeroed() };
        cfg.struct_size = std::mem::size_of::<ffi::OxhipPrmConfig>() as u32;
        cfg.dim = dim as u32;
        for (k, (lo, hi)) in space.bounds.iter().enumerate() {
            cfg.bounds[2 * k] = *lo;
            cfg.bounds[2 * k + 1] = *hi;
        }
        cfg.timeout = self.timeout;
        cfg.connection_radius = self.connection_radius;
        cfg.lvs_fraction = fraction_of(space);
        cfg.max_milestones = self.max_milestones;
        cfg.device = self.options.device;
        cfg.max_samples = 0;
        cfg.seed = self.options.seed;
        cfg.stream = self.options.stream;
        let mut prm: *mut ffi::OxhipPrm = ptr::null_mut();
        let st = unsafe { ffi::oxhip_prm_create(&cfg, &mut prm) };
        if st != ffi::OXHIP_OK {
            panic!("oxhip_prm_create failed with status {}: {}", st, last_error());
        }
        self.prm = prm;
        if !spheres.is_empty() {
            let centres = flatten(&spheres.iter().map(|(c, _)| c.clone()).collect::<Vec<_>>(), dim, "a sphere centre");
            let radii: Vec<f64> = spheres.iter().map(|(_, r)| *r).collect();
            let st = unsafe { ffi::oxhip_prm_set_spheres(self.prm, centres.as_ptr(), radii.as_ptr(), radii.len() as u32) };
            assert_eq!(st, ffi::OXHIP_OK, "set_spheres: {}", last_error());
        }
        if !boxes.is_empty() {
            let lo = flatten(&boxes.iter().map(|(l, _)| l.clone()).collect::<Vec<_>>(), dim, "a box corner");
            let hi = flatten(&boxes.iter().map(|(_, h)| h.clone()).collect::<Vec<_>>(), dim, "a box corner");
            let st = unsafe { ffi::oxhip_prm_set_boxes(self.prm, lo.as_ptr(), hi.as_ptr(), boxes.len() as u32) };
            assert_eq!(st, ffi::OXHIP_OK, "set_boxes: {}", last_error());
        }
        let (centre, radius) = problem_def.goal.ball();
        let st = unsafe { ffi::oxhip_prm_setup(self.prm, start.values.as_ptr(), centre.as_ptr(), radius) };
        assert_eq!(st, ffi::OXHIP_OK, "setup: {}", last_error());
        self.problem_def = Some(problem_def);
    }

    /// `Planner::solve` (prm.rs:227-307): the roadmap query
    fn solve(&mut self, timeout: Duration) -> Result<Path<RealVectorState>, PlanningError> {
        let pd = match (&self.problem_def, self.prm.is_null()) {
            (Some(pd), false) => pd.clone(),
            _ => return Err(PlanningError::PlannerUninitialised), // prm.rs:229-236
        };
        let dim = pd.space.dimension;
        let mut len = 0u32;
        let st = unsafe { ffi::oxhip_prm_solve(self.prm, timeout.as_secs_f64(), ptr::null_mut(), 0, &mut len) };
        if st != ffi::OXHIP_OK && st != ffi::OXHIP_ERR_CAPACITY {
            return Err(to_planning_error(st));
        }
        let mut flat = vec![0.0f64; len as usize * dim];
        let st = unsafe { ffi::oxhip_prm_solve(self.prm, timeout.as_secs_f64(), flat.as_mut_ptr(), len, &mut len) };
        if st != ffi::OXHIP_OK {
            return Err(to_planning_error(st));
        }
        Ok(Path(flat.chunks(dim).map(|c| RealVectorState::new(c.to_vec())).collect()))
    }
}
