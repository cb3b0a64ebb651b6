// oxmpl.hpp -- C++17 host side of the drop-in boundary: the reference's trait surface for the RRT
// path, over the C ABI of include/oxmpl_hip.h.  Header-only; link with liboxmpl_hip.so.
//
// Names, argument meaning and error behaviour follow the reference (rossng/oxmpl, Rust):
//   oxmpl::base::RealVectorState            oxmpl/src/base/states/real_vector_state.rs:5-13
//   oxmpl::base::RealVectorStateSpace       oxmpl/src/base/spaces/real_vector_state_space.rs:65-129,137-186
//   oxmpl::base::StateValidityChecker       oxmpl/src/base/validity.rs:39-48
//   oxmpl::base::Goal / GoalRegion / GoalSampleableRegion   oxmpl/src/base/goal.rs:12-41
//   oxmpl::base::ProblemDefinition, Path    oxmpl/src/base/problem_definition.rs:16-20, planner.rs:16-17
//   oxmpl::base::PlanningError, StateSpaceError             oxmpl/src/base/error.rs:28-52,97-108
//   oxmpl::geometric::RRT                   oxmpl/src/geometric/planners/rrt.rs:53-83,140-227
//
// Rust's Result<T, E> becomes oxmpl::Result<T, E> (value or error, never an exception), Arc<T>
// becomes std::shared_ptr<T>.  Everything that computes runs on the GPU through the C ABI: there is
// no CPU fallback here.  A validity checker / goal must be device-describable (spheres + boxes, a
// ball); the planner refuses anything else with PlanningError::PlannerUninitialised at solve().
#pragma once

#include <chrono>
#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../oxmpl_hip.h"

namespace oxmpl {

template <typename T, typename E>
class Result {
  public:
    Result(T v) : v_(std::move(v)) {}                      // NOLINT: Ok(v)
    Result(E e) : v_(std::move(e)) {}                      // NOLINT: Err(e)
    bool is_ok() const { return v_.index() == 0; }
    bool is_err() const { return !is_ok(); }
    T& unwrap() { return std::get<0>(v_); }
    const T& unwrap() const { return std::get<0>(v_); }
    const E& err() const { return std::get<1>(v_); }

  private:
    std::variant<T, E> v_;
};

namespace base {

// ---- error.rs
enum class PlanningError { Timeout, NoSolutionFound, PlannerUninitialised, InvalidStartState, UnsampledStateSpace };
inline const char* to_string(PlanningError e) {
    switch (e) {
        case PlanningError::Timeout: return "No solution found within timeout.";
        case PlanningError::NoSolutionFound: return "No solution found.";
        case PlanningError::PlannerUninitialised: return "<Planner>.setup() was not called, thus Planner is uninitialised.";
        case PlanningError::InvalidStartState: return "Start state is not valid in the current StateSpace.";
        default: return "StateSpace is not sampled. Either Tree or Roadmap is empty.";
    }
}
struct StateSpaceError {
    enum Kind { DimensionMismatch, InvalidBound, ZeroDimensionUnbounded } kind;
    std::size_t expected = 0, found = 0;
    double lower = 0.0, upper = 0.0;
    bool operator==(const StateSpaceError& o) const { return kind == o.kind; }
};

// ---- states/real_vector_state.rs
struct RealVectorState {
    std::vector<double> values;
    RealVectorState() = default;
    explicit RealVectorState(std::vector<double> v) : values(std::move(v)) {}
    bool operator==(const RealVectorState& o) const { return values == o.values; }
};

// ---- spaces/real_vector_state_space.rs
class RealVectorStateSpace {
  public:
    std::size_t dimension = 0;
    std::vector<std::pair<double, double>> bounds;
    double longest_valid_segment_fraction = 0.05;

    // RealVectorStateSpace::new (rvss.rs:65-100): bounds == nullptr means unbounded in every dimension
    static Result<RealVectorStateSpace, StateSpaceError> create(std::size_t dimension,
                                                                const std::vector<std::pair<double, double>>* bounds_option) {
        RealVectorStateSpace s;
        s.dimension = dimension;
        if (bounds_option) {
            if (bounds_option->size() != dimension)
                return StateSpaceError{StateSpaceError::DimensionMismatch, dimension, bounds_option->size()};
            for (auto& b : *bounds_option)
                if (b.first >= b.second) return StateSpaceError{StateSpaceError::InvalidBound, 0, 0, b.first, b.second};
            s.bounds = *bounds_option;
        } else {
            if (dimension == 0) return StateSpaceError{StateSpaceError::ZeroDimensionUnbounded};
            const double inf = std::numeric_limits<double>::infinity();
            s.bounds.assign(dimension, {-inf, inf});
        }
        return s;
    }
    // rvss.rs:121-129
    void set_longest_valid_segment_fraction(double fraction) {
        if (fraction > 0.0 && fraction <= 1.0) longest_valid_segment_fraction = fraction;
        else if (fraction <= 0.0) longest_valid_segment_fraction = 0.0;
        else longest_valid_segment_fraction = 1.0;
    }
    // StateSpace::distance (rvss.rs:137-155), evaluated by the HIP library (oxhip_distance_batch)
    double distance(const RealVectorState& a, const RealVectorState& b, int device = 0) const {
        double out = std::numeric_limits<double>::quiet_NaN();
        if (a.values.size() == dimension && b.values.size() == dimension)
            (void)oxhip_distance_batch(device, (uint32_t)dimension, a.values.data(), b.values.data(), 1, &out);
        return out;
    }
    // StateSpace::interpolate (rvss.rs:161-186), evaluated by the HIP library (oxhip_interpolate_batch)
    void interpolate(const RealVectorState& from, const RealVectorState& to, double t, RealVectorState& out_state,
                     int device = 0) const {
        out_state.values.assign(dimension, std::numeric_limits<double>::quiet_NaN());
        (void)oxhip_interpolate_batch(device, (uint32_t)dimension, from.values.data(), to.values.data(), &t, 1,
                                      out_state.values.data());
    }
};

// ---- validity.rs: the reference's trait plus what a device-describable checker exposes
struct Sphere { std::vector<double> centre; double radius; };   // valid iff distance(centre, p) > radius
struct Box { std::vector<double> lo, hi; };                      // invalid iff lo_k <= p_k <= hi_k for all k
class StateValidityChecker {
  public:
    virtual ~StateValidityChecker() = default;
    // is_valid (validity.rs:39-48) is answered on the device from the description below
    virtual std::vector<Sphere> spheres() const { return {}; }
    virtual std::vector<Box> boxes() const { return {}; }
};

// ---- goal.rs: Goal / GoalRegion / GoalSampleableRegion collapsed onto the device-describable ball goal
class GoalSampleableRegion {
  public:
    virtual ~GoalSampleableRegion() = default;
    virtual RealVectorState target() const = 0;   // sample_goal() returns this state (README.md:160-162)
    virtual double radius() const = 0;            // is_satisfied(s) = distance(s, target) <= radius
};

// ---- problem_definition.rs / planner.rs
struct ProblemDefinition {
    std::shared_ptr<RealVectorStateSpace> space;
    std::vector<RealVectorState> start_states;
    std::shared_ptr<GoalSampleableRegion> goal;
};
struct Path { std::vector<RealVectorState> states; };   // Path(pub Vec<S>)

}  // namespace base

namespace geometric {

// RRT (rrt.rs:53-62) for one planning problem, grown on the GPU.
class RRT {
  public:
    double max_distance;
    double goal_bias;
    // build-defined termination and RNG key (the reference stops on wall clock only and cannot be seeded)
    uint32_t max_nodes = 10000;
    uint64_t seed = 0;
    uint64_t problem_id = 0;
    int device = 0;

    RRT(double max_distance_, double goal_bias_) : max_distance(max_distance_), goal_bias(goal_bias_) {}  // rrt.rs:75-83
    virtual ~RRT() { reset(); }
    RRT(const RRT&) = delete;
    RRT& operator=(const RRT&) = delete;

    // Planner::setup (rrt.rs:140-156)
    void setup(std::shared_ptr<base::ProblemDefinition> problem_def, std::shared_ptr<base::StateValidityChecker> validity_checker) {
        reset();
        pd_ = std::move(problem_def);
        vc_ = std::move(validity_checker);
        last_status_ = OXHIP_OK;
        if (!pd_ || !vc_ || !pd_->space || !pd_->goal || pd_->start_states.empty()) { last_status_ = OXHIP_ERR_BAD_ARG; return; }
        const auto& sp = *pd_->space;
        oxhip_rrt_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.dim = (uint32_t)sp.dimension;
        for (std::size_t k = 0; k < sp.bounds.size() && k < OXHIP_MAX_DIM; ++k) {
            cfg.bounds[2 * k] = sp.bounds[k].first;
            cfg.bounds[2 * k + 1] = sp.bounds[k].second;
        }
        cfg.max_distance = max_distance;
        cfg.goal_bias = goal_bias;
        cfg.lvs_fraction = sp.longest_valid_segment_fraction;
        cfg.n_problems = 1;
        cfg.max_nodes = max_nodes;
        cfg.stop_at_goal = 1;
        cfg.kernel = OXHIP_KERNEL_AUTO;
        cfg.seed = seed;
        cfg.first_problem_id = problem_id;
        cfg.device = device;
        cfg.planner = planner_kind();
        cfg.search_radius = search_radius_value();
        if ((last_status_ = oxhip_rrt_batch_create(&cfg, &batch_)) != OXHIP_OK) return;
        std::vector<double> c, r, lo, hi;
        for (auto& s : vc_->spheres()) { c.insert(c.end(), s.centre.begin(), s.centre.end()); r.push_back(s.radius); }
        for (auto& b : vc_->boxes()) { lo.insert(lo.end(), b.lo.begin(), b.lo.end()); hi.insert(hi.end(), b.hi.begin(), b.hi.end()); }
        if (!r.empty() && (last_status_ = oxhip_rrt_batch_set_spheres(batch_, c.data(), r.data(), (uint32_t)r.size())) != OXHIP_OK) return;
        if (!lo.empty() && (last_status_ = oxhip_rrt_batch_set_boxes(batch_, lo.data(), hi.data(), (uint32_t)(lo.size() / sp.dimension))) != OXHIP_OK) return;
        const auto target = pd_->goal->target();
        const double radius = pd_->goal->radius();
        last_status_ = oxhip_rrt_batch_setup(batch_, pd_->start_states[0].values.data(), target.values.data(), &radius);
    }

    // Planner::solve (rrt.rs:158-227)
    Result<base::Path, base::PlanningError> solve(std::chrono::duration<double> timeout) {
        if (!batch_ || last_status_ != OXHIP_OK) return base::PlanningError::PlannerUninitialised;  // rrt.rs:160-163
        // rrt.rs:172-174: `start_time.elapsed() > timeout` holds at the first check for a zero Duration (Rust's Duration has
        // no negative values or NaN: both are treated like zero here)
        if (!(timeout.count() > 0.0)) return base::PlanningError::Timeout;
        int32_t st = OXHIP_ERR_NO_SOLUTION_FOUND;
        last_status_ = oxhip_rrt_batch_solve(batch_, 1ull << 40, timeout.count(), 0, &st);
        if (last_status_ == OXHIP_ERR_PLANNER_UNINITIALISED) return base::PlanningError::PlannerUninitialised;
        if (last_status_ != OXHIP_OK) return base::PlanningError::NoSolutionFound;
        if (st == OXHIP_ERR_TIMEOUT) return base::PlanningError::Timeout;
        if (st != OXHIP_OK) return base::PlanningError::NoSolutionFound;
        uint32_t len = 0;
        (void)oxhip_rrt_batch_get_path(batch_, 0, nullptr, 0, &len);
        const std::size_t dim = pd_->space->dimension;
        std::vector<double> flat((std::size_t)len * dim);
        if ((last_status_ = oxhip_rrt_batch_get_path(batch_, 0, flat.data(), len, &len)) != OXHIP_OK)
            return base::PlanningError::NoSolutionFound;
        base::Path path;
        for (uint32_t i = 0; i < len; ++i)
            path.states.emplace_back(std::vector<double>(flat.begin() + i * dim, flat.begin() + (i + 1) * dim));
        return path;
    }

    // batched StateValidityChecker::is_valid / RRT::check_motion on the device (test helpers)
    bool is_valid(const base::RealVectorState& s) const {
        uint8_t ok = 0;
        if (batch_) (void)oxhip_rrt_batch_is_valid(batch_, s.values.data(), 1, &ok);
        return ok != 0;
    }
    uint32_t num_nodes() const {
        uint32_t n = 0;
        if (batch_) (void)oxhip_rrt_batch_get_counts(batch_, nullptr, &n, nullptr, nullptr, nullptr, nullptr);
        return n;
    }
    int32_t last_status() const { return last_status_; }

  protected:
    virtual uint32_t planner_kind() const { return OXHIP_PLANNER_RRT; }
    virtual double search_radius_value() const { return 0.0; }
    oxhip_rrt_batch* batch() const { return batch_; }

  private:
    void reset() {
        if (batch_) (void)oxhip_rrt_batch_destroy(batch_);
        batch_ = nullptr;
    }
    oxhip_rrt_batch* batch_ = nullptr;
    std::shared_ptr<base::ProblemDefinition> pd_;
    std::shared_ptr<base::StateValidityChecker> vc_;
    int32_t last_status_ = OXHIP_ERR_PLANNER_UNINITIALISED;
};

// RRTConnect (rrt_connect.rs:49-95): same Planner surface, two trees grown towards each other.
class RRTConnect : public RRT {
  public:
    RRTConnect(double max_distance_, double goal_bias_) : RRT(max_distance_, goal_bias_) {}  // rrt_connect.rs:86-95

  protected:
    uint32_t planner_kind() const override { return OXHIP_PLANNER_RRT_CONNECT; }
};

// RRTStar (rrt_star.rs:39-52): RRT with choose-parent and rewire inside `search_radius`.
class RRTStar : public RRT {
  public:
    double search_radius;
    RRTStar(double max_distance_, double goal_bias_, double search_radius_)   // rrt_star.rs:68-77
        : RRT(max_distance_, goal_bias_), search_radius(search_radius_) {}

    // Node::cost (rrt_star.rs:26) of every node, in node order
    std::vector<double> costs() const {
        uint32_t n = 0;
        if (!batch() || oxhip_rrt_batch_get_costs(batch(), 0, nullptr, 0, &n) != OXHIP_OK) return {};
        std::vector<double> c(n);
        if (oxhip_rrt_batch_get_costs(batch(), 0, c.data(), n, &n) != OXHIP_OK) return {};
        return c;
    }

  protected:
    uint32_t planner_kind() const override { return OXHIP_PLANNER_RRT_STAR; }
    double search_radius_value() const override { return search_radius; }
};

// PRM (prm.rs:48-57): roadmap built and queried on the GPU.
class PRM {
  public:
    double timeout;             // seconds of roadmap construction (prm.rs:50); the device path also stops at max_milestones
    double connection_radius;   // prm.rs:52
    // build-defined termination and RNG key
    uint32_t max_milestones = 16384;
    uint64_t max_samples = 0;   // 0 = unlimited
    uint64_t seed = 0;
    uint64_t stream = 0;
    int device = 0;

    PRM(double timeout_, double connection_radius_) : timeout(timeout_), connection_radius(connection_radius_) {}  // prm.rs:70-78
    ~PRM() { reset(); }
    PRM(const PRM&) = delete;
    PRM& operator=(const PRM&) = delete;

    // Planner::setup (prm.rs:217-225)
    void setup(std::shared_ptr<base::ProblemDefinition> problem_def, std::shared_ptr<base::StateValidityChecker> validity_checker) {
        reset();
        pd_ = std::move(problem_def);
        last_status_ = OXHIP_OK;
        if (!pd_ || !validity_checker || !pd_->space || !pd_->goal || pd_->start_states.empty()) { last_status_ = OXHIP_ERR_BAD_ARG; return; }
        const auto& sp = *pd_->space;
        oxhip_prm_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.dim = (uint32_t)sp.dimension;
        for (std::size_t k = 0; k < sp.bounds.size() && k < OXHIP_MAX_DIM; ++k) {
            cfg.bounds[2 * k] = sp.bounds[k].first;
            cfg.bounds[2 * k + 1] = sp.bounds[k].second;
        }
        cfg.timeout = timeout;
        cfg.connection_radius = connection_radius;
        cfg.lvs_fraction = sp.longest_valid_segment_fraction;
        cfg.max_milestones = max_milestones;
        cfg.max_samples = max_samples;
        cfg.seed = seed;
        cfg.stream = stream;
        cfg.device = device;
        if ((last_status_ = oxhip_prm_create(&cfg, &prm_)) != OXHIP_OK) return;
        std::vector<double> c, r, lo, hi;
        for (auto& s : validity_checker->spheres()) { c.insert(c.end(), s.centre.begin(), s.centre.end()); r.push_back(s.radius); }
        for (auto& b : validity_checker->boxes()) { lo.insert(lo.end(), b.lo.begin(), b.lo.end()); hi.insert(hi.end(), b.hi.begin(), b.hi.end()); }
        if (!r.empty() && (last_status_ = oxhip_prm_set_spheres(prm_, c.data(), r.data(), (uint32_t)r.size())) != OXHIP_OK) return;
        if (!lo.empty() && (last_status_ = oxhip_prm_set_boxes(prm_, lo.data(), hi.data(), (uint32_t)(lo.size() / sp.dimension))) != OXHIP_OK) return;
        const auto target = pd_->goal->target();
        last_status_ = oxhip_prm_setup(prm_, pd_->start_states[0].values.data(), target.values.data(), pd_->goal->radius());
    }

    // PRM::set_problem_definition (prm.rs:88-90): keeps the roadmap
    void set_problem_definition(std::shared_ptr<base::ProblemDefinition> pd) {
        pd_ = std::move(pd);
        if (prm_ && pd_ && pd_->goal && !pd_->start_states.empty()) {
            const auto target = pd_->goal->target();
            last_status_ = oxhip_prm_set_problem(prm_, pd_->start_states[0].values.data(), target.values.data(), pd_->goal->radius());
        }
    }

    // PRM::construct_roadmap (prm.rs:96-154)
    Result<bool, base::PlanningError> construct_roadmap() {
        if (!prm_ || last_status_ != OXHIP_OK) return base::PlanningError::PlannerUninitialised;  // prm.rs:97-104
        last_status_ = oxhip_prm_construct_roadmap(prm_);
        if (last_status_ != OXHIP_OK) return base::PlanningError::PlannerUninitialised;
        return true;
    }

    // Planner::solve (prm.rs:227-307)
    Result<base::Path, base::PlanningError> solve(std::chrono::duration<double> timeout_) {
        if (!prm_) return base::PlanningError::PlannerUninitialised;
        uint32_t len = 0;
        int32_t st = oxhip_prm_solve(prm_, timeout_.count(), nullptr, 0, &len);
        if (st != OXHIP_OK) return to_error(st);
        const std::size_t dim = pd_->space->dimension;
        std::vector<double> flat((std::size_t)len * dim);
        if ((st = oxhip_prm_solve(prm_, timeout_.count(), flat.data(), len, &len)) != OXHIP_OK) return to_error(st);
        base::Path path;
        for (uint32_t i = 0; i < len; ++i)
            path.states.emplace_back(std::vector<double>(flat.begin() + i * dim, flat.begin() + (i + 1) * dim));
        return path;
    }

    // PRM::get_roadmap (prm.rs:82-84): milestone count and, per node, its `edges`
    uint32_t num_milestones() const {
        uint32_t n = 0;
        if (prm_) (void)oxhip_prm_get_sizes(prm_, &n, nullptr, nullptr);
        return n;
    }
    struct Roadmap { std::vector<base::RealVectorState> states; std::vector<std::vector<uint32_t>> edges; };
    Roadmap get_roadmap() const {
        Roadmap rm;
        if (!prm_) return rm;
        uint32_t n = 0;
        uint64_t e = 0;
        (void)oxhip_prm_get_sizes(prm_, &n, &e, nullptr);
        const std::size_t dim = pd_->space->dimension;
        std::vector<double> flat((std::size_t)n * dim);
        std::vector<uint64_t> off((std::size_t)n + 1);
        std::vector<uint32_t> nb((std::size_t)e);
        if (oxhip_prm_get_roadmap(prm_, flat.data(), n, off.data(), nb.data(), e) != OXHIP_OK) return rm;
        for (uint32_t i = 0; i < n; ++i) {
            rm.states.emplace_back(std::vector<double>(flat.begin() + i * dim, flat.begin() + (i + 1) * dim));
            rm.edges.emplace_back(nb.begin() + off[i], nb.begin() + off[i + 1]);
        }
        return rm;
    }
    int32_t last_status() const { return last_status_; }

  private:
    static base::PlanningError to_error(int32_t st) {
        switch (st) {
            case OXHIP_ERR_TIMEOUT: return base::PlanningError::Timeout;
            case OXHIP_ERR_PLANNER_UNINITIALISED: return base::PlanningError::PlannerUninitialised;
            case OXHIP_ERR_INVALID_START_STATE: return base::PlanningError::InvalidStartState;
            case OXHIP_ERR_UNSAMPLED_STATE_SPACE: return base::PlanningError::UnsampledStateSpace;
            default: return base::PlanningError::NoSolutionFound;
        }
    }
    void reset() {
        if (prm_) (void)oxhip_prm_destroy(prm_);
        prm_ = nullptr;
    }
    oxhip_prm* prm_ = nullptr;
    std::shared_ptr<base::ProblemDefinition> pd_;
    int32_t last_status_ = OXHIP_ERR_PLANNER_UNINITIALISED;
};

}  // namespace geometric
}  // namespace oxmpl
