// C++ twin of the reference's integration test oxmpl/tests/rrt_rvss_tests.rs (same scene, same
// assertions), written against include/oxmpl/oxmpl.hpp: [0,10]^2, a wall x in [4.75,5.25],
// y in [2,8], start (1,5), circular goal (9,5) r=0.5, RRT::new(0.5, 0.0), 5 s timeout.
// Exit code 0 = all assertions hold; 77 = no GPU (the planner refuses to run: no CPU fallback).
#include <cmath>
#include <cstdio>
#include <memory>

#include "oxmpl/oxmpl.hpp"

using namespace oxmpl::base;
using oxmpl::geometric::RRT;
using oxmpl::geometric::RRTConnect;
using oxmpl::geometric::RRTStar;

// rrt_rvss_tests.rs:17-36
struct WallObstacleChecker : StateValidityChecker {
    double wall_x_pos, wall_y_min, wall_y_max, wall_thickness;
    WallObstacleChecker(double x, double ymin, double ymax, double t) : wall_x_pos(x), wall_y_min(ymin), wall_y_max(ymax), wall_thickness(t) {}
    std::vector<Box> boxes() const override {
        return {Box{{wall_x_pos - wall_thickness / 2.0, wall_y_min}, {wall_x_pos + wall_thickness / 2.0, wall_y_max}}};
    }
};

// rrt_rvss_tests.rs:39-69
struct CircularGoalRegion : GoalSampleableRegion {
    RealVectorState target_;
    double radius_;
    CircularGoalRegion(RealVectorState t, double r) : target_(std::move(t)), radius_(r) {}
    RealVectorState target() const override { return target_; }
    double radius() const override { return radius_; }
};

#define CHECK(cond, msg) do { if (!(cond)) { std::printf("FAILED: %s\n", msg); return 1; } } while (0)

// rrt_rvss_tests.rs:72-107
static bool is_path_valid(const Path& path, const RealVectorStateSpace& space, const RRT& checker) {
    for (std::size_t i = 0; i + 1 < path.states.size(); ++i) {
        const auto& a = path.states[i];
        const auto& b = path.states[i + 1];
        if (!checker.is_valid(a)) return false;
        if (i + 1 == path.states.size() - 1 && !checker.is_valid(b)) return false;
        double extent = 0.0;
        for (auto& bd : space.bounds) extent += (bd.second - bd.first) * (bd.second - bd.first);
        const double lvsl = std::sqrt(extent) * space.longest_valid_segment_fraction;
        const double dist = space.distance(a, b);
        const std::size_t num_steps = (std::size_t)std::ceil(dist / lvsl);
        if (num_steps > 1) {
            RealVectorState interp = a;
            for (std::size_t j = 1; j <= num_steps; ++j) {
                space.interpolate(a, b, (double)j / (double)num_steps, interp);
                if (!checker.is_valid(interp)) return false;
            }
        }
    }
    return true;
}

int main() {
    int32_t ndev = 0;
    std::vector<std::pair<double, double>> bounds{{0.0, 10.0}, {0.0, 10.0}};
    auto new_rvss_result = RealVectorStateSpace::create(2, &bounds);
    CHECK(new_rvss_result.is_ok(), "Error creating new RealVectorState!");
    auto space = std::make_shared<RealVectorStateSpace>(new_rvss_result.unwrap());

    // constructor contract (real_vector_state_space.rs:54-93)
    std::vector<std::pair<double, double>> bad{{1.0, 1.0}, {0.0, 1.0}};
    CHECK(RealVectorStateSpace::create(2, &bad).err().kind == StateSpaceError::InvalidBound, "InvalidBound");
    CHECK(RealVectorStateSpace::create(3, &bounds).err().kind == StateSpaceError::DimensionMismatch, "DimensionMismatch");
    CHECK(RealVectorStateSpace::create(0, nullptr).err().kind == StateSpaceError::ZeroDimensionUnbounded, "ZeroDimensionUnbounded");
    CHECK(std::isinf(RealVectorStateSpace::create(3, nullptr).unwrap().bounds[0].second), "unbounded default");

    RealVectorState start_state({1.0, 5.0});
    auto goal_definition = std::make_shared<CircularGoalRegion>(RealVectorState({9.0, 5.0}), 0.5);
    auto problem_definition = std::make_shared<ProblemDefinition>(ProblemDefinition{space, {start_state}, goal_definition});
    auto validity_checker = std::make_shared<WallObstacleChecker>(5.0, 2.0, 8.0, 0.5);

    RRT planner(0.5, 0.0);
    CHECK(planner.solve(std::chrono::seconds(5)).err() == PlanningError::PlannerUninitialised, "solve before setup");  // rrt.rs:160-163
    if (oxhip_device_count(&ndev) != OXHIP_OK) {
        planner.setup(problem_definition, validity_checker);
        CHECK(planner.last_status() == OXHIP_ERR_NO_DEVICE, "without a GPU setup must fail loudly");
        CHECK(planner.solve(std::chrono::seconds(5)).is_err(), "no CPU fallback");
        std::printf("no GPU: refused as designed\n");
        return 77;
    }
    planner.setup(problem_definition, validity_checker);
    CHECK(planner.last_status() == OXHIP_OK, "setup");
    CHECK(planner.is_valid(start_state), "Start state should be valid!");
    CHECK(planner.is_valid(goal_definition->target()), "Goal target should be valid!");

    auto result = planner.solve(std::chrono::seconds(5));
    CHECK(result.is_ok(), "Planner failed to find a solution when one should exist.");
    const Path& path = result.unwrap();
    std::printf("Found path with %zu states (%u nodes).\n", path.states.size(), planner.num_nodes());
    CHECK(!path.states.empty(), "Path should not be empty");
    CHECK(space->distance(path.states.front(), start_state) < 1e-9, "Path should start at the start state");
    CHECK(space->distance(path.states.back(), goal_definition->target()) <= goal_definition->radius(), "Path should end in the goal region");
    CHECK(is_path_valid(path, *space, planner), "The returned path was found to be invalid.");

    // an unbounded space cannot be sampled: the reference panics on unwrap() (rrt.rs:183); here setup reports it
    auto unb = std::make_shared<RealVectorStateSpace>(RealVectorStateSpace::create(2, nullptr).unwrap());
    RRT p2(0.5, 0.0);
    p2.setup(std::make_shared<ProblemDefinition>(ProblemDefinition{unb, {start_state}, goal_definition}), validity_checker);
    CHECK(p2.last_status() == OXHIP_ERR_UNBOUNDED, "unbounded space");
    CHECK(p2.solve(std::chrono::seconds(1)).err() == PlanningError::PlannerUninitialised, "unbounded solve");
    // oxmpl/tests/rrt_connect_rvss_tests.rs: the same scene through RRTConnect::new(0.5, 0.0)
    RRTConnect pc(0.5, 0.0);
    pc.setup(problem_definition, validity_checker);
    CHECK(pc.last_status() == OXHIP_OK, "RRTConnect setup");
    auto rc = pc.solve(std::chrono::seconds(5));
    CHECK(rc.is_ok(), "RRTConnect failed to find a solution when one should exist.");
    const Path& pathc = rc.unwrap();
    CHECK(!pathc.states.empty(), "RRTConnect path should not be empty");
    CHECK(space->distance(pathc.states.front(), start_state) < 1e-9, "RRTConnect path should start at the start state");
    CHECK(space->distance(pathc.states.back(), goal_definition->target()) <= goal_definition->radius(), "RRTConnect path should end in the goal region");
    CHECK(is_path_valid(pathc, *space, pc), "The RRTConnect path was found to be invalid.");
    // oxmpl/tests/rrt_star_rvss_tests.rs: the same scene through RRTStar::new(0.5, 0.0, 0.25)
    RRTStar ps(0.5, 0.0, 0.25);
    ps.setup(problem_definition, validity_checker);
    CHECK(ps.last_status() == OXHIP_OK, "RRTStar setup");
    auto rs = ps.solve(std::chrono::seconds(5));
    CHECK(rs.is_ok(), "RRTStar failed to find a solution when one should exist.");
    const Path& paths = rs.unwrap();
    CHECK(!paths.states.empty(), "RRTStar path should not be empty");
    CHECK(space->distance(paths.states.front(), start_state) < 1e-9, "RRTStar path should start at the start state");
    CHECK(space->distance(paths.states.back(), goal_definition->target()) <= goal_definition->radius(), "RRTStar path should end in the goal region");
    CHECK(is_path_valid(paths, *space, ps), "The RRTStar path was found to be invalid.");
    CHECK(ps.costs().size() == ps.num_nodes() && ps.costs()[0] == 0.0, "RRTStar costs");
    std::printf("RRT planner test passed!\n");
    return 0;
}
