// C++ twin of the reference's integration test oxmpl/tests/prm_rvss_tests.rs (same scene, same
// assertions), written against include/oxmpl/oxmpl.hpp: [0,10]^2, a wall x in [4.75,5.25], y in [2,8],
// start (1,5), circular goal (9,5) r=0.5, PRM::new(5.0, 0.5), 5 s query timeout.
// Exit code 0 = all assertions hold; 77 = no GPU (the planner refuses to run: no CPU fallback).
#include <cmath>
#include <cstdio>
#include <memory>

#include "oxmpl/oxmpl.hpp"

using namespace oxmpl::base;
using oxmpl::geometric::PRM;

// prm_rvss_tests.rs:16-36
struct WallObstacleChecker : StateValidityChecker {
    double wall_x_pos, wall_y_min, wall_y_max, wall_thickness;
    WallObstacleChecker(double x, double ymin, double ymax, double t) : wall_x_pos(x), wall_y_min(ymin), wall_y_max(ymax), wall_thickness(t) {}
    std::vector<Box> boxes() const override {
        return {Box{{wall_x_pos - wall_thickness / 2.0, wall_y_min}, {wall_x_pos + wall_thickness / 2.0, wall_y_max}}};
    }
    bool is_valid(const RealVectorState& s) const {   // the reference's predicate, for the path validator below
        const double x = s.values[0], y = s.values[1];
        return !(x >= wall_x_pos - wall_thickness / 2.0 && x <= wall_x_pos + wall_thickness / 2.0 && y >= wall_y_min && y <= wall_y_max);
    }
};

// prm_rvss_tests.rs:39-69
struct CircularGoalRegion : GoalSampleableRegion {
    RealVectorState target_;
    double radius_;
    CircularGoalRegion(RealVectorState t, double r) : target_(std::move(t)), radius_(r) {}
    RealVectorState target() const override { return target_; }
    double radius() const override { return radius_; }
};

#define CHECK(cond, msg) do { if (!(cond)) { std::printf("FAILED: %s\n", msg); return 1; } } while (0)

// prm_rvss_tests.rs:72-107
static bool is_path_valid(const Path& path, const RealVectorStateSpace& space, const WallObstacleChecker& checker) {
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
    auto space = std::make_shared<RealVectorStateSpace>(RealVectorStateSpace::create(2, &bounds).unwrap());
    RealVectorState start_state({1.0, 5.0});
    auto goal_definition = std::make_shared<CircularGoalRegion>(RealVectorState({9.0, 5.0}), 0.5);
    auto problem_definition = std::make_shared<ProblemDefinition>(ProblemDefinition{space, {start_state}, goal_definition});
    auto validity_checker = std::make_shared<WallObstacleChecker>(5.0, 2.0, 8.0, 0.5);
    CHECK(validity_checker->is_valid(start_state), "Start state should be valid!");
    CHECK(validity_checker->is_valid(goal_definition->target()), "Goal target should be valid!");

    PRM planner(5.0, 0.5);
    CHECK(planner.solve(std::chrono::seconds(5)).err() == PlanningError::PlannerUninitialised, "solve before setup");           // prm.rs:229-236
    CHECK(planner.construct_roadmap().err() == PlanningError::PlannerUninitialised, "construct_roadmap before setup");           // prm.rs:97-104
    if (oxhip_device_count(&ndev) != OXHIP_OK) {
        planner.setup(problem_definition, validity_checker);
        CHECK(planner.last_status() == OXHIP_ERR_NO_DEVICE, "without a GPU setup must fail loudly");
        CHECK(planner.construct_roadmap().is_err(), "no CPU fallback");
        std::printf("no GPU: refused as designed\n");
        return 77;
    }
    planner.setup(problem_definition, validity_checker);
    CHECK(planner.last_status() == OXHIP_OK, "setup");
    CHECK(planner.solve(std::chrono::seconds(5)).err() == PlanningError::UnsampledStateSpace, "solve before construct_roadmap");  // prm.rs:239-241
    CHECK(planner.construct_roadmap().is_ok(), "Issue constructing roadmap!");
    CHECK(planner.num_milestones() > 0, "Roadmap was not populated.");

    auto result = planner.solve(std::chrono::seconds(5));
    CHECK(result.is_ok(), "Planner failed to find a solution when one should exist.");
    const Path& path = result.unwrap();
    std::printf("Found path with %zu states (%u milestones).\n", path.states.size(), planner.num_milestones());
    CHECK(!path.states.empty(), "Path should not be empty");
    CHECK(space->distance(path.states.front(), start_state) < 1e-9, "Path should start at the start state");
    CHECK(space->distance(path.states.back(), goal_definition->target()) <= goal_definition->radius(), "Path should end in the goal region");
    CHECK(is_path_valid(path, *space, *validity_checker), "The returned path was found to be invalid.");

    // get_roadmap (prm.rs:82-84): edges are symmetric and every list ascends
    auto rm = planner.get_roadmap();
    CHECK(rm.states.size() == planner.num_milestones() && rm.edges.size() == rm.states.size(), "roadmap size");
    for (std::size_t i = 0; i < rm.edges.size(); i += 97)
        for (std::size_t e = 0; e < rm.edges[i].size(); ++e) {
            const uint32_t nb = rm.edges[i][e];
            CHECK(e == 0 || rm.edges[i][e - 1] < nb, "edges ascend");
            bool back = false;
            for (uint32_t v : rm.edges[nb]) back = back || v == i;
            CHECK(back, "edges are symmetric (prm.rs:143-145)");
            CHECK(space->distance(rm.states[i], rm.states[nb]) < 0.5, "edge within the connection radius");
        }
    // a start inside the wall: InvalidStartState (prm.rs:243-246), on the same roadmap (prm.rs:88-90)
    planner.set_problem_definition(std::make_shared<ProblemDefinition>(ProblemDefinition{space, {RealVectorState({5.0, 5.0})}, goal_definition}));
    CHECK(planner.solve(std::chrono::seconds(5)).err() == PlanningError::InvalidStartState, "invalid start");
    std::printf("PRM planner test passed!\n");
    return 0;
}
