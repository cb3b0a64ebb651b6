import sys
sys.path.insert(0, '.')
from oxmpl_amd import capi, scenarios
sc2 = scenarios.config2()
for P in (1, 1024):
    for budget in (1, 8, 16, 32, 64):
        ts = []
        for rep in range(3):
            g = scenarios.make_batch(sc2, P, 10000, True, 43, 0, 0, 0, capi.PLANNER_RRT_CONNECT)
            g.solve(budget)
            ts.append(g.last_timing()["kernel_ms"])
            it = g.counts()["iterations"]
            g.close()
        print("P=%4d budget %3d: kernel %.4f ms (min of 3; max iterations run %d)" % (P, budget, min(ts), int(it.max())))
