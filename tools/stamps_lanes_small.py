import os, sys
sys.path.insert(0, "/root/repo")
from oxmpl_amd import capi, scenarios
sc = scenarios.config2()
for N in (500, 2000):
    gpu = scenarios.make_batch(sc, 1024, N, False, 42, 0, 0, capi.KERNEL_LANES)
    gpu.enable_stamps(True)
    gpu.solve(10 ** 7)
    s = gpu.stamps()
    itg = int(s[7])
    print("N", N, "kernel %.3f ms" % gpu.last_timing()["kernel_ms"], "iters", itg, "rounds", int(s[5]), "lanes/round %.1f" % (int(s[6]) / max(1, int(s[5]))), "commits/round %.1f" % (itg / max(1, int(s[5]))), "exact", int(s[4]))
    print("  resolver wait %.0f work %.0f per iteration; per round %.0f; lifetime %.3f ms" % (int(s[1]) / itg, int(s[2]) / itg, int(s[2]) / max(1, int(s[5])), int(s[14]) / 1e5))
    PH = ("combine+candidates", "ring fold", "coords+steer", "sphere filter", "motion check", "prefix", "commit")
    print("  phases per round:", ", ".join("%s %.0f" % (nm, int(s[32 + i]) / max(1, int(s[5]))) for i, nm in enumerate(PH)))
    print("  exact path per event %.0f" % (int(s[3]) / max(1, int(s[4]))))
    print("  scanner wait", " ".join("%6.0f" % (int(v) / itg) for v in s[16:24]), " work", " ".join("%6.0f" % (int(v) / itg) for v in s[24:32]))
    gpu.close()
