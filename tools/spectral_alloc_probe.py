"""Is the 512^3 spectral step's speed a property of the ALLOCATION it runs on?  Creates several solvers side by side (each
with its own 3.3 GB spectrum block), times each in turn, twice.  Usage on the GPU box: python tools/spectral_alloc_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PFHIP_SPECTRAL_VERBOSE"] = "1"
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

N = 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sol = []
for i in range(K):
    s = PhaseFieldSolver(dim=3, n=N, h=1.0, scheme="spectral")
    s.set_ic_bm1()
    sol.append(s)
for rnd in range(2):
    for i, s in enumerate(sol):
        s.step(1e-2, 40)
        s.sync()
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            s.step(1e-2, 20)
            s.sync()
            t.append((time.perf_counter() - t0) / 20 * 1e3)
        print("round %d solver %d: %.3f ms/step" % (rnd, i, sorted(t)[1]), flush=True)
for s in sol:
    s.close()
