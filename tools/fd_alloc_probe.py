"""Does the fused FD stencil's speed depend on the allocation it runs on (as the spectral column passes' does)?  Several
512^3 FD solvers side by side, each timed in turn, twice.  Usage on the GPU box: python tools/fd_alloc_probe.py [k=6]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

N = 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sol = []
for i in range(K):
    s = PhaseFieldSolver(dim=3, n=N, h=1.0, kernel="fused")
    s.set_ic_bm1()
    sol.append(s)
for rnd in range(2):
    for i, s in enumerate(sol):
        s.step(5e-4, 400)
        s.sync()
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            s.step(5e-4, 100)
            s.sync()
            t.append((time.perf_counter() - t0) / 100 * 1e3)
        print("round %d solver %d: %.4f ms/step (%s)" % (rnd, i, sorted(t)[2], " ".join("%.4f" % v for v in t)), flush=True)
for s in sol:
    s.close()
