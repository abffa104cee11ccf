"""Per-block kernel time of the fused step from a cold start (does the first ~100 ms run slower than steady state?).
Usage on the GPU box: python tools/cold_start_ramp.py [n=512] [block=25] [blocks=16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
block = int(sys.argv[2]) if len(sys.argv) > 2 else 25
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 16
with PhaseFieldSolver(dim=3, n=n, h=1.0) as s:
    s.set_ic_bm1(0.5, 0.05)
    s.sync()
    time.sleep(2.0)                     # let the GPU go idle
    out = []
    for b in range(blocks):
        s.timing(True)
        t0 = time.perf_counter()
        s.step(5e-4, block)
        s.sync()
        wall = (time.perf_counter() - t0) / block * 1e3
        ms, _ = s.timing_read()
        s.timing(False)
        out.append((ms, wall))
    print("n=%d, blocks of %d steps: kernel ms (wall ms)" % (n, block))
    print("  ".join("%.4f(%.4f)" % o for o in out))
