"""Kernel time of the fused FD step on an arbitrary periodic box (e.g. the per-GPU slab shapes of the scaling runs).
Usage: python tools/time_shape.py nx ny nz [steps=200]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

nx, ny, nz = (int(v) for v in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
with PhaseFieldSolver(dim=3, n=(nx, ny, nz), h=1.0) as s:
    s.set_ic_bm1(0.5, 0.05)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        s.step(5e-4, 50)
        s.sync()
    s.timing(True)
    t0 = time.perf_counter()
    s.step(5e-4, steps)
    s.sync()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms, n = s.timing_read()
    cells = nx * ny * nz
    print("%d x %d x %d: kernel %.4f ms (wall %.4f) -> %.3e cell-updates/s, %.0f GB/s = %.3f of 8 TB/s" % (
        nx, ny, nz, ms, wall, cells / (ms * 1e-3), 16.0 * cells / (ms * 1e-3) / 1e9, 16.0 * cells / (ms * 1e-3) / 8e12))
