"""Times the 2-D BM6 step (Poisson solve + coupled FD step) of the reference-sized problem; prints us/step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver  # noqa: E402

for n, bc in ((101, "mirror"), (257, "mirror"), (512, "periodic")):
    with PhaseFieldSolver(dim=2, n=n, h=1.0, bc=bc, model="bm6") as s:
        s.set_ic_bm6()
        s.step(1e-3, 50)
        s.sync()
        t = time.perf_counter()
        s.step(1e-3, 2000)
        s.sync()
        el = time.perf_counter() - t
        print("bm6 2-D n=%d %s: %.1f us/step" % (n, bc, el / 2000 * 1e6), flush=True)
