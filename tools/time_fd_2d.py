"""Times the 2-D multi-step FD path at 512^2 (pf_step with many steps); used with rocprofv3 --stats to split kernel time
from launch overhead."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver  # noqa: E402

with PhaseFieldSolver(dim=2, n=512, h=1.0) as s:
    s.set_ic_bm1()
    s.step(1e-3, 101)
    s.sync()
    t = time.perf_counter()
    s.step(1e-3, 20001)
    s.sync()
    el = time.perf_counter() - t
    print("fd 2-D 512^2: %.2f us/step" % (el / 20001 * 1e6), flush=True)
