"""512^2 spectral step: two launches per step vs the single-XCD multi-step kernel (PFHIP_SPECTRAL_PERSIST=1), microseconds per
step over long pf_step calls, same process.  Usage on the GPU box: python tools/time_spectral_persist.py [steps=2000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for rnd in range(2):
    for mode in ("0", "1"):
        os.environ["PFHIP_SPECTRAL_PERSIST"] = mode
        with PhaseFieldSolver(dim=2, n=512, h=1.0, scheme="spectral") as s:
            s.set_ic_bm1(0.5, 0.05)
            s.step(1e-2, 200)
            s.sync()
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                s.step(1e-2, steps)
                s.sync()
                t.append((time.perf_counter() - t0) / steps * 1e6)
            F = s.diagnostics()[0]
        print("PFHIP_SPECTRAL_PERSIST=%s  %.2f us/step (3 calls of %d steps: %s)  F=%.10f" % (
            mode, sorted(t)[1], steps, " ".join("%.2f" % v for v in t), F), flush=True)
