"""n^3 spectral step with the generic column kernel at 4 / 8 columns per workgroup (PFHIP_FFT3D_CWG): ms per step.
Usage on the GPU box: python tools/spectral3d_cwg.py [n=256]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd.solver import PhaseFieldSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(5)
c0 = 0.5 + 0.05 * rng.standard_normal((N, N, N), dtype=np.float32).astype(np.float64)
for cwg in ("0", "4", "8"):
    os.environ["PFHIP_FFT3D_CWG"] = cwg
    os.environ["PFHIP_SPECTRAL_3D"] = "lds"
    with PhaseFieldSolver(dim=3, n=N, h=1.0, scheme="spectral") as s:
        s.set_c(c0)
        s.step(1e-2, 5)
        s.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            s.step(1e-2, 20)
            s.sync()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        print("n = %d  CWG = %s: %.3f ms/step  C = %.10e" % (N, cwg, best, s.diagnostics()[1]), flush=True)
