"""512^3 spectral step: radix-8 one-wave column kernel (default) vs the multi-thread radix-2^2 column kernel
(PFHIP_FFT3D_GENERIC512=1, 8 and 4 columns per workgroup).  Usage on the GPU box: python tools/spectral3d_generic512.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd.solver import PhaseFieldSolver

N = 512
rng = np.random.default_rng(5)
c0 = 0.5 + 0.05 * rng.standard_normal((N, N, N), dtype=np.float32).astype(np.float64)
ref = None
for name, env in (("radix-8 wave", {}), ("generic cw8", {"PFHIP_FFT3D_GENERIC512": "1", "PFHIP_FFT3D_CWG": "8"}),
                  ("generic cw4", {"PFHIP_FFT3D_GENERIC512": "1", "PFHIP_FFT3D_CWG": "4"}),
                  ("generic row", {"PFHIP_FFT3D_ROW": "generic"}), ("radix-8 again", {})):
    for k in ("PFHIP_FFT3D_GENERIC512", "PFHIP_FFT3D_CWG", "PFHIP_FFT3D_ROW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with PhaseFieldSolver(dim=3, n=N, h=1.0, scheme="spectral") as s:
        s.set_c(c0)
        s.step(1e-2, 40)
        s.sync()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            s.step(1e-2, 20)
            s.sync()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        c = s.get_c()
        if ref is None:
            ref = c
        print("%-13s %.3f ms/step   max |c - c_radix8| = %.2e" % (name, best, np.abs(c - ref).max()), flush=True)
