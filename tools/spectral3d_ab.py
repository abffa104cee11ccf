"""n^3 semi-implicit spectral step: hand-written LDS-FFT passes vs rocFFT (PFHIP_SPECTRAL_3D=rocfft), same initial
state: max field difference, diagnostics, ms per step.  Usage on the GPU box: python tools/spectral3d_ab.py [n=512]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd.solver import PhaseFieldSolver


N = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def make(env):
    os.environ.pop("PFHIP_SPECTRAL_3D", None)
    os.environ.update(env)
    return PhaseFieldSolver(dim=3, n=N, h=1.0, scheme="spectral")


rng = np.random.default_rng(5)
c0 = 0.5 + 0.05 * rng.standard_normal((N, N, N), dtype=np.float32).astype(np.float64)
res = {}
for name, env in (("lds-fft", {"PFHIP_SPECTRAL_3D": "lds"}), ("rocfft", {"PFHIP_SPECTRAL_3D": "rocfft"})):
    with make(env) as s:
        s.set_c(c0)
        d0 = s.diagnostics()
        s.step(1e-2, 3)
        s.sync()
        t0 = time.perf_counter()
        s.step(1e-2, 20)
        s.sync()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        d1 = s.diagnostics()
        res[name] = (s.get_c(), d0, d1, ms)
    print("%-8s %.3f ms/step  %.3e cell-updates/s  F0=%.10e F=%.10e C=%.10e" % (name, ms, N ** 3 / ms * 1e3, d0[0], d1[0], d1[1]),
          flush=True)
a, b = res["lds-fft"][0], res["rocfft"][0]
print("max |c_lds - c_rocfft| = %.3e  (max |c| %.3f)" % (np.abs(a - b).max(), np.abs(b).max()))
print("F rel diff after 23 steps: %.3e" % (abs(res["lds-fft"][2][0] - res["rocfft"][2][0]) / abs(res["rocfft"][2][0])))
