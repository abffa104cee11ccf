"""2-D semi-implicit spectral step, microseconds per step (wall clock over a python loop of pf_step(dt, n)):
sizes x {radix-8 wave FFT (512 only), radix-2^2 multi-wave LDS FFT, rocFFT}.  Usage on the GPU box:
python tools/time_spectral_2d.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver


def run(n, steps, env):
    for k in ("PFHIP_SPECTRAL_2D",):
        os.environ.pop(k, None)
    os.environ.update(env)
    with PhaseFieldSolver(dim=2, n=n, h=1.0, scheme="spectral") as s:
        s.set_ic_bm1(0.5, 0.05)
        s.step(1e-2, 50)
        s.sync()
        t0 = time.perf_counter()
        s.step(1e-2, steps)
        s.sync()
        el = time.perf_counter() - t0
        F, C, _ = s.diagnostics()
    return el / steps * 1e6, F


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for n in (256, 512, 1024):
        for name, env in (("lds-fft (default)", {}), ("rocFFT", {"PFHIP_SPECTRAL_2D": "rocfft"})):
            us, F = run(n, steps, env)
            print("n=%4d %-20s %8.2f us/step  %.3e cell-updates/s  F=%.10f" % (n, name, us, n * n / us * 1e6, F),
                  flush=True)


if __name__ == "__main__":
    main()
