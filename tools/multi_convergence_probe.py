"""BM2 / BM3: explicit FD schemes vs the reference's algorithm (GPU fem_be), Richardson limits -- probe for the tolerances
asserted in tests/test_gpu_parity.py.  Usage on the GPU box: python tools/multi_convergence_probe.py [bm2|bm3]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd import verification as V

model = sys.argv[1] if len(sys.argv) > 1 else "bm2"
if model == "bm2":
    ts, dt = (0.02, 0.04), 0.01
else:
    ts, dt = (0.5, 1.0), 0.25
t0 = time.time()
fd, lim = V.multi_fd_limit(model, ts, log=print)
print("fd limits per h:", {k: v[:, 0] for k, v in lim.items()}, "-> (h, dt) -> 0:", fd[:, 0], "second:", fd[:, 1], "%.1f s" % (time.time() - t0))
t0 = time.time()
fem, runs = V.multi_fem_dt_limit(model, ts, dt, log=print)
print("fem_be at the reference mesh, dt -> 0: F", fem[:, 0], "second:", fem[:, 1], "%.1f s" % (time.time() - t0))
print("rel F:", np.abs(fd[:, 0] - fem[:, 0]) / np.abs(fem[:, 0]), " rel second:", np.abs(fd[:, 1] - fem[:, 1]) / np.abs(fem[:, 1]))
for N, v in lim.items():
    print("  fd h = %g alone vs fem: rel F %s" % (V.L_DOM[model] / N, np.abs(v[:, 0] - fem[:, 0]) / np.abs(fem[:, 0])))
if model == "bm2" and len(sys.argv) > 2 and sys.argv[2] == "full":
    # the reference's algorithm on the refined mesh h = 1 (242 406 unknowns, 46 GB of dense blocks): Richardson in h as well
    t0 = time.time()
    f1 = V.multi_energy(model, "fem_be", 200, dt, ts)
    f2 = V.multi_energy(model, "fem_be", 200, dt / 2, ts)
    print("bm2 fem_be h = 1 dt = %g / %g: F %s / %s  %.1f s" % (dt, dt / 2, f1[:, 0], f2[:, 0], time.time() - t0))
    # remove the dt^2 term measured at h = 2, then linear extrapolation in dt (as verification.fem_be_limit does)
    _, b2dt2 = V.quad_extrapolate(*runs)
    g1 = 2.0 * (f2 - b2dt2 / 4.0) - (f1 - b2dt2)
    star = (4.0 * g1 - fem) / 3.0
    print("fem_be h = 1, dt -> 0:", g1[:, 0], " (h, dt) -> 0:", star[:, 0])
    print("rel F, FD limit vs reference-algorithm limit:", np.abs(fd[:, 0] - star[:, 0]) / np.abs(star[:, 0]))
