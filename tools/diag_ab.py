"""A/B of the diagnostics reduction kernel forms (pfk_set_tuning keys 8 / 9) on one resident field: wall time per
pf_diagnostics call (kernel + 64-byte read-back + sync) and the implied read bandwidth at 8 B/cell.
Usage on the GPU box: python tools/diag_ab.py [n=512] [model=bm1|bm6]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd import lib as L
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = sys.argv[2] if len(sys.argv) > 2 else "bm1"
lib = L.load()
with PhaseFieldSolver(dim=3, n=n, h=1.0, model=model) as s:
    (s.set_ic_bm6 if model == "bm6" else s.set_ic_bm1)()
    s.step(5e-4, 20)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:      # pre-heat
        s.step(5e-4, 20)
        s.sync()
    ref = None
    bpc = 16.0 if model == "bm6" else 8.0
    for variant in (0, 12, 13, 21, 22, 23, 41, 42, 43):
        for target in ((0,) if variant == 0 else (512, 1024, 2048)):
            assert lib.pfk_set_tuning(8, variant) == 0
            if target:
                assert lib.pfk_set_tuning(9, target) == 0
            d = s.diagnostics()
            if ref is None:
                ref = d
            assert all(abs(a - b) <= 1e-13 * abs(b) for a, b in zip(d[:2], ref[:2])), (variant, d, ref)
            for _ in range(20):
                s.diagnostics()
            reps = 100
            t0 = time.perf_counter()
            for _ in range(reps):
                s.diagnostics()
            ms = (time.perf_counter() - t0) / reps * 1e3
            print("n=%d %s variant %2d target %4d: %.4f ms/call  -> %.0f GB/s at %g B/cell (incl. launch + read-back)"
                  % (n, model, variant, target, ms, bpc * n ** 3 / ms / 1e6, bpc), flush=True)
