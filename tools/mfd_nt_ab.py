"""In-process A/B of the BM2 / BM3 streaming kernels with plain vs non-temporal output stores (pfk_set_tuning key 10), same
handle, alternating blocks.  Usage on the GPU box: python tools/mfd_nt_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd import lib as L
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

lib = L.load()
for model, dt in (("bm3", 2e-3), ("bm2", 2e-4)):
    with PhaseFieldSolver(dim=3, n=512, h=1.0, scheme="fd", model=model) as s:
        (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            s.step(dt, 20)
            s.sync()
        for rnd in range(3):
            for nt in (0, 1):
                lib.pfk_set_tuning(10, nt)
                s.step(dt, 10)
                s.sync()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    s.step(dt, 40)
                    s.sync()
                    ts.append((time.perf_counter() - t0) / 40 * 1e3)
                print("%s round %d nt=%d  %.4f ms/step (min %.4f max %.4f)" % (model, rnd, nt, sorted(ts)[2], min(ts), max(ts)), flush=True)
lib.pfk_set_tuning(10, 1)
