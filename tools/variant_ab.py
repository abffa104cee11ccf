"""In-process A/B of fused-kernel variants on the SAME buffers (process-to-process spread of +-5 % hides a 2-3 %
difference between variants: physical page placement changes with every allocation).
Usage on the GPU box: python tools/variant_ab.py [n=512] [variants=14,2,13] [reps=6] [steps=60]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd import lib as L
from pfhubbenchmarks_amd.solver import PhaseFieldSolver


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    variants = (sys.argv[2] if len(sys.argv) > 2 else "14,2,13").split(",")   # "variant[/target_wgs[/min_chunk]]"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    lib = L.load()
    res = {v: [] for v in variants}
    with PhaseFieldSolver(dim=3, n=n, h=1.0) as s:
        s.set_ic_bm1(0.5, 0.05)
        s.step(5e-4, 10)
        s.sync()
        for r in range(reps):
            for v in variants:
                f = [int(x) for x in v.split("/")]
                lib.pfk_set_tuning(0, f[0])
                lib.pfk_set_tuning(1, f[1] if len(f) > 1 and f[1] > 0 else 256)
                lib.pfk_set_tuning(2, f[2] if len(f) > 2 and f[2] > 0 else 16)
                s.step(5e-4, 3)
                s.sync()
                s.timing(True)
                s.step(5e-4, steps)
                s.sync()
                ms, _ = s.timing_read()
                s.timing(False)
                res[v].append(ms)
    for v in variants:
        a = np.array(res[v])
        print("n=%d variant %-10s: median %.4f ms  min %.4f  max %.4f  -> %.0f GB/s (median)"
              % (n, v, np.median(a), a.min(), a.max(), 16.0 * n ** 3 / np.median(a) / 1e6), flush=True)


if __name__ == "__main__":
    main()
