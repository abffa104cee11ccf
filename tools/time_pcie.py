"""Measures pf_set_field / pf_get_field (pageable host buffer <-> HBM over PCIe) for a 512^3 field: the PCIe-inclusive
cost a caller would pay if it moved the state across the C ABI every step (the benchmark does not: state stays in HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver  # noqa: E402

with PhaseFieldSolver(dim=3, n=512, h=1.0) as s:
    s.set_ic_bm1()
    c = s.get_c()
    for name, fn in (("pf_get_field", lambda: s.get_c()), ("pf_set_field", lambda: s.set_c(c))):
        fn()
        t = time.perf_counter()
        for _ in range(3):
            fn()
        el = (time.perf_counter() - t) / 3
        print("%s 512^3 (1 GiB): %.1f ms = %.1f GB/s" % (name, el * 1e3, c.nbytes / el / 1e9), flush=True)
