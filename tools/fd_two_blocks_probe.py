#!/usr/bin/env python
"""The FD headline step with its two time levels (a) in the library's one placed block, (b) in two separate pf_device_malloc
allocations, (c) the same with a transient spacer allocated between them: does the 'different physical regions' effect of the
spectral arrays reach the headline kernel?   usage: fd_two_blocks_probe.py [spacer MiB ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfhubbenchmarks_amd import lib as _lib
from pfhubbenchmarks_amd.solver import PhaseFieldSolver
L = _lib.load()
n = 512
elems = n ** 3


def timed(s):
    s.set_ic_bm1()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        s.step(1e-3, 50); s.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); s.step(1e-3, 100); s.sync()
        ts.append((time.perf_counter() - t0) / 100 * 1e3)
    return sorted(ts)[3]


def alloc(b):
    p = C.c_void_p()
    assert L.pf_device_malloc(C.byref(p), C.c_size_t(b)) == 0
    return p


with PhaseFieldSolver(dim=3, n=n, h=1.0, scheme="fd") as s:
    print("one placed block          %.4f ms/step" % timed(s), flush=True)
for spacer in [0] + [int(a) for a in sys.argv[1:]]:
    a = alloc(8 * elems)
    sp = alloc(spacer << 20) if spacer else None
    b = alloc(8 * elems)
    if sp:
        L.pf_device_free(sp)
    with PhaseFieldSolver(dim=3, n=n, h=1.0, scheme="fd", ext_c=(a.value, b.value)) as s:
        print("two blocks, spacer %5d MiB  %.4f ms/step   (distance %+.3f GiB)" % (spacer, timed(s), (b.value - a.value) / 2 ** 30), flush=True)
    L.pf_device_free(a); L.pf_device_free(b)
