"""Cost of one grid-wide barrier inside a cooperative launch vs the dependent-launch floor (tools/launch_floor.py).
Usage on the GPU box: python tools/grid_barrier_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd import lib as L

lib = L.load()
for nb, nt in ((129, 128), (256, 64), (256, 256), (64, 256), (512, 128)):
    us = C.c_double()
    rc = lib.pfk_grid_barrier_probe(nb, nt, 2000, C.byref(us))
    print("blocks %4d x %3d threads: rc %d, %.2f us per grid barrier" % (nb, nt, rc, us.value), flush=True)
    if rc != 0:
        print(lib.pf_last_error(None).decode())
        break
