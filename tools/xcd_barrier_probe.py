"""What would a persistent multi-step kernel pay per phase boundary if all its workgroups sat on ONE XCD?  (VERDICT r02
item 8: the 512^2 spectral step is two dependent launches, ~6 us of its 12.8 us is launch floor; the chip-wide grid barrier
costs 7-8 us -- tools/grid_barrier_probe.py.)  Prints us per round (= two single-XCD barriers, with and without a 128-byte
record hand-off per workgroup through L2 with L1-bypassing loads) for several grid / workgroup sizes.
Usage on the GPU box: python tools/xcd_barrier_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd import lib as L

lib = L.load()
print("grid x threads  handoff  participants  us/round (2 barriers)  stale")
for nb, nt in ((256, 64), (256, 256), (512, 64), (512, 256), (1024, 64)):
    for handoff in (0, 1):
        us, n, st = C.c_double(), C.c_int(), C.c_int()
        rc = lib.pfk_xcd_barrier_probe(nb, nt, 2000, handoff, C.byref(us), C.byref(n), C.byref(st))
        if rc != 0:
            print("%4d x %3d  %d  failed: %s" % (nb, nt, handoff, (lib.pf_last_error(None) or b"").decode()))
            continue
        print("%4d x %3d      %d        %4d          %8.3f            %d" % (nb, nt, handoff, n.value, us.value, st.value), flush=True)
