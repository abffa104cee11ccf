"""Does the relative placement of the two time-level buffers matter?  One torch allocation holds both c buffers; the
second starts `pad` bytes after the first ends.  Same process, same physical block, pads swept.
Usage on the GPU box: python tools/buffer_offset_ab.py [n=512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pfhubbenchmarks_amd import lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = L.load()
elems = n ** 3
pads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [k * 32768 for k in range(0, 34)]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
big = torch.zeros(2 * elems + (max(pads) // 8) + 1024, dtype=torch.float64, device="cuda")
base = big.data_ptr()
print("base VA %x" % base)
for rep in range(reps):
    for pad in pads:
        cfg = L.default_config(3, n, 1.0)
        cfg.ext_c[0] = C.c_void_p(base)
        cfg.ext_c[1] = C.c_void_p(base + elems * 8 + pad)
        h = C.c_void_p()
        L.check(lib.pf_create(C.byref(cfg), C.byref(h)))
        L.check(lib.pf_set_ic_bm1(h, 0.5, 0.05), h)
        L.check(lib.pf_step(h, 5e-4, 40, None), h)
        lib.pf_sync(h)
        lib.pf_timing_enable(h, 1)
        L.check(lib.pf_step(h, 5e-4, 40, None), h)
        ms, cnt = C.c_double(), C.c_int64()
        lib.pf_timing_read(h, C.byref(ms), C.byref(cnt))
        lib.pf_destroy(h)
        print("pad %9d B: %.4f ms  %.0f GB/s" % (pad, ms.value, 16.0 * elems / ms.value / 1e6), flush=True)
