"""Back-to-back dependent launches of a trivial kernel (pfk_stream_copy of 2 doubles) on one stream: the per-launch
floor that any multi-kernel time step pays.  Usage on the GPU box: python tools/launch_floor.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pfhubbenchmarks_amd import lib as L

lib = L.load()
a = torch.ones(1024, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
st = torch.cuda.current_stream()
for n in (2, 1024):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20000):
            lib.pfk_stream_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, C.c_void_p(st.cuda_stream))
        torch.cuda.synchronize()
        print("n=%d: %.2f us per launch" % (n, (time.perf_counter() - t0) / 20000 * 1e6), flush=True)
