#!/usr/bin/env python
"""Does the fast / slow state of the spectral 512^3 step follow the STREAM or the ALLOCATION?

Handles are created one after the other on caller-supplied streams (cfg.stream): first N handles all on stream A,
then N on stream B, then alternating, each with its own fresh allocation; optionally the previous handle is kept alive
so that the next one cannot get the same memory back.  A state that follows the stream shows as a constant per stream;
one that follows the allocation changes from handle to handle on the same stream.
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--handles", type=int, default=5)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--keep", type=int, default=0, help="keep the previous handle alive while the next is timed")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--chunked", action="store_true", help="leave PFHIP_FFT3D_CHUNK alone (default: whole-box passes, the placement-sensitive form)")
a = ap.parse_args()
if not a.chunked:
    os.environ.setdefault("PFHIP_FFT3D_CHUNK", "0")
import torch
from pfhubbenchmarks_amd.solver import PhaseFieldSolver


def timed(s):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        s.step(1e-2, 20); s.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); s.step(1e-2, a.steps); s.sync()
        ts.append((time.perf_counter() - t0) / a.steps * 1e3)
    return sorted(ts)[2]


streams = [torch.cuda.Stream() for _ in range(a.streams)]
order = [(i, k) for i in range(a.streams) for k in range(a.handles)] + [(k % a.streams, 100 + k) for k in range(2 * a.streams)]
prev = None
for si, k in order:
    st = streams[si]
    s = PhaseFieldSolver(dim=3, n=512, h=1.0, scheme="spectral", model="bm1", stream=st.cuda_stream)
    s.set_ic_bm1()
    ms = timed(s)
    print("stream %d (0x%x) handle %3d  c at 0x%x   %.4f ms/step" % (si, st.cuda_stream, k, s.c_device_ptr() if hasattr(s, "c_device_ptr") else 0, ms), flush=True)
    if prev is not None:
        prev.close()
    if a.keep:
        prev = s
    else:
        s.close(); prev = None
if prev is not None:
    prev.close()
