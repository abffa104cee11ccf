"""Is the speed of the 512^3 spectral step a property of the HIP stream (hardware queue) its handle runs on, inside a
PyTorch process?  8 torch streams are created and KEPT; for each, a whole-box handle (PFHIP_FFT3D_CHUNK=0) bound to that stream
is created, timed and destroyed, twice over.  Then the overlap matrix of the streams (two spin kernels at once)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pfhubbenchmarks_amd.solver import PhaseFieldSolver

os.environ["PFHIP_FFT3D_CHUNK"] = sys.argv[1] if len(sys.argv) > 1 else "0"
OWN = len(sys.argv) > 2 and sys.argv[2] == "own"      # the library creates its own stream (8 torch streams alive beside it)
streams = [torch.cuda.Stream() for _ in range(8)]
for rnd in range(2):
    for i, st in enumerate(streams):
        with PhaseFieldSolver(dim=3, n=512, h=1.0, scheme="spectral", stream=None if OWN else st.cuda_stream) as s:
            s.set_ic_bm1()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                s.step(1e-2, 20)
                s.sync()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                s.step(1e-2, 20)
                s.sync()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
            print("round %d stream %d (0x%x): %.4f ms/step" % (rnd, i, st.cuda_stream, sorted(ts)[1]), flush=True)
x = torch.zeros(1, device="cuda")
print("overlap (us for two 200 us sleeps at once):")
for i in range(8):
    row = []
    for j in range(8):
        if j <= i:
            row.append("     ")
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[i]):
            torch.cuda._sleep(480000)     # ~200 us at 2.4 GHz
        with torch.cuda.stream(streams[j]):
            torch.cuda._sleep(480000)
        streams[i].synchronize()
        streams[j].synchronize()
        row.append("%5.0f" % ((time.perf_counter() - t0) * 1e6))
    print("  %d %s" % (i, " ".join(row)))
