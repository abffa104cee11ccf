"""Is the slab (multi-GPU) step host-bound?  Times the Python enqueue loop of SlabSolver.step (no device sync inside) against
the device time of the same steps, RCCL group of size 1 on one GPU.  Usage: python tools/slab_host_time.py [narrow|wide]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from pfhubbenchmarks_amd.solver import HipSlabEngine, SlabSolver

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
wide = (sys.argv[1] if len(sys.argv) > 1 else "wide") == "wide"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = HipSlabEngine((512, 512, 512), 1.0, 1, 0, 0, wide=wide)
eng.set_ic_bm1(0.5, 0.05)
s = SlabSolver(eng)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    s.step(5e-4, 20)
    eng.sync()
torch.cuda.synchronize()
for n in (20, 100, 400):
    t0 = time.perf_counter()
    s.step(5e-4, n)
    t_host = time.perf_counter() - t0
    eng.sync()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s halo, %4d steps: host enqueue %.1f us/step, device-complete %.1f us/step" % (
        "wide" if wide else "narrow", n, t_host / n * 1e6, t_all / n * 1e6), flush=True)
dist.destroy_process_group()
