"""Is the slab (multi-GPU) path host-bound?  RCCL group of size 1 on one GPU: wall time per step with and without
waiting for the GPU, and the time of the pieces of SlabSolver.step.  Usage: python tools/slab_host_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
import torch
import torch.distributed as dist

from pfhubbenchmarks_amd.solver import HipSlabEngine, SlabSolver

torch.cuda.set_device(0)
opts = None
if os.environ.get("HIPRIO") == "1":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    print("high-priority NCCL stream")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
shape = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (512, 512, 512)
print("grid", shape, "-> 4 messages of %.1f MB per step" % (shape[0] * shape[1] * 2 * 8 / 1e6))
from pfhubbenchmarks_amd import lib as L
if len(sys.argv) > 2:
    L.load().pfk_set_tuning(0, int(sys.argv[2]))
    print("fused variant", sys.argv[2])
eng = HipSlabEngine(shape, 1.0, 1, 0, 0)
eng.set_ic_bm1(0.5, 0.05)
s = SlabSolver(eng)
s.step(5e-4, 60)
eng.sync(); torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
s.step(5e-4, N)
t_enq = time.perf_counter() - t0
eng.sync(); torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("enqueue %.1f us/step, end-to-end %.1f us/step" % (t_enq / N * 1e6, t_all / N * 1e6))
# pieces (host time only)
tp = tb = tw = tf = 0.0
for _ in range(N):
    with eng.stream_context():
        a = time.perf_counter(); reqs = s._post_exchange()
        b = time.perf_counter(); eng.step_begin(5e-4)
        c = time.perf_counter()
        for r in reqs:
            r.wait()
        d = time.perf_counter(); eng.step_finish()
        e = time.perf_counter()
    tp += b - a; tb += c - b; tw += d - c; tf += e - d
eng.sync(); torch.cuda.synchronize()
print("host us/step: post_exchange %.1f  step_begin %.1f  wait %.1f  step_finish %.1f"
      % (tp / N * 1e6, tb / N * 1e6, tw / N * 1e6, tf / N * 1e6))
# the same two launches per step WITHOUT the exchange (ghosts go stale: timing only)
eng.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
with eng.stream_context():
    for _ in range(N):
        eng.step_begin(5e-4)
        eng.step_finish()
eng.sync(); torch.cuda.synchronize()
print("begin/finish without exchange: %.1f us/step" % ((time.perf_counter() - t0) / N * 1e6))
# exchange posted but on the critical path only through the stream wait (as in SlabSolver.step) -- repeated for reference
t0 = time.perf_counter()
s.step(5e-4, N)
eng.sync(); torch.cuda.synchronize()
print("SlabSolver.step: %.1f us/step" % ((time.perf_counter() - t0) / N * 1e6))
# self-exchange by plain device copies on a side stream (same dependency structure, no RCCL kernel)
side = torch.cuda.Stream()
nz, g = eng.nz, eng.ghost
eng.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    buf = eng.buffers[eng.cur]
    e1 = torch.cuda.Event(); e2 = torch.cuda.Event()
    e1.record(eng.stream)
    side.wait_event(e1)
    with torch.cuda.stream(side):
        buf[0:g].copy_(buf[nz:nz + g])
        buf[nz + g:nz + 2 * g].copy_(buf[g:2 * g])
        e2.record(side)
    with eng.stream_context():
        eng.step_begin(5e-4)
        eng.stream.wait_event(e2)
        eng.step_finish()
eng.sync(); torch.cuda.synchronize()
print("exchange by side-stream copies: %.1f us/step" % ((time.perf_counter() - t0) / N * 1e6))
dist.destroy_process_group()
