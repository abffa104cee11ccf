import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
import torch, torch.distributed as dist
from pfhubbenchmarks_amd.solver import HipSlabEngine, SlabSolver
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
shape = tuple(int(x) for x in sys.argv[1].split(","))
eng = HipSlabEngine(shape, 1.0, 1, 0, 0); eng.set_ic_bm1(0.5, 0.05)
s = SlabSolver(eng); s.step(5e-4, 80); eng.sync(); torch.cuda.synchronize()
dist.destroy_process_group()
