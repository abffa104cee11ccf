"""Probe: does RCCL accept two ranks on ONE device?  (No: 'Duplicate GPU detected' -- which is why the N > 1 path is
validated on a 1-GPU box with a world-size-1 group instead, tests/nccl_single_rank_worker.py.)
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/nccl_same_gpu_probe.py"""
import os, torch, torch.distributed as dist
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    a = torch.full((4,), float(rank), device="cuda", dtype=torch.float64)
    b = torch.empty_like(a)
    ops=[dist.P2POp(dist.isend, a, (rank+1)%world), dist.P2POp(dist.irecv, b, (rank-1)%world)]
    for r in dist.batch_isend_irecv(ops): r.wait()
    torch.cuda.synchronize()
    print(rank, "P2P ok", b.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(rank, "FAILED:", repr(e)[:300], flush=True)
