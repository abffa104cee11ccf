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
