"""One rank of a multi-process slab run on ONE GPU with the peer-copy halo transport (IpcHaloTransport): real HIP
kernels, real inter-process ghost exchange (CUDA IPC), gloo only for bootstrap / reductions -- the 2- and 3-rank GPU
coverage that RCCL cannot give on a 1-GPU box (it refuses two ranks on one device).
Usage: python tests/ipc_slab_worker.py <out.npz> <bc> [fused]      (env: RANK WORLD_SIZE MASTER_*)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from pfhubbenchmarks_amd.solver import HipSlabEngine, SlabSolver
    out, bc = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nzg = int(os.environ.get("PF_TEST_NZ", "18"))
    n = (128, 24, nzg) if bc == "periodic" else (65, 13, nzg)
    rng = np.random.default_rng(41)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    mode = sys.argv[3] if len(sys.argv) > 3 else "split"
    eng = HipSlabEngine(n, 1.0, world, rank, 0, bc=bc, wide=mode.endswith("_wide"))
    eng.set_local(full[eng.z0:eng.z0 + eng.nz])
    if mode in ("p2p", "p2p_wide"):             # torch.distributed isend / irecv of the GPU ghost planes (gloo stages them on the host)
        s = SlabSolver(eng, transport="rccl")
    else:
        s = SlabSolver(eng, transport="ipc", fused=mode.startswith("fused"))
    d0 = s.diagnostics()
    s.step(1e-3, 25)
    d1 = s.diagnostics()
    s.step(1e-3, 2)               # steps right after diagnostics re-use the fresh ghosts
    eng.sync()
    if s.transport is not None:
        s.transport.check()
    if nzg % world == 0:
        field = s.gather_field()
        if rank == 0:
            np.savez(out, field=field, d0=np.array(d0), d1=np.array(d1), full=full)
    else:          # unequal slabs: every rank reports its own planes
        np.savez(out + ".rank%d.npz" % rank, local=eng.get_local(), z0=eng.z0, d0=np.array(d0), d1=np.array(d1),
                 full=full)
    dist.barrier()
    if s.transport is not None:
        s.transport.close()
    eng.close()
    dist.destroy_process_group()
    print("IPC_SLAB_OK rank %d" % rank, flush=True)


if __name__ == "__main__":
    main()
