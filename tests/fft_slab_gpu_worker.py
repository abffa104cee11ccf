"""One rank of a 2-process run of the slab-FFT modes (spectral scheme, BM6 with and without phi eliminated) on ONE GPU:
real HIP engines (HipFFTSlabEngine), the collectives the library asks for (all_to_all_single, ghost isend/irecv) served
by gloo on the GPU tensors.  Rank 0 compares the gathered field with the single-domain PhaseFieldSolver.
Usage: python tests/fft_slab_gpu_worker.py <mode>      (env: RANK WORLD_SIZE MASTER_*)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from pfhubbenchmarks_amd.solver import FFTSlabSolver, HipFFTSlabEngine, PhaseFieldSolver
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scheme, model, dt, elim = {"spectral": ("spectral", "bm1", 1e-2, False), "bm6": ("fd", "bm6", 1e-3, False),
                               "bm6_elim": ("fd", "bm6", 1e-3, True), "spectral_mirror": ("spectral", "bm1", 1e-2, False),
                               "bm6_mirror": ("fd", "bm6", 1e-3, False)}[mode]       # the reference's BCs (bench6.py:77-90)
    bc = "mirror" if mode.endswith("mirror") else "periodic"
    n = (65, 17, 13) if bc == "mirror" else (128, 32, 24)      # mirror: nodes of the no-flux box -> lattice 128 x 32 x 24
    rng = np.random.default_rng(19)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    eng = HipFFTSlabEngine(n, 1.0, world, rank, 0, scheme=scheme, model=model, eliminate_phi=elim, bc=bc)
    if bc == "mirror":
        eng.set_global(full)
    else:
        eng.set_local(full[eng.z0:eng.z0 + eng.nz])
    s = FFTSlabSolver(eng)
    d0 = s.diagnostics()
    s.step(dt, 4)
    d1 = s.diagnostics()
    s.step(dt, 1)
    eng.sync()
    field = s.gather_field()
    if rank == 0:
        with PhaseFieldSolver(dim=3, n=n, h=1.0, scheme=scheme, model=model, eliminate_phi=elim, bc=bc) as ref:
            ref.set_c(full)
            r0 = ref.diagnostics()
            ref.step(dt, 4)
            r1 = ref.diagnostics()
            ref.step(dt, 1)
            err = np.abs(field - ref.get_c()).max()
        assert abs(d0[0] - r0[0]) <= 1e-11 * abs(r0[0]) and abs(d1[0] - r1[0]) <= 1e-11 * abs(r1[0]), (d0, r0, d1, r1)
        assert abs(d1[1] - r1[1]) <= 1e-13 * abs(r1[1])
        assert err <= 1e-12, err
        print("FFT_SLAB_GPU_OK %s err %.2e" % (mode, err), flush=True)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
