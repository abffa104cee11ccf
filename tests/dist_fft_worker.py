"""Worker of tests/test_dist_gloo.py (FFT slab modes): one gloo rank running FFTSlabSolver over the numpy mirror of the
library's distributed state machine.  Usage: python tests/dist_fft_worker.py <out> <mode>   mode: spectral | spectral_mirror | bm6 | bm6_elim | bm6_mirror"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from oracle_engine import OracleFFTSlabEngine
    from pfhubbenchmarks_amd.solver import FFTSlabSolver
    out, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = (16, 12, 8) if world <= 4 else (16, 2 * world, 2 * world)    # slab FFT: ny and nz divisible by the ranks
    mirror = mode in ("spectral_mirror", "bm6_mirror")
    eng = OracleFFTSlabEngine(n, 1.0, world, rank, scheme="spectral" if mode.startswith("spectral") else "fd",
                              model="bm6" if mode.startswith("bm6") else "bm1", eliminate_phi=mode == "bm6_elim",
                              dirichlet=(n[0] // 2 + 1, n[1] // 2 + 1) if mode == "bm6_mirror" else None)
    rng = np.random.default_rng(4)
    if mirror:
        # the no-flux box of (9, 7, 5) nodes on its even extension = the 16 x 12 x 8 lattice: the slabs form a ring over the
        # lattice planes, the physical planes are the first 5 of the gathered stack (HipFFTSlabEngine(bc="mirror") does this)
        from oracle.multi_fd import even_extend
        phys = 0.5 + 0.05 * rng.standard_normal((n[2] // 2 + 1, n[1] // 2 + 1, n[0] // 2 + 1))
        full = even_extend(phys)                                   # x and y
        full = np.concatenate([full, full[-2:0:-1]], axis=0)       # and z
        assert full.shape == (n[2], n[1], n[0])
    else:
        full = 0.5 + 0.05 * rng.standard_normal((n[2], n[1], n[0]))
    eng.set_local(full[eng.z0:eng.z0 + eng.nz])
    if mirror:                                                     # (what gather_field looks at)
        eng.bc, eng.nz_physical = "mirror", phys.shape[0]
    s = FFTSlabSolver(eng)
    dt = 1e-2 if mode.startswith("spectral") else 1e-3
    d0 = s.diagnostics()
    s.step(dt, 3)
    d1 = s.diagnostics()
    s.step(dt, 1)
    field = s.gather_field()
    if rank == 0:
        np.savez(out, field=field, d0=np.array(d0), d1=np.array(d1), full=full, dt=dt)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
