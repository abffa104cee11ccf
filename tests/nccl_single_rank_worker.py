"""Runs the REAL multi-GPU host path (torch.distributed backend "nccl" = RCCL, SlabSolver / FFTSlabSolver, ghost planes,
all-to-all) on one GPU: world size 1 with pf_config.force_slab, so the rank is its own ring neighbour.  RCCL refuses
two ranks on one device, so this is as close to the 8-GPU job as a 1-GPU box gets: every API call, stream dependency
and buffer view of the production path is exercised; only the transport is trivial.
Invoked by tests/test_gpu_parity.py::test_nccl_path_single_rank (env: RANK=0 WORLD_SIZE=1 MASTER_*)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from pfhubbenchmarks_amd.solver import (FFTSlabSolver, HipFFTSlabEngine, HipSlabEngine, PhaseFieldSolver,
                                            SlabSolver)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    n = (128, 32, 24)
    rng = np.random.default_rng(17)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])

    # --- FD slab path: overlapped ghost exchange + begin/finish
    eng = HipSlabEngine(n, 1.0, 1, 0, 0)
    eng.set_local(full)
    s = SlabSolver(eng)
    with PhaseFieldSolver(dim=3, n=n, h=1.0) as ref:
        ref.set_c(full)
        d0, r0 = s.diagnostics(), ref.diagnostics()
        assert abs(d0[0] - r0[0]) <= 1e-13 * abs(r0[0]) and abs(d0[1] - r0[1]) <= 1e-13 * abs(r0[1]), (d0, r0)
        s.step(1e-3, 7)
        ref.step(1e-3, 7)
        eng.sync()
        assert np.array_equal(eng.get_local(), ref.get_c()), "FD slab path differs from the single-domain path"
        d1, r1 = s.diagnostics(), ref.diagnostics()
        assert abs(d1[0] - r1[0]) <= 1e-13 * abs(r1[0])
        s.step(1e-3, 2)                 # a step right after diagnostics re-uses the fresh ghosts
        ref.step(1e-3, 2)
        eng.sync()
        assert np.array_equal(s.gather_field(), ref.get_c())
    eng.close()
    print("FD slab path over NCCL: ok", flush=True)

    # --- the same with PF_FLAG_WIDE_HALO (4 ghost planes every second step) through SlabSolver over RCCL, incl. a step
    # right after diagnostics and the no-flux line with both walls on this rank; and with the boundary strips on a stream
    # of their own (pf_set_strip_stream: interior || exchange -> strips), narrow and wide halo
    for bc, nn, wide, side in (("periodic", n, True, False), ("mirror", (65, 17, 12), True, False),
                               ("periodic", n, False, True), ("periodic", n, True, True), ("mirror", (65, 17, 12), True, True)):
        fw = 0.5 + 0.05 * rng.standard_normal(nn[::-1])
        eng = HipSlabEngine(nn, 1.0, 1, 0, 0, bc=bc, wide=wide)
        if side:
            eng.use_strip_stream()
        eng.set_local(fw)
        s = SlabSolver(eng)
        with PhaseFieldSolver(dim=3, n=nn, h=1.0, bc=bc) as ref:
            ref.set_c(fw)
            s.step(1e-3, 5)
            ref.step(1e-3, 5)
            d1, r1 = s.diagnostics(), ref.diagnostics()
            assert abs(d1[0] - r1[0]) <= 1e-13 * abs(r1[0]) and abs(d1[1] - r1[1]) <= 1e-13 * abs(r1[1]), (bc, d1, r1)
            s.step(1e-3, 4)
            ref.step(1e-3, 4)
            eng.sync()
            assert np.array_equal(s.gather_field(), ref.get_c()), "slab path differs (%s wide=%s strips-on-side-stream=%s)" % (bc, wide, side)
        eng.close()
    print("wide-halo slab path and strip stream over NCCL: ok", flush=True)

    # --- BASELINE.json config 4's per-GPU shape: one 1024 x 1024 x 128 slab of the 1024^3 box over 8 GPUs (8 MiB planes,
    # 16 MiB ghost messages, 32-bit in-plane offsets at their largest) on the production path, against the plain
    # single-domain handle on the same planes, bit for bit
    nb = (1024, 1024, 128)
    big = (0.5 + 0.05 * rng.standard_normal(nb[::-1], dtype=np.float32)).astype(np.float64)
    eng = HipSlabEngine(nb, 1.0, 1, 0, 0)
    eng.set_local(big)
    s = SlabSolver(eng)
    with PhaseFieldSolver(dim=3, n=nb, h=1.0) as ref:
        ref.set_c(big)
        del big
        s.step(5e-4, 5)
        ref.step(5e-4, 5)
        eng.sync()
        assert np.array_equal(eng.get_local(), ref.get_c()), "1024 x 1024 x 128 slab differs from the single-domain path"
        d1, r1 = s.diagnostics(), ref.diagnostics()
        assert abs(d1[0] - r1[0]) <= 1e-13 * abs(r1[0]) and abs(d1[1] - r1[1]) <= 1e-13 * abs(r1[1]), (d1, r1)
    eng.close()
    torch.cuda.empty_cache()
    print("1024 x 1024 x 128 slab over NCCL: ok", flush=True)

    # --- no-flux box (PF_BC_MIRROR) as a one-slab line: both walls on this rank, nothing to exchange, library mirrors
    nn = (65, 17, 12)
    fm = 0.5 + 0.05 * rng.standard_normal(nn[::-1])
    eng = HipSlabEngine(nn, 1.0, 1, 0, 0, bc="mirror")
    assert (eng.rank_lo, eng.rank_hi) == (-1, -1)
    eng.set_local(fm)
    s = SlabSolver(eng)
    with PhaseFieldSolver(dim=3, n=nn, h=1.0, bc="mirror") as ref:
        ref.set_c(fm)
        d0, r0 = s.diagnostics(), ref.diagnostics()
        assert abs(d0[0] - r0[0]) <= 1e-13 * abs(r0[0]) and abs(d0[1] - r0[1]) <= 1e-13 * abs(r0[1]), (d0, r0)
        s.step(1e-3, 5)
        ref.step(1e-3, 5)
        eng.sync()
        assert np.array_equal(s.gather_field(), ref.get_c()), "mirror-bc slab line differs from the even extension"
    eng.close()
    print("mirror-bc slab line over NCCL: ok", flush=True)

    # --- slab FFT modes: all_to_all_single + halo requests from the library's state machine
    for scheme, model, dt, elim in (("spectral", "bm1", 1e-2, False), ("fd", "bm6", 1e-3, False),
                                    ("fd", "bm6", 1e-3, True)):
        eng = HipFFTSlabEngine(n, 1.0, 1, 0, 0, scheme=scheme, model=model, eliminate_phi=elim)
        eng.set_local(full)
        s = FFTSlabSolver(eng)
        with PhaseFieldSolver(dim=3, n=n, h=1.0, scheme=scheme, model=model, eliminate_phi=elim) as ref:
            ref.set_c(full)
            d0, r0 = s.diagnostics(), ref.diagnostics()
            assert abs(d0[0] - r0[0]) <= 1e-11 * abs(r0[0]), (scheme, model, d0, r0)
            s.step(dt, 4)
            ref.step(dt, 4)
            eng.sync()
            err = np.abs(eng.get_local() - ref.get_c()).max()
            assert err <= 1e-12, (scheme, model, err)
            d1, r1 = s.diagnostics(), ref.diagnostics()
            assert abs(d1[0] - r1[0]) <= 1e-11 * abs(r1[0]) and abs(d1[1] - r1[1]) <= 1e-13 * abs(r1[1])
        eng.close()
        print("slab FFT mode %s/%s%s over NCCL: ok" % (scheme, model, " (phi eliminated)" if elim else ""), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_SINGLE_RANK_OK", flush=True)


if __name__ == "__main__":
    main()
