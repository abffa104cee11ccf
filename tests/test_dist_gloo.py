"""N > 1 path on CPU: world_size-2 (and 3) gloo jobs drive pfhubbenchmarks_amd.solver.SlabSolver (the product's
halo-exchange / overlap / all-reduce logic) over an oracle-backed engine and must reproduce the single-domain
result bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,halo", [(2, "narrow"), (3, "narrow"), (8, "narrow"), (2, "wide"), (3, "wide"), (8, "wide")])
def test_slab_solver_matches_single_domain(world, halo, tmp_path, orc):
    """halo = "wide": PF_FLAG_WIDE_HALO's protocol (4 ghost planes every second step) through SlabSolver; 5 steps so that
    both step kinds, an odd count and a step right after diagnostics are exercised"""
    out = str(tmp_path / "res.npz")
    nsteps = 4 if halo == "narrow" else 5
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out,
                                       str(nsteps), "periodic", halo], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = np.load(out)
    c = res["full"]
    F0, C0, _ = orc.diagnostics(c, h=1.0)
    np.testing.assert_allclose(res["d0"][:2], [F0, C0], rtol=1e-13)
    for _ in range(nsteps):
        c = orc.fd_step(c, 1e-3)
    F1, C1, _ = orc.diagnostics(c, h=1.0)
    np.testing.assert_allclose(res["d1"][:2], [F1, C1], rtol=1e-13)
    c = orc.fd_step(c, 1e-3)
    np.testing.assert_array_equal(res["field"], c)


@pytest.mark.parametrize("world,halo", [(2, "narrow"), (3, "narrow"), (8, "narrow"), (2, "wide"), (3, "wide")])
def test_mirror_bc_line_of_slabs_matches_even_extension(world, halo, tmp_path, orc):
    """PF_BC_MIRROR in slab mode ('a line with local reflection at the ends'): SlabSolver skips the wall
    neighbours (-1), the engine mirrors its own planes; result = the whole-domain oracle on the 3-D even extension."""
    out = str(tmp_path / "res.npz")
    nsteps = 4 if halo == "narrow" else 5
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out,
                                       str(nsteps), "mirror", halo], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = np.load(out)
    nz, ny, nx = res["full"].shape
    e = orc.even_extend(res["full"])
    F0, C0, _ = orc.diagnostics(e, h=1.0, mirror=True)
    np.testing.assert_allclose(res["d0"][:2], [F0, C0], rtol=1e-13)
    for _ in range(nsteps):
        e = orc.fd_step(e, 1e-3)
    F1, C1, _ = orc.diagnostics(e, h=1.0, mirror=True)
    np.testing.assert_allclose(res["d1"][:2], [F1, C1], rtol=1e-13)
    e = orc.fd_step(e, 1e-3)
    np.testing.assert_array_equal(res["field"], e[:nz, :ny, :nx])


@pytest.mark.parametrize("mode,world", [("spectral", 2), ("bm6", 2), ("bm6_elim", 2), ("spectral", 8), ("bm6", 8), ("spectral_mirror", 2), ("spectral_mirror", 4), ("bm6_mirror", 2), ("bm6_mirror", 4)])
def test_fft_slab_solver_matches_single_domain(mode, world, tmp_path, orc):
    """the all-to-all / halo orchestration of FFTSlabSolver (world size 2, and 8 = the driver's node, gloo) against the
    single-domain oracles"""
    from oracle import bm6_fd, ch_spectral
    out = str(tmp_path / "res.npz")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_fft_worker.py"), out, mode],
                                      env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = np.load(out)
    dt = float(res["dt"])
    if mode.startswith("spectral"):
        o = ch_spectral.SpectralCH(res["full"], h=1.0)
        F0, C0 = o.diagnostics()
        o.step(dt, 3)
        F1, C1 = o.diagnostics()
        o.step(dt, 1)
        ref = o.c
    else:
        full = res["full"]
        mn = (full.shape[2] // 2 + 1, full.shape[1] // 2 + 1) if mode == "bm6_mirror" else None    # reference BCs (bench6.py:77-90)
        o = bm6_fd.BM6FD(full, 1.0, mirror_nodes=mn, eliminate_phi=mode == "bm6_elim")
        F0, C0, _ = o.diagnostics()
        o.step(dt, 3)
        F1, C1, _ = o.diagnostics()
        o.step(dt, 1)
        ref = o.c
    np.testing.assert_allclose(res["d0"][:2], [F0, C0], rtol=1e-11)
    np.testing.assert_allclose(res["d1"][:2], [F1, C1], rtol=1e-11)
    if mode in ("spectral_mirror", "bm6_mirror"):     # gather_field keeps the physical planes of the ring over the even extension
        nzp = ref.shape[0] // 2 + 1
        assert res["field"].shape[0] == nzp
        ref = ref[:nzp]
    assert np.abs(res["field"] - ref).max() <= 1e-12


@pytest.mark.parametrize("model,world", [("bm2", 2), ("bm2", 3), ("bm3", 2), ("bm3", 4)])
def test_multifield_slab_solver_matches_single_domain(model, world, tmp_path):
    """BM2 / BM3 explicit FD (dolfin/bench2.py:76-113, bench3.py:63-97 as explicit schemes) on a ring of slabs:
    pfhubbenchmarks_amd.solver.MultiFieldSlabSolver over gloo -- ghost refresh of every field (2 / 1 planes per side) before
    each step, uneven partitions -- reproduces oracle/multi_fd.py on the whole periodic box BIT for bit; the all-reduced
    diagnostics equal the global sums."""
    from oracle import multi_fd
    out = str(tmp_path / "res.npz")
    nsteps = 4
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_multi_worker.py"), out, model,
                                       str(nsteps)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = np.load(out)
    u = res["full"]
    np.testing.assert_allclose(res["d0"][:2], [u[0].sum(), (u[-1] * u[-1]).sum()], rtol=1e-13)
    dt, h, step = (2e-3, 1.3, multi_fd.bm2_step) if model == "bm2" else (5e-3, 0.9, multi_fd.bm3_step)
    for _ in range(nsteps):
        u = step(u, dt, h)
    np.testing.assert_array_equal(res["field"], u)
    np.testing.assert_allclose(res["d1"][:2], [u[0].sum(), (u[-1] * u[-1]).sum()], rtol=1e-13)


def test_eight_rank_partitions_of_the_baseline_configs():
    """pure-host geometry of the driver's 8-GPU runs (no GPU, no process group): BASELINE.json config 4 strong
    (1024^3 -> 128 planes per rank, 16 MiB ghost messages), config 3 weak (512 planes per rank of a 512 x 512 x 4096
    box), config 5 / the spectral scheme strong (ONE 512^3 box -> 64 planes per rank: every axis within the hand-written
    passes' 128..1024 points), and the slab-FFT all-to-all block size -- what bench.py --gpus 8 will allocate per rank."""
    import ctypes as C
    from pfhubbenchmarks_amd import lib as L
    lib = L.load()
    for (nx, ny, nzg, world) in ((1024, 1024, 1024, 8), (512, 512, 4096, 8), (512, 512, 1024, 2), (512, 512, 2048, 4),
                                 (512, 512, 512, 8), (512, 512, 512, 4), (512, 512, 512, 2)):
        planes = []
        for r in range(world):
            f, c = C.c_int(), C.c_int()
            assert lib.pf_slab_partition(nzg, world, r, C.byref(f), C.byref(c)) == 0
            planes.append((f.value, c.value))
            cfg = L.default_config(3, nx, 1.0)
            cfg.n[0], cfg.n[1], cfg.n[2] = nx, ny, nzg
            cfg.nranks, cfg.rank = world, r
            assert lib.pf_field_elems(C.byref(cfg)) == nx * ny * c.value
            assert lib.pf_field_elems_with_ghosts(C.byref(cfg)) == nx * ny * (c.value + 4)
            off1 = lib.pf_ext_buffer_offset(C.byref(cfg), 1)
            assert off1 >= nx * ny * (c.value + 4) and (off1 * 8) % (512 * 1024) == 64 * 1024
            cfg.scheme = L.PF_SCHEME_SPECTRAL_SI
            per_rank = lib.pf_a2a_buffer_doubles(C.byref(cfg))
            # complex half spectrum of the local planes; power-of-two boxes (every axis 128..1024) use the hand-written
            # passes and their 128-byte-aligned row pitch, anything else rocFFT's natural nx/2 + 1
            fast = all(128 <= m <= 1024 and m & (m - 1) == 0 for m in (nx, ny, nzg))
            pitch = (nx // 2 + 1 + 7) // 8 * 8 if fast else nx // 2 + 1
            assert per_rank == 2 * pitch * ny * c.value, (nx, ny, nzg, per_rank)
        assert planes[0][0] == 0 and all(planes[i][0] + planes[i][1] == planes[i + 1][0] for i in range(world - 1))
        assert planes[-1][0] + planes[-1][1] == nzg and len({c for _, c in planes}) == 1
    # 2-D problems do not shard (replicas only)
    cfg = L.default_config(2, 512, 1.0)
    cfg.nranks, cfg.rank = 8, 0
    assert lib.pf_field_elems_with_ghosts(C.byref(cfg)) < 0
