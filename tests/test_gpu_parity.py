"""Parity of the HIP path (through the C ABI of libpfhip.so) with the CPU oracle -- the -m gpu suite.

Bit-exact where the arithmetic is defined operation by operation (the FD step: oracle/ch_fd.c states the same
fma/add order as csrc/ch_fd_kernels.hip); 1e-13 relative for reductions (different summation order); a few ulp for
the initial condition (device cos vs libm cos)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from pfhubbenchmarks_amd import lib as L  # noqa: E402
from pfhubbenchmarks_amd.solver import HipSlabEngine, PhaseFieldSolver  # noqa: E402

FUSED_VARIANTS = [0, 1, 2, 3, 4, 5, 6, 9, 11, 13, 14, 16, 17]


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "the gpu suite needs a GPU"
    return L.load()


def gpu_fd_step(lib, c, dt, impl, h=1.0, ghost=0, zwrap=1, zlo=None, zhi=None, out=None):
    c3 = c[None] if c.ndim == 2 else c
    nzb, ny, nx = c3.shape
    nz = nzb - 2 * ghost
    zlo = 0 if zlo is None else zlo
    zhi = nz if zhi is None else zhi
    t_in = torch.from_numpy(np.ascontiguousarray(c3)).cuda()
    t_out = torch.zeros_like(t_in) if out is None else torch.from_numpy(out).cuda()
    p = L.PfkChParams(0.3, 0.7, 10.0, 2.0 / (h * h), dt * 5.0 / (h * h), 0.0)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.pfk_ch_fd_step(t_in.data_ptr(), t_out.data_ptr(), None, nx, ny, nz, ghost, zwrap, zlo, zhi,
                            C.byref(p), impl, C.c_void_p(stream))
    assert rc == 0, lib.pf_last_error(None)
    torch.cuda.synchronize()
    o = t_out.cpu().numpy()
    return o[0] if c.ndim == 2 else o


SHAPES = [  # (nx, ny, nz)
    (128, 16, 8),    # exactly one full tile
    (256, 32, 12),   # 2 x 2 full tiles
    (64, 16, 4),     # partial in x
    (130, 17, 5),    # partial in x and y, second tile 2 wide / 1 high
    (2, 1, 1), (4, 3, 2), (6, 2, 3),   # degenerate periodic wraps
    (384, 40, 9),
    (256, 48, 1),    # 2-D
    (258, 20, 2),
    (1024, 1024, 1),  # BASELINE.json config 4's plane (8 MiB: the largest 32-bit in-plane offsets), 2-D
    (1024, 1024, 3),  # ... and with z-neighbours
]


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_step_bit_exact_all_variants(lib, orc, shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(nx * 7 + ny * 3 + nz)
    c = 0.5 + 0.2 * rng.standard_normal((nz, ny, nx))
    ref = orc.fd_step(c, 1e-3)
    two = gpu_fd_step(lib, c, 1e-3, L.PF_KERNEL_TWOPASS)
    np.testing.assert_array_equal(two, ref)
    for v in FUSED_VARIANTS:
        assert lib.pfk_set_tuning(0, v) == 0
        got = gpu_fd_step(lib, c, 1e-3, L.PF_KERNEL_FUSED)
        np.testing.assert_array_equal(got, ref, err_msg="variant %d shape %s" % (v, shape))
    lib.pfk_set_tuning(0, 0)


def test_fused_multi_chunk_and_many_steps(lib, orc):
    rng = np.random.default_rng(1)
    c = 0.5 + 0.05 * rng.standard_normal((160, 48, 256))      # 160 planes -> several z-chunks
    ref, got = c, c
    for _ in range(3):
        ref = orc.fd_step(ref, 1e-3)
        got = gpu_fd_step(lib, got, 1e-3, L.PF_KERNEL_FUSED)
    np.testing.assert_array_equal(got, ref)


def test_odd_nx_needs_twopass(lib, orc):
    with pytest.warns(RuntimeWarning, match="nx is odd"):          # the 6x cliff is announced, not silent
        with PhaseFieldSolver(dim=3, n=(7, 6, 5), h=1.0) as s_odd:
            assert "two-pass" in s_odd.status
    with PhaseFieldSolver(dim=3, n=(8, 6, 5), h=1.0) as s_even:
        assert s_even.status.startswith("fd: fused 2.5-D")
    rng = np.random.default_rng(2)
    c = 0.5 + 0.1 * rng.standard_normal((3, 5, 7))
    np.testing.assert_array_equal(gpu_fd_step(lib, c, 1e-3, L.PF_KERNEL_AUTO), orc.fd_step(c, 1e-3))
    t = torch.zeros(3 * 5 * 7, dtype=torch.float64, device="cuda")
    p = L.PfkChParams(0.3, 0.7, 10.0, 2.0, 5e-3, 0.0)
    rc = lib.pfk_ch_fd_step(t.data_ptr(), t.data_ptr(), None, 7, 5, 3, 0, 1, 0, 3, C.byref(p), L.PF_KERNEL_FUSED, None)
    assert rc == L.PF_ERR_UNSUPPORTED


@pytest.mark.parametrize("variant", [0, 3, 6])
def test_slab_ghost_mode_and_plane_ranges(lib, orc, variant):
    """zwrap = 0: planes -2..nz+1 come from ghost planes; interior / boundary launches as in pf_step_begin/finish."""
    lib.pfk_set_tuning(0, variant)
    rng = np.random.default_rng(5)
    nzl = 10
    slab = 0.5 + 0.1 * rng.standard_normal((nzl + 4, 24, 128))
    ref = orc.fd_step(slab, 1e-3, ghost=2, zwrap=0)
    got = gpu_fd_step(lib, slab, 1e-3, L.PF_KERNEL_FUSED, ghost=2, zwrap=0)
    np.testing.assert_array_equal(got[2:2 + nzl], ref[2:2 + nzl])
    acc = np.zeros_like(slab)
    for zlo, zhi in [(2, nzl - 2), (0, 2), (nzl - 2, nzl)]:
        acc = gpu_fd_step(lib, slab, 1e-3, L.PF_KERNEL_FUSED, ghost=2, zwrap=0, zlo=zlo, zhi=zhi, out=acc)
    np.testing.assert_array_equal(acc[2:2 + nzl], ref[2:2 + nzl])
    assert not acc[:2].any() and not acc[-2:].any()            # ghost planes of the output are never written
    lib.pfk_set_tuning(0, 0)


def test_handle_periodic_2d_trajectory_and_diagnostics(lib, orc):
    n = 192
    c0 = orc.ic(n, n, 1)[0]
    with PhaseFieldSolver(dim=2, n=n, h=1.0) as s:
        s.set_ic_bm1(0.5, 0.05)
        got0 = s.get_c()
        assert np.abs(got0 - c0).max() < 1e-14                 # device cos vs libm cos
        s.set_c(c0)                                            # identical start -> bit-exact evolution
        np.testing.assert_array_equal(s.get_c(), c0)
        ref = c0
        for k in range(1, 21):
            s.step(1e-3)
            ref = orc.fd_step(ref, 1e-3)
            if k in (1, 7, 20):
                np.testing.assert_array_equal(s.get_c(), ref)
        F, Ctot, E = s.diagnostics()
        Fo, Co, _ = orc.diagnostics(ref)
        assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(Ctot - Co) <= 1e-13 * abs(Co) and E == 0.0
        # rollback restores the state before the last step (bench1.py:171-173 semantics)
        before = s.get_c()
        ok, cmin, cmax = s.step(1e-3, check=True)
        assert ok and 0.0 < cmin < cmax < 1.0
        s.rollback()
        np.testing.assert_array_equal(s.get_c(), before)
        with pytest.raises(L.PfhipError):
            s.rollback()
        # blow-up guard: an unstable dt must be reported, not silently accepted
        ok, cmin, cmax = s.step(1.0, nsteps=30, check=True)
        assert not ok


def test_handle_mirror_bc_is_the_even_extension(lib, orc):
    from oracle import fem_be
    n = 41
    x = np.arange(n) * 2.0
    c = fem_be.ic_bm1(x[None, :], x[:, None])
    with PhaseFieldSolver(dim=2, n=n, h=2.0, bc="mirror") as s:
        s.set_ic_bm1()
        assert np.abs(s.get_c() - c).max() < 1e-14
        s.set_c(c)
        e = orc.even_extend(c)
        for _ in range(25):
            e = orc.fd_step(e, 0.02, h=2.0)
        s.step(0.02, 25)
        np.testing.assert_array_equal(s.get_c(), e[:n, :n])
        F, Ctot, _ = s.diagnostics()
        Fo, Co, _ = orc.diagnostics(e, h=2.0, mirror=True)
        assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(Ctot - Co) <= 1e-13 * abs(Co)


def test_handle_3d(lib, orc):
    n = (128, 32, 20)
    rng = np.random.default_rng(11)
    c = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    for kern in ("fused", "twopass"):
        with PhaseFieldSolver(dim=3, n=n, h=1.0, kernel=kern) as s:
            s.set_c(c)
            s.step(5e-4, 4)
            ref = c
            for _ in range(4):
                ref = orc.fd_step(ref, 5e-4)
            np.testing.assert_array_equal(s.get_c(), ref)
            F, Ctot, _ = s.diagnostics()
            Fo, Co, _ = orc.diagnostics(ref)
            assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(Ctot - Co) <= 1e-13 * abs(Co)


def test_slab_engines_with_manual_exchange_equal_single_domain(lib, orc):
    """Two slab-mode handles on the one GPU, ghost planes copied by hand where SlabSolver would use RCCL:
    checks pf_step_begin / pf_step_finish / pf_halo_layout against the whole-domain oracle."""
    n = (128, 24, 16)
    rng = np.random.default_rng(13)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    engs = [HipSlabEngine(n, 1.0, 2, r, 0) for r in range(2)]
    for e in engs:
        e.set_local(full[e.z0:e.z0 + e.nz])

    def exchange():
        torch.cuda.synchronize()
        for e, o in ((engs[0], engs[1]), (engs[1], engs[0])):
            be, bo = e.buffers[e.cur], o.buffers[o.cur]
            be[0:2].copy_(bo[o.nz:o.nz + 2])               # my lo ghosts <- neighbour's top planes
            be[e.nz + 2:e.nz + 4].copy_(bo[2:4])           # my hi ghosts <- neighbour's bottom planes
        torch.cuda.synchronize()

    ref = full
    for _ in range(3):
        exchange()
        for e in engs:
            e.step_begin(1e-3)
        for e in engs:
            e.step_finish()
        ref = orc.fd_step(ref, 1e-3)
    got = np.concatenate([e.get_local() for e in engs], 0)
    np.testing.assert_array_equal(got, ref)
    exchange()
    d = np.sum([e.diag_local() for e in engs], 0)
    Fo, Co, _ = orc.diagnostics(ref)
    assert abs(d[0] - Fo) <= 1e-13 * abs(Fo) and abs(d[1] - Co) <= 1e-13 * abs(Co)
    for e in engs:
        e.close()


@pytest.mark.parametrize("bc", ["periodic", "mirror"])
def test_wide_halo_slab_engines_equal_single_domain(lib, orc, bc):
    """PF_FLAG_WIDE_HALO: 4 ghost planes exchanged every SECOND step (step A: interior + two 4-plane strips, outputs
    [-2, nz+2); step B: one launch, no exchange).  Three slab handles on the one GPU, ghost planes copied by hand only when
    the library asks for them (pf_halo_layout.needs_exchange); 7 steps (both step kinds, odd count); bit-identical to the
    whole-domain oracle (periodic ring) / to the even extension (mirror line), and to the 2-ghost path by transitivity."""
    n = (128, 24, 18) if bc == "periodic" else (65, 13, 17)
    rng = np.random.default_rng(31)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    engs = [HipSlabEngine(n, 1.0, 3, r, 0, bc=bc, wide=True) for r in range(3)]
    assert all(e.ghost == 4 and e.buffers[0].shape[0] == e.nz + 8 for e in engs)
    for e in engs:
        e.set_local(full[e.z0:e.z0 + e.nz])

    def exchange():
        torch.cuda.synchronize()
        pairs = ((0, 1), (1, 2), (2, 0)) if bc == "periodic" else ((0, 1), (1, 2))
        for lo, hi in pairs:                               # `hi` sits above `lo`
            el, eh = engs[lo], engs[hi]
            bl, bh = el.buffers[el.cur], eh.buffers[eh.cur]
            bl[el.nz + 4:el.nz + 8].copy_(bh[4:8])         # lo's upper ghosts <- hi's bottom 4 planes
            bh[0:4].copy_(bl[el.nz:el.nz + 4])             # hi's lower ghosts <- lo's top 4 planes
        torch.cuda.synchronize()

    ref = orc.even_extend(full) if bc == "mirror" else full
    exchanges = 0
    for k in range(7):
        need = [e.needs_exchange() for e in engs]
        assert len(set(need)) == 1 and need[0] == (k % 2 == 0)
        if need[0]:
            exchange()
            exchanges += 1
        for e in engs:
            e.step_begin(1e-3)
        for e in engs:
            e.step_finish()
        ref = orc.fd_step(ref, 1e-3)
    assert exchanges == 4
    got = np.concatenate([e.get_local() for e in engs], 0)
    np.testing.assert_array_equal(got, ref[:n[2], :n[1], :n[0]] if bc == "mirror" else ref)
    exchange()
    d = np.sum([e.diag_local() for e in engs], 0)
    Fo, Co, _ = orc.diagnostics(ref, h=1.0, mirror=bc == "mirror")
    assert abs(d[0] - Fo) <= 1e-13 * abs(Fo) and abs(d[1] - Co) <= 1e-13 * abs(Co)
    for e in engs:
        e.close()
    with pytest.raises(Exception):                         # too few planes per rank for 4 ghost planes
        HipSlabEngine((128, 24, 9), 1.0, 3, 0, 0, wide=True)


def test_mirror_bc_line_of_slabs_equals_even_extension(lib, orc):
    """PF_BC_MIRROR in slab mode (b13d.py's no-flux box across GPUs): three slab handles on the one GPU in a LINE,
    inner ghost planes copied by hand, wall ghosts mirrored by the library; the result is bit for bit the whole-domain
    run on the 3-D even extension (single-GPU mirror handle and the oracle), diagnostics use trapezoid weights in z."""
    n = (65, 13, 14)                                       # nodes (x, y, z): lattice planes are 128 x 24
    rng = np.random.default_rng(29)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    engs = [HipSlabEngine(n, 1.0, 3, r, 0, bc="mirror") for r in range(3)]
    assert [(e.rank_lo, e.rank_hi) for e in engs] == [(-1, 1), (0, 2), (1, -1)]
    assert [e.nz for e in engs] == [5, 5, 4]
    for e in engs:
        e.set_local(full[e.z0:e.z0 + e.nz])

    def exchange():
        torch.cuda.synchronize()
        for lo, hi in ((engs[0], engs[1]), (engs[1], engs[2])):
            bl, bh = lo.buffers[lo.cur], hi.buffers[hi.cur]
            bl[lo.nz + 2:lo.nz + 4].copy_(bh[2:4])         # lower slab's hi ghosts <- upper slab's bottom planes
            bh[0:2].copy_(bl[lo.nz:lo.nz + 2])             # upper slab's lo ghosts <- lower slab's top planes
        torch.cuda.synchronize()

    ext = orc.even_extend(full)
    with PhaseFieldSolver(dim=3, n=n, h=1.0, bc="mirror") as whole:
        whole.set_c(full)
        for _ in range(3):
            exchange()
            for e in engs:
                e.step_begin(1e-3)
            for e in engs:
                e.step_finish()
            ext = orc.fd_step(ext, 1e-3)
        whole.step(1e-3, 3)
        got = np.concatenate([e.get_local() for e in engs], 0)
        np.testing.assert_array_equal(got, ext[:n[2], :n[1], :n[0]])
        np.testing.assert_array_equal(got, whole.get_c())
        exchange()
        d = np.sum([e.diag_local() for e in engs], 0)
        Fo, Co, _ = orc.diagnostics(ext, h=1.0, mirror=True)
        Fw, Cw, _ = whole.diagnostics()
        assert abs(d[0] - Fo) <= 1e-13 * abs(Fo) and abs(d[1] - Co) <= 1e-13 * abs(Co)
        assert abs(d[0] - Fw) <= 1e-13 * abs(Fw) and abs(d[1] - Cw) <= 1e-13 * abs(Cw)
    for e in engs:
        e.close()
    # too few planes per rank for the reflection: refused at create time, not a fault later
    with pytest.raises(Exception):
        HipSlabEngine((65, 13, 5), 1.0, 2, 0, 0, bc="mirror")


def test_full_size_properties_512cubed(lib):
    """BASELINE.json config 3 (512^3): size-independent properties instead of a full CPU comparison --
    z-invariance of the extruded problem (b13d.py:55), bitwise equality with the 2-D run, mass conservation."""
    n = 512
    with PhaseFieldSolver(dim=3, n=n, h=1.0) as s3, PhaseFieldSolver(dim=2, n=n, h=1.0) as s2:
        s3.set_ic_bm1()
        s2.set_ic_bm1()
        F0, C0, _ = s3.diagnostics()
        s3.step(1e-3, 6)
        s2.step(1e-3, 6)
        F1, C1, _ = s3.diagnostics()
        assert abs(C1 - C0) <= 1e-13 * abs(C0)
        assert F1 < F0
        f3 = s3.get_c()
        f2 = s2.get_c()
        for z in (0, 1, 255, 511):
            np.testing.assert_array_equal(f3[z], f2)
        F2, C2, _ = s2.diagnostics()
        assert abs(F1 - n * F2) <= 1e-12 * abs(F1)             # F_3D = L_z F_2D


def test_full_size_properties_1024cubed(lib):
    """BASELINE.json config 4 (1024^3, north_star's roofline target; 16 GiB of state on ONE GPU): the same
    size-independent properties as at 512^3 -- z-invariance of the extruded problem (b13d.py:24-26,55), bitwise equality
    with the 1024^2 run (itself bit-compared with the oracle above), mass conservation, F_3D = L_z F_2D."""
    n = 1024
    with PhaseFieldSolver(dim=3, n=n, h=1.0) as s3, PhaseFieldSolver(dim=2, n=n, h=1.0) as s2:
        s3.set_ic_bm1()
        s2.set_ic_bm1()
        F0, C0, _ = s3.diagnostics()
        ok, cmin, cmax = s3.step(5e-4, 4, check=True)
        assert ok and 0.3 < cmin < cmax < 0.7
        s2.step(5e-4, 4)
        F1, C1, _ = s3.diagnostics()
        assert abs(C1 - C0) <= 1e-13 * abs(C0)
        assert F1 < F0
        f2 = s2.get_c()
        F2, C2, _ = s2.diagnostics()
        assert abs(F1 - n * F2) <= 1e-12 * abs(F1) and abs(C1 - n * C2) <= 1e-12 * abs(C1)
        f3 = s3.get_c()                                          # 8 GiB on the host
        assert f3.shape == (n, n, n)
        for z in (0, 1, 2, 127, 128, 511, 512, 1022, 1023):      # first / last planes of z-chunks and of the box
            np.testing.assert_array_equal(f3[z], f2)
        # every plane, cheaply: all planes equal plane 0 (z-invariance), checked on 1/8 of the columns
        assert bool((f3[:, ::8, ::8] == f2[None, ::8, ::8]).all())
        del f3


@pytest.mark.parametrize("shape", [(512, 512), (128, 256), (1024, 128), (256, 512), (512, 128), (96, 40), (34, 18, 10),
                                   (64, 64, 64), (128, 128, 128), (256, 128, 512), (128, 512, 256), (1024, 128, 128),
                                   (128, 128, 1024), (128, 1024, 128), (200, 200), (400, 400), (40, 96, 200), (9, 12, 30),
                                   (1000, 8), (250, 1024, 16), (512, 128, 128), (512, 256, 128)])
def test_spectral_scheme_matches_numpy_oracle(lib, shape, monkeypatch):
    """BASELINE.json config 2 (512^2 semi-implicit spectral) vs the pocketfft oracle, 1e-11 relative: 2-D power-of-two
    shapes take the fused LDS-FFT path (csrc/spectral2d_fused.hip: one-wave radix-8 kernels on 512-point axes -- (256, 512)
    and (512, 128) pair one of them with the generic radix-2^2 kernel of the other axis); 3-D boxes whose axes are all
    powers of two in 128..1024 take the hand-written 4-pass path too (every axis length on every axis: 128, 256, 512 and
    1024 along x, y and z); axes that factor into 2, 3 and 5 (8..1024 points: the reference's 200- and 400-point lattices,
    40 x 96, 64^3, 200 x 96 x 40, 30 x 12 x 9, 8 x 1000, 16 x 1024 x 250) the mixed-radix Stockham kernels (mx_row_kernel /
    mx_col_kernel); only (10, 18, 34), with its factor 17, goes to the library (rocFFT through its native API)."""
    from oracle import ch_spectral
    dim = len(shape)
    n = shape[::-1]                      # (nx, ny[, nz])
    monkeypatch.setenv("PFHIP_SPECTRAL_3D", "lds")   # 128^3 would default to rocFFT (faster there); no effect on other shapes
    rng = np.random.default_rng(sum(shape))
    if shape == (512, 512):
        from oracle import ch_fd
        c = ch_fd.ic(512, 512, 1)[0]
    else:
        c = 0.5 + 0.05 * rng.standard_normal(shape)
    sp = ch_spectral.SpectralCH(c, h=1.0)
    with PhaseFieldSolver(dim=dim, n=n, h=1.0, scheme="spectral") as s:
        s.set_c(c)
        F, Ctot, _ = s.diagnostics()
        Fo, Co = sp.diagnostics()
        assert abs(F - Fo) <= 1e-11 * abs(Fo) and abs(Ctot - Co) <= 1e-13 * abs(Co)
        for k in (1, 9):
            s.step(1e-2, k)
            sp.step(1e-2, k)
            got = s.get_c()
            assert np.abs(got - sp.c).max() <= 1e-11 * np.abs(sp.c).max()
        F, Ctot, _ = s.diagnostics()
        Fo, Co = sp.diagnostics()
        assert abs(F - Fo) <= 1e-10 * abs(Fo) and abs(Ctot - Co) <= 1e-12 * abs(Co)
        # rollback drops the resident spectrum and restores the previous real-space field
        s.step(1e-2)
        s.rollback()
        assert np.abs(s.get_c() - sp.c).max() <= 1e-11
        s.step(1e-2)
        sp.step(1e-2)
        assert np.abs(s.get_c() - sp.c).max() <= 1e-11


@pytest.mark.parametrize("shape", [(512, 512), (128, 256), (96, 40), (34, 18, 10), (128, 128, 128), (256, 128, 512),
                                   (200, 400), (20, 24, 50), (512, 128, 128)])
def test_bm6_spectral_scheme_matches_numpy_oracle(lib, shape, monkeypatch):
    """BM6 with the semi-implicit spectral scheme in a periodic box: phi eliminated in Fourier space (one more implicit
    term in the k-space update, every kernel form: radix-8 LDS FFT, radix-2^2 LDS FFT, rocFFT 2-D and 3-D), f_elec by
    Parseval; against oracle/ch_spectral.py (1e-11); and consistent with the FD scheme's phi-eliminated step in the
    small-dt limit is left to the convergence study."""
    from oracle import ch_spectral
    dim = len(shape)
    monkeypatch.setenv("PFHIP_SPECTRAL_3D", "lds")
    rng = np.random.default_rng(sum(shape) + 1)
    c = 0.5 + 0.04 * rng.standard_normal(shape)
    sp = ch_spectral.SpectralCH(c, h=1.0, bm6=True)
    with PhaseFieldSolver(dim=dim, n=shape[::-1], h=1.0, scheme="spectral", model="bm6") as s:
        s.set_c(c)
        F, Ctot, E = s.diagnostics()
        Fo, Co, Eo = sp.diagnostics()
        assert abs(F - Fo) <= 1e-11 * abs(Fo) and abs(Ctot - Co) <= 1e-13 * abs(Co) and abs(E - Eo) <= 1e-10 * abs(Eo)
        assert Eo > 0.0
        for k in (1, 9):
            s.step(1e-2, k)
            sp.step(1e-2, k)
            assert np.abs(s.get_c() - sp.c).max() <= 1e-11 * np.abs(sp.c).max()
        F, Ctot, E = s.diagnostics()
        Fo, Co, Eo = sp.diagnostics()
        assert abs(F - Fo) <= 1e-10 * abs(Fo) and abs(Ctot - Co) <= 1e-12 * abs(Co) and abs(E - Eo) <= 1e-9 * abs(Eo)
    with pytest.raises(Exception):          # reference boundary conditions need the FD scheme's Dirichlet Poisson solve
        PhaseFieldSolver(dim=2, n=65, h=1.0, bc="mirror", scheme="spectral", model="bm6")


@pytest.mark.gpu
def test_allocation_policies_give_identical_fields_and_say_what_they_are(lib, monkeypatch):
    """csrc/device_alloc.hip: arrays of 32 MiB and more are physically contiguous allocations by default (reproducible step
    times); PFHIP_ALLOC=plain / scatter[:KiB] are the other measured policies.  Placement must never change a result: the
    fields of a 256^3 spectral run and of a 3-field BM3 run are BIT-identical under all of them, pf_status_string names
    the policy, handles can be created and destroyed repeatedly (the scattered kind unmaps its pieces), and
    pf_device_malloc / pf_device_free hand the same memory to callers with buffers of their own."""
    import ctypes as C
    rng = np.random.default_rng(11)
    c0 = 0.5 + 0.05 * rng.standard_normal((256, 256, 256))
    want = {"": "physically contiguous", "plain": "plain hipMalloc", "scatter": "scattered in 2048 KiB", "scatter:4096": "scattered in 4096 KiB"}
    out = {}
    for pol, clause in want.items():
        if pol:
            monkeypatch.setenv("PFHIP_ALLOC", pol)
        else:
            monkeypatch.delenv("PFHIP_ALLOC", raising=False)
        for rep in range(2):
            with PhaseFieldSolver(dim=3, n=256, h=1.0, scheme="spectral", model="bm1") as s:
                assert clause in s.status and "WARNING" not in s.status, s.status
                s.set_c(c0)
                s.step(1e-2, 3)
                out[pol] = s.get_c()
        # memory for the caller's own buffers: usable as the field pair of a handle (ext_c), 256-byte aligned
        elems = 256 ** 3
        ptr = C.c_void_p()
        assert lib.pf_device_malloc(C.byref(ptr), C.c_size_t(2 * 8 * elems)) == 0 and ptr.value and ptr.value % 256 == 0
        with PhaseFieldSolver(dim=3, n=256, h=1.0, scheme="fd", model="bm1", ext_c=(ptr.value, ptr.value + 8 * elems)) as s:
            s.set_c(c0)
            s.step(1e-3, 2)
            out[pol + "/fd"] = s.get_c()
        assert lib.pf_device_free(ptr) == 0
    for pol in want:
        np.testing.assert_array_equal(out[""], out[pol], err_msg="PFHIP_ALLOC=%s" % pol)
        np.testing.assert_array_equal(out["/fd"], out[pol + "/fd"], err_msg="PFHIP_ALLOC=%s (caller's buffers)" % pol)
    assert lib.pf_device_malloc(None, C.c_size_t(16)) < 0 and lib.pf_device_free(None) == 0


@pytest.mark.parametrize("shape,model", [((128, 128, 256), "bm1"), ((40, 96, 200), "bm1"), ((128, 256, 128), "bm6"),
                                         ((64, 128, 128), "bm6fd"), ((512, 128, 128), "bm1")])
def test_spectral_plane_local_passes_in_chunks_are_bit_identical(lib, monkeypatch, shape, model):
    """The passes of a 3-D spectral step / Poisson solve that only couple points of one z-plane (x rows, y columns) run chunk
    of planes by chunk of planes, the chunks dealt to side streams (run_chunked in csrc/spectral2d_fused.hip; default on
    boxes above 96 MiB of half spectrum, e.g. 512^3: 32 planes, 2 streams).  Same kernels, same arithmetic: the fields of
    PFHIP_FFT3D_CHUNK="0" (whole box per launch) and of several chunkings -- uneven last chunk, 1 to 4 streams -- must be
    BIT-identical, on the power-of-two and the mixed-radix kernels, for the spectral schemes (BM1, BM6) and for the FD
    scheme's periodic Poisson solve; pf_status_string names the chunking."""
    monkeypatch.setenv("PFHIP_SPECTRAL_3D", "lds")
    n = shape[::-1]
    rng = np.random.default_rng(7)
    c0 = 0.5 + 0.05 * rng.standard_normal(shape)
    scheme = "fd" if model == "bm6fd" else "spectral"
    dt = 5e-4 if model == "bm6fd" else 1e-2
    out = {}
    for chunk in ("0", "16,2", "24,3", "8,1", "5,4"):
        monkeypatch.setenv("PFHIP_FFT3D_CHUNK", chunk)
        with PhaseFieldSolver(dim=3, n=n, h=1.0, scheme=scheme, model=model[:3]) as s:
            s.set_c(c0)
            s.step(dt, 1)
            a = s.get_c()
            s.step(dt, 4)
            out[chunk] = (a, s.get_c(), np.array(s.diagnostics()), s.get_phi() if model == "bm6fd" else None)
            if chunk == "0":
                assert "whole box per launch" in s.status, s.status
            else:       # the lanes are the handle's stream + side streams TESTED to run beside it: at most what was asked for
                import re
                m = re.search(r"chunks of (\d+) planes on (\d+) stream", s.status)
                assert m and m.group(1) == chunk.split(",")[0] and 1 <= int(m.group(2)) <= int(chunk.split(",")[1]), s.status
    for chunk, got in out.items():
        for x, y in zip(out["0"], got):
            if x is not None:
                np.testing.assert_array_equal(x, y, err_msg="PFHIP_FFT3D_CHUNK=%s" % chunk)



def test_spectral_512cubed_one_step_against_the_cpu_oracle(lib):
    """512^3 (the size the hand-written passes exist for) against the CPU oracle itself, not against another path of the
    same library: one semi-implicit step of oracle/ch_spectral.py (pocketfft through scipy.fft, threaded: three 1 GiB
    transforms) from the BM1 initial condition plus noise.  Field to 1e-11 (pocketfft and the radix-8 LDS FFTs round
    differently), F and C to 1e-11 / 1e-13."""
    import os
    from oracle import ch_fd, ch_spectral
    rng = np.random.default_rng(79)
    c0 = np.repeat(ch_fd.ic(512, 512, 1), 512, 0) + 0.01 * rng.standard_normal((512, 512, 512), dtype=np.float32)
    o = ch_spectral.SpectralCH(c0, h=1.0, workers=min(16, os.cpu_count() or 1))
    with PhaseFieldSolver(dim=3, n=512, h=1.0, scheme="spectral") as s:
        s.set_c(c0)
        F0, C0, _ = s.diagnostics()
        Fo, Co = o.diagnostics()
        assert abs(F0 - Fo) <= 1e-11 * abs(Fo) and abs(C0 - Co) <= 1e-13 * abs(Co)
        s.step(1e-2, 1)
        o.step(1e-2, 1)
        got = s.get_c()
        err = np.abs(got - o.c).max()
        assert err <= 1e-11, err
        F1, C1, _ = s.diagnostics()
        Fo, Co = o.diagnostics()
        assert abs(F1 - Fo) <= 1e-11 * abs(Fo) and abs(C1 - Co) <= 1e-13 * abs(Co)


def test_spectral_512cubed_lds_fft_passes_equal_the_rocfft_path(lib):
    """512^3 semi-implicit spectral step: the hand-written passes (x rows by f2_row512_kernel, y and z columns by
    f3_col512_kernel, 4 launches per step) against the rocFFT path of the same library (PFHIP_SPECTRAL_3D=rocfft, itself
    checked against the numpy oracle at small sizes above) on the same random field: fields to 1e-12, diagnostics to
    1e-12, mass conserved, rollback consistent."""
    import os
    rng = np.random.default_rng(77)
    c0 = 0.5 + 0.05 * rng.standard_normal((512, 512, 512))
    out = {}
    for name, env in (("lds", None), ("rocfft", "rocfft")):
        os.environ.pop("PFHIP_SPECTRAL_3D", None)
        if env:
            os.environ["PFHIP_SPECTRAL_3D"] = env
        try:
            with PhaseFieldSolver(dim=3, n=512, h=1.0, scheme="spectral") as s:
                s.set_c(c0)
                d0 = s.diagnostics()
                s.step(1e-2, 4)
                d1 = s.diagnostics()
                s.step(1e-2, 1)
                s.rollback()
                d2 = s.diagnostics()
                out[name] = (s.get_c(), d0, d1, d2)
        finally:
            os.environ.pop("PFHIP_SPECTRAL_3D", None)
    a, b = out["lds"], out["rocfft"]
    assert np.abs(a[0] - b[0]).max() <= 1e-12
    for k in (1, 2, 3):
        assert abs(a[k][0] - b[k][0]) <= 1e-12 * abs(b[k][0]) and abs(a[k][1] - b[k][1]) <= 1e-13 * abs(b[k][1])
    assert abs(a[2][1] - a[1][1]) <= 1e-12 * abs(a[1][1])          # mass conserved
    assert abs(a[3][0] - a[2][0]) <= 1e-12 * abs(a[2][0])          # rollback restores the state before the last step


def test_bm6_512cubed_poisson_lds_fft_passes_equal_the_rocfft_path(lib):
    """BM6 in the 512^3 periodic box: the Poisson solve by the hand-written passes (x, y, z with the division by the
    7-point Laplacian's eigenvalue, y, x) against the rocFFT solve of the same library (checked against the numpy
    oracle at small sizes in test_bm6_periodic_box_3d): phi to 1e-12, then two coupled steps."""
    import os
    rng = np.random.default_rng(78)
    c0 = 0.5 + 0.04 * rng.standard_normal((512, 512, 512))
    out = {}
    for name, env in (("lds", None), ("rocfft", "rocfft")):
        os.environ.pop("PFHIP_SPECTRAL_3D", None)
        if env:
            os.environ["PFHIP_SPECTRAL_3D"] = env
        try:
            with PhaseFieldSolver(dim=3, n=512, h=1.0, model="bm6") as s:
                s.set_c(c0)
                phi = s.get_phi()
                s.step(5e-4, 2)
                out[name] = (phi, s.get_c(), s.diagnostics())
        finally:
            os.environ.pop("PFHIP_SPECTRAL_3D", None)
    a, b = out["lds"], out["rocfft"]
    assert np.abs(b[0]).max() > 1e-5 and abs(b[0].mean()) < 1e-15
    assert np.abs(a[0] - b[0]).max() <= 1e-11 * np.abs(b[0]).max()
    assert np.abs(a[1] - b[1]).max() <= 1e-12
    assert abs(a[2][0] - b[2][0]) <= 1e-12 * abs(b[2][0]) and abs(a[2][2] - b[2][2]) <= 1e-10 * abs(b[2][2])


def test_bench1_driver_rows_against_fixture_and_oracle(lib, orc, golden_dir, tmp_path):
    """bench1 driver end to end on the GPU (first rows): CSV format identical to the reference's, rows at the
    fixture's times, values = the FD oracle's to 1e-12 and within the documented physical distance of the fixture."""
    import os
    from oracle import fem_be
    from pfhubbenchmarks_amd.drivers import run_bench1
    rows, _ = run_bench1(intervals=100, scheme="fd", dt=0.02, end_time=3.0, out_dir=str(tmp_path), verbose=False)
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    assert rows.shape == (5, 3)                       # t = 0.1 .. 3.1: the first row past end_time is still emitted
    np.testing.assert_allclose(rows[:, 0], csv[:5, 0], rtol=0, atol=1e-12)
    text = open(os.path.join(str(tmp_path), "bench1_out.csv")).read().splitlines()
    assert text[0] == "time,total_free_energy,total_solute" and len(text) == 6
    assert text[1].split(",")[0] == "0.1000000000"
    assert os.path.exists(os.path.join(str(tmp_path), "bench1", "stats.csv"))
    # oracle replay of the same sub-stepping
    x = np.arange(101) * 2.0
    e = orc.even_extend(fem_be.ic_bm1(x[None, :], x[:, None]))
    t = 0.0
    for i, tn in enumerate(csv[:5, 0]):
        n = int(np.floor((tn - t) / 0.02 + 1e-12))
        for _ in range(n):
            e = orc.fd_step(e, 0.02, h=2.0)
        t += n * 0.02
        if tn - t > 1e-14:
            e = orc.fd_step(e, tn - t, h=2.0)
        t = tn
        F, C, _ = orc.diagnostics(e, h=2.0, mirror=True)
        assert abs(rows[i, 1] - F) <= 1e-9 * abs(F)   # IC differs by ulps (device cos) -> not bitwise
        assert abs(rows[i, 2] - C) <= 1e-12 * abs(C)
    # physical distance to the reference's backward-Euler trajectory: grows with the reference's dt
    rel = np.abs(rows[:, 1] - csv[:5, 1]) / csv[:5, 1]
    assert rel[0] < 3e-3 and rel.max() < 3e-2


def test_bm6_reference_boundary_conditions(lib, golden_dir):
    """BM6 (bench6.py): Poisson solve with phi = 0 / sin(y/7) on x = 0 / Lx + coupled FD step vs the CPU restatement
    (oracle/bm6_fd.py); then the physical distance to the reference's first CSV row."""
    import os
    from oracle import bm6_fd, ch_fd, fem_be
    n, h = 101, 1.0
    x = np.arange(n) * h
    c = fem_be.ic_bm6(x[None, :], x[:, None])
    o = bm6_fd.BM6FD(ch_fd.even_extend(c), h, (n, n))
    with PhaseFieldSolver(dim=2, n=n, h=h, bc="mirror", model="bm6") as s:
        s.set_ic_bm6(0.5, 0.04)
        assert np.abs(s.get_c() - c).max() < 1e-14
        s.set_c(c)
        phi = s.get_phi()
        pref = o.phi()[:n, :n]
        assert np.abs(phi - pref).max() <= 1e-12
        assert np.abs(phi[:, 0]).max() == 0.0 and np.abs(phi[:, -1] - np.sin(x / 7.0)).max() < 1e-15
        F, C, E = s.diagnostics()
        Fo, Co, Eo = o.diagnostics()
        assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co) and abs(E - Eo) <= 1e-11 * abs(Eo)
        s.step(5e-4, 20)
        o.step(5e-4, 20)
        assert np.abs(s.get_c() - o.c[:n, :n]).max() <= 1e-12
        F, C, E = s.diagnostics()
        Fo, Co, Eo = o.diagnostics()
        assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)
        csv = np.loadtxt(os.path.join(golden_dir, "bench6_out.csv"), delimiter=",", skiprows=1)
        assert abs(F - csv[0, 1]) / csv[0, 1] < 1e-3      # t = 0.01: two discretisations of the same PDE
        assert abs(C - csv[0, 2]) / csv[0, 2] < 1e-4


@pytest.mark.parametrize("nodes", [(65, 65), (61, 49), (101, 51), (21, 17, 13), (33, 21, 9)])
def test_bm6_dirichlet_poisson_on_the_physical_nodes(lib, nodes, monkeypatch):
    """The reference's BM6 Poisson problem (phi = 0 / sin(y/7) on x = 0 / Lx, no flux elsewhere: dolfin/bench6.py:77-90,
    pfbase.py:410-421) by the hand-written sine (x) / cosine (y, z) passes on the PHYSICAL nodes (fused_poisson_dirichlet:
    dst_row_fwd_kernel, mx_col_kernel with the even extension in LDS, dst_row_inv_kernel) for lattices of 128, 120 / 96,
    200 / 100 points and two 3-D boxes: phi within 1e-12 of the numpy restatement (FFT of the odd / even extension on the
    whole lattice), exact boundary values, a few coupled steps -- and equal to the library route of round 2
    (PFHIP_POISSON_DIRICHLET=rocfft: rocFFT of the extension + three pointwise kernels), which stays the fallback for
    lattices with a prime factor above 5."""
    from oracle import bm6_fd, ch_fd
    dim = len(nodes)
    shape = nodes[::-1]                                  # numpy order (z, y, x)
    rng = np.random.default_rng(sum(nodes))
    c = 0.5 + 0.04 * rng.standard_normal(shape)
    ext = ch_fd.even_extend(c if dim == 3 else c[None])
    ext = ext if dim == 3 else ext[0]
    sl = tuple(slice(0, m) for m in shape)
    got = {}
    for route in ("lds", "rocfft"):
        monkeypatch.setenv("PFHIP_POISSON_DIRICHLET", route)
        o = bm6_fd.BM6FD(ext.copy(), 1.0, nodes[:2])
        with PhaseFieldSolver(dim=dim, n=nodes, h=1.0, bc="mirror", model="bm6") as s:
            s.set_c(c)
            phi = s.get_phi()
            assert np.abs(phi - o.phi()[sl]).max() <= 1e-12, route
            y = np.arange(nodes[1])
            assert np.abs(phi[..., 0]).max() == 0.0
            assert np.abs(phi[..., -1] - np.sin(y / 7.0)[(None,) * (dim - 2) + (slice(None),)]).max() < 1e-15
            s.step(5e-4, 6)
            o.step(5e-4, 6)
            assert np.abs(s.get_c() - o.c[sl]).max() <= 1e-12, route
            F, C, E = s.diagnostics()
            Fo, Co, Eo = o.diagnostics()
            assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(E - Eo) <= 1e-10 * abs(Eo)
            got[route] = (phi, s.get_c())
    assert np.abs(got["lds"][0] - got["rocfft"][0]).max() <= 1e-12
    assert np.abs(got["lds"][1] - got["rocfft"][1]).max() <= 1e-12


@pytest.mark.parametrize("shape", [(12, 20, 128), (128, 128, 128), (256, 128, 1024), (24, 40, 200), (100, 100), (384, 200)])
def test_bm6_periodic_box_3d(lib, shape, monkeypatch):
    """BM6 FD scheme in a periodic box; the Poisson solve runs on the hand-written passes: power-of-two boxes on
    fused3d_poisson's f2_row_kernel + f3_col_kernel MODE 4, boxes whose axes factor into 2, 3, 5 (128 x 20 x 12,
    200 x 40 x 24, and the 2-D 100 x 100 and 200 x 384) on the mixed-radix kernels (mx_row_kernel, mx_col_kernel MODE 4)."""
    from oracle import bm6_fd
    monkeypatch.setenv("PFHIP_SPECTRAL_3D", "lds")
    rng = np.random.default_rng(21)
    c = 0.5 + 0.04 * rng.standard_normal(shape)
    o = bm6_fd.BM6FD(c, 1.0)
    with PhaseFieldSolver(dim=len(shape), n=shape[::-1], h=1.0, model="bm6") as s:
        s.set_c(c)
        assert np.abs(s.get_phi() - o.phi()).max() <= 1e-12
        s.step(5e-4, 5)
        o.step(5e-4, 5)
        assert np.abs(s.get_c() - o.c).max() <= 1e-12
        F, C, E = s.diagnostics()
        Fo, Co, Eo = o.diagnostics()
        assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)


def test_bench6_driver(lib, golden_dir, tmp_path):
    import os
    from pfhubbenchmarks_amd.drivers import run_bench6
    rows, _ = run_bench6(intervals=100, end_time=0.1, out_dir=str(tmp_path), verbose=False)
    csv = np.loadtxt(os.path.join(golden_dir, "bench6_out.csv"), delimiter=",", skiprows=1)
    assert rows.shape == (4, 3)
    np.testing.assert_allclose(rows[:, 0], csv[:4, 0], atol=1e-12)
    rel = np.abs(rows[:, 1] - csv[:4, 1]) / csv[:4, 1]
    assert rel.max() < 1e-3
    assert open(os.path.join(str(tmp_path), "bench6_out.csv")).readline().strip() == "time,total_free_energy,total_solute"


def test_fem_be_parity_mode_bm1_against_reference_fixtures(lib, golden_dir):
    """PF_SCHEME_FEM_BE: the reference's own discretisation on the GPU.  Compared (a) row by row with the reference's
    committed results/bench1_out.csv and VTU c-fields, (b) with the pinned CPU oracle oracle/fem_be.py."""
    import os
    from oracle import fem_be
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    fields = np.load(os.path.join(golden_dir, "bm1_fields.npz"))
    o = fem_be.FemBE("bm1", newton_max=100)
    with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", max_newton=100) as s:
        s.set_ic_bm1(0.5, 0.05)
        assert s.get_c().shape == (20201,)
        assert np.abs(s.get_c() - o.c).max() < 1e-14
        F, C, _ = s.diagnostics()
        assert abs(F - 297.6736899201) < 1e-8 and abs(C - 20504.4690550850) < 1e-8      # known answers at t = 0
        tprev = 0.0
        for i in range(8):
            dt = csv[i, 0] - tprev
            ok, _, _ = s.step(dt, 1, check=True)
            its_o, ok_o = o.step(dt)
            assert ok and ok_o and s.last_iters == its_o
            tprev = csv[i, 0]
            F, C, _ = s.diagnostics()
            assert abs(F - csv[i, 1]) <= 1e-8 * csv[i, 1], (i, F, csv[i, 1])       # the reference's committed row
            assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2]
            Fo, Co = o.diagnostics()
            assert abs(F - Fo) <= 1e-11 * abs(Fo) and abs(C - Co) <= 1e-12 * abs(Co)   # the CPU oracle
            assert np.abs(s.get_c() - o.c).max() <= 1e-10
            assert np.abs(s.get_mu() - o.mu).max() <= 1e-9
            if i < 6:
                assert abs(fields["times"][i] - csv[i, 0]) < 1e-9
                assert np.abs(s.get_c() - fields["c"][i]).max() < 5e-9           # the reference's VTU snapshot
        before = s.get_c()
        s.step(1.6, 1, check=True)
        s.rollback()
        np.testing.assert_array_equal(s.get_c(), before)


def test_fem_be_parity_mode_bm6(lib, golden_dir):
    import os
    from oracle import fem_be
    csv = np.loadtxt(os.path.join(golden_dir, "bench6_out.csv"), delimiter=",", skiprows=1)
    o = fem_be.FemBE("bm6", newton_max=100)
    with PhaseFieldSolver(dim=2, n=101, h=1.0, bc="mirror", scheme="fem_be", model="bm6", max_newton=100) as s:
        s.set_ic_bm6(0.5, 0.04)
        tprev = 0.0
        for i in range(4):
            dt = csv[i, 0] - tprev
            ok, _, _ = s.step(dt, 1, check=True)
            its_o, ok_o = o.step(dt)
            assert ok and ok_o
            tprev = csv[i, 0]
            F, C, E = s.diagnostics()
            assert abs(F - csv[i, 1]) <= 1e-6 * csv[i, 1]          # reference rows (its two committed runs differ by 1e-7)
            assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2]
            Fo, Co = o.diagnostics()
            assert abs(F - Fo) <= 1e-10 * abs(Fo) and abs(C - Co) <= 1e-12 * abs(Co)
            assert np.abs(s.get_c() - o.c).max() <= 1e-9
            assert np.abs(s.get_phi() - o.phi).max() <= 1e-9


def test_fem_be_full_reference_trajectory_bm1(lib, golden_dir):
    """The WHOLE committed run of the reference (results/bench1_out.csv, 73 accepted steps, dt 0.1 ... 102.4) through the
    GPU BE-parity mode: every row's F within 1e-8 and C within 1e-9 of the reference's CSV, the six VTU c-fields within
    5e-9, and the plain-Newton iteration counts of the pinned CPU oracle (oracle/logs/fem_be_bm1_full.log) -- the large
    steps (rows 21, 37: dt = 12.8 / 51.2, 24 iterations; row 62: 16) are where the block cyclic reduction has to hold."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    fields = np.load(os.path.join(golden_dir, "bm1_fields.npz"))
    assert csv.shape == (73, 3)
    its, worst_f, worst_c = [], 0.0, 0.0
    with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", max_newton=100) as s:
        s.set_ic_bm1(0.5, 0.05)
        tprev = 0.0
        for i in range(73):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok, (i, s.last_iters)
            its.append(s.last_iters)
            tprev = csv[i, 0]
            F, C, _ = s.diagnostics()
            worst_f = max(worst_f, abs(F - csv[i, 1]) / csv[i, 1])
            worst_c = max(worst_c, abs(C - csv[i, 2]) / csv[i, 2])
            assert abs(F - csv[i, 1]) <= 1e-8 * csv[i, 1], (i, F, csv[i, 1])
            assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2], (i, C, csv[i, 2])
            if i < 6:
                assert np.abs(s.get_c() - fields["c"][i]).max() < 5e-9
    assert abs(tprev - 1031.9) < 1e-9
    assert (its[21], its[37], its[62]) == (24, 24, 16), its
    assert max(v for i, v in enumerate(its) if i not in (21, 37, 62)) <= 9, its
    print("BM1 full trajectory: worst rel F %.3g, worst rel C %.3g, Newton its %s" % (worst_f, worst_c, its))


def test_fem_be_full_reference_trajectory_bm6(lib, golden_dir):
    """All 14 rows of results/bench6_out.csv (dt 0.01 ... 0.32) through the monolithic (c, mu, phi) BE-parity mode: F within
    1e-6 (the reference's own two committed BM6 runs differ by 1e-7), C within 1e-9, the six VTU c / phi fields."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench6_out.csv"), delimiter=",", skiprows=1)
    fields = np.load(os.path.join(golden_dir, "bm6_fields.npz"))
    assert csv.shape == (14, 3)
    with PhaseFieldSolver(dim=2, n=101, h=1.0, bc="mirror", scheme="fem_be", model="bm6", max_newton=100) as s:
        s.set_ic_bm6(0.5, 0.04)
        tprev = 0.0
        for i in range(14):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok and s.last_iters <= 10, (i, s.last_iters)
            tprev = csv[i, 0]
            F, C, _ = s.diagnostics()
            assert abs(F - csv[i, 1]) <= 1e-6 * csv[i, 1], (i, F, csv[i, 1])
            assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2]
            if i < 6:
                assert abs(fields["times"][i] - csv[i, 0]) < 1e-9
                # the field frames come from another run of the reference than its CSV (they differ by ~1e-7 in F)
                assert np.abs(s.get_c() - fields["c"][i]).max() < 5e-6
                assert np.abs(s.get_phi() - fields["phi"][i]).max() < 5e-6


def test_fem_be_newton_cap_is_the_references_and_failure_restores_state(lib, golden_dir):
    """Row a9 (solver contract): with the default cap -- the reference's maximum_iterations = 10 (bench1.py:88) -- a
    solve that needs more reports ok = 0 after exactly 10 iterations with the state untouched, and the caller's
    halve-dt retry (bench1.py:164-177) then succeeds.  Rows 0-20 of the committed grid need <= 7 iterations; row 21
    (dt = 12.8) needs 24 with exact linear solves."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be") as s:       # max_newton = 0 -> 10
        s.set_ic_bm1(0.5, 0.05)
        tprev = 0.0
        for i in range(21):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok and s.last_iters <= 7
            tprev = csv[i, 0]
        before_c, before_mu, t_before = s.get_c(), s.get_mu(), s.t
        dt = csv[21, 0] - tprev
        assert abs(dt - 12.8) < 1e-9
        ok, _, _ = s.step(dt, 1, check=True)
        assert not ok and s.last_iters == 10
        np.testing.assert_array_equal(s.get_c(), before_c)
        np.testing.assert_array_equal(s.get_mu(), before_mu)
        assert s.t == t_before
        ok, _, _ = s.step(0.5 * dt, 1, check=True)       # bench1.py:171: dt.assign(max(0.5 * dt, dt_min)); retry
        assert ok and s.last_iters <= 10 and abs(s.t - (t_before + 6.4)) < 1e-9


def test_schemes_converge_to_the_reference_algorithm(lib):
    """north_star: "free-energy trajectory within 1e-4 relative of the FEniCS reference on PFHub BM1".  The committed CSV
    carries the backward-Euler error of its own dt (3.5e-2 at t = 7.9), so the throughput schemes are pinned to the
    reference's ALGORITHM instead (SURVEY 7.0-5 route (a)): GPU fem_be runs (the mode that reproduces the committed CSV
    to 5e-9) Richardson-extrapolated in dt (0.1 / 0.05 / 0.025) and h (2 / 1) against the FD scheme extrapolated in dt
    and h (1 / 0.5) and the spectral scheme extrapolated in dt, at t = 0.7, 3.1, 4.7, 7.9.
    Measured (results/CONVERGENCE.md): FD vs reference-algorithm limit <= 1.4e-5, spectral <= 3.1e-5."""
    from pfhubbenchmarks_amd import verification as V
    ts = (0.7, 3.1, 4.7)           # results/CONVERGENCE.md also lists t = 7.9 (1.4e-5 / 3.1e-5); dropped here for run time
    fem, _ = V.fem_be_limit(ts)
    fd, _ = V.fd_limit(ts)
    sp, _ = V.spectral_limit(ts)
    tol = 1e-4                                     # north_star's tolerance
    assert (np.abs(fd - fem) <= tol * np.abs(fem)).all(), (fd, fem)
    assert (np.abs(sp - fem) <= tol * np.abs(fem)).all(), (sp, fem)
    assert (np.abs(fd - sp) <= 0.5 * tol * np.abs(sp)).all(), (fd, sp)
    # and the un-extrapolated production settings stay within their known first-order distance of that limit
    raw = V.grid_energy("fd", 200, 0.00125, ts)
    assert (np.abs(raw - fem) <= 3e-4 * np.abs(fem)).all(), (raw, fem)


def test_bm6_fd_scheme_converges_to_the_reference_algorithm(lib):
    """The same pin for BM6's throughput scheme (explicit FD Cahn-Hilliard + FFT Poisson solve with the reference's
    Dirichlet / no-flux phi, explicit coupling): GPU fem_be runs (monolithic c, mu, phi backward Euler = bench6.py,
    reproduces results/bench6_out.csv to 5.6e-7) Richardson-extrapolated in dt (0.01 / 0.005 / 0.0025) and h (1 / 0.5)
    against the FD scheme extrapolated in dt and h (1 / 0.5) at t = 0.1, 0.3.  Measured: 2.3e-9 and 2.4e-8 relative."""
    from pfhubbenchmarks_amd import verification as V
    ts = (0.1, 0.3)
    fem, _ = V.fem_be_limit(ts, dt=0.01, model="bm6")
    fd, _ = V.fd_limit_bm6(ts)
    assert (np.abs(fd - fem) <= 1e-6 * np.abs(fem)).all(), (fd, fem)
    # the energy moves by 1e-3 relative over this interval: the agreement is 4 orders finer than the signal
    assert abs(fem[1] - fem[0]) > 5e-4 * fem[0]


def test_no_scheme_reads_memory_that_nobody_wrote(lib):
    """Fresh processes get zero pages from the driver; recycled memory holds anything.  With the test hook of the allocator
    (pfk_set_tuning key 11: every new device allocation filled with a byte pattern -- 0x3F.. is the double 4.8e-4, 0xFF.. a
    NaN) every scheme must return BIT-identical fields and diagnostics: explicit FD (2-D multi-step, 3-D fused, mirror box),
    spectral (2-D, 3-D power-of-two with chunked passes, mixed radix, BM6), BM6 FD + Poisson (periodic box; the no-flux boxes are
    an open item, see below), BM2 / BM3 explicit FD (cell kernels and streaming kernels), diagnostics included."""
    rng = np.random.default_rng(23)

    def cases():
        yield dict(dim=2, n=(128, 96), h=1.0, scheme="fd"), 1e-3
        yield dict(dim=3, n=(128, 64, 48), h=1.0, scheme="fd"), 1e-3
        yield dict(dim=3, n=(33, 17, 9), h=1.0, scheme="fd", bc="mirror"), 1e-3
        yield dict(dim=2, n=(512, 128), h=1.0, scheme="spectral"), 1e-2
        yield dict(dim=3, n=(128, 128, 256), h=1.0, scheme="spectral"), 1e-2
        yield dict(dim=3, n=(40, 96, 200), h=1.0, scheme="spectral"), 1e-2
        yield dict(dim=3, n=(128, 128, 128), h=1.0, scheme="spectral", model="bm6"), 1e-2
        yield dict(dim=3, n=(128, 128, 128), h=1.0, scheme="fd", model="bm6"), 5e-4
        # OPEN (round 4): the no-flux BM6 boxes (FD scheme + Dirichlet Poisson solve on the physical nodes) are NOT in
        # this list -- dict(dim=3, n=(65, 17, 13), scheme="fd", model="bm6", bc="mirror") deviates by 9e-7 under the 0x3F
        # pattern (NaN under 0xFF) when the whole -m gpu suite has run before it in the same process, and not when this test
        # runs alone or after any subset of the suite that was tried: a read outside what the route wrote, not yet found

    def run(fill):
        assert lib.pfk_set_tuning(11, fill) == 0
        out = []
        try:
            os.environ["PFHIP_SPECTRAL_3D"] = "lds"
            os.environ["PFHIP_FFT3D_CHUNK"] = "24,2"
            for kw, dt in cases():
                shape = tuple(kw["n"])[::-1]
                c0 = 0.5 + 0.05 * np.random.default_rng(len(out)).standard_normal(shape)
                with PhaseFieldSolver(**kw) as s:
                    s.set_c(c0)
                    d0 = np.array(s.diagnostics())
                    s.step(dt, 3)
                    out.append((kw, s.get_c(), d0, np.array(s.diagnostics())))
            for model, n in (("bm2", (128, 32, 8)), ("bm3", (128, 32, 8)), ("bm2", (40, 24)), ("bm3", (48, 40))):
                with PhaseFieldSolver(dim=len(n), n=n, h=1.0, scheme="fd", model=model) as s:
                    (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
                    s.step(1e-3, 3)
                    out.append(((model, n), s.get_c() if model == "bm2" else s.get_field("phi"), None, np.array(s.diagnostics())))
        finally:
            os.environ.pop("PFHIP_SPECTRAL_3D", None)
            os.environ.pop("PFHIP_FFT3D_CHUNK", None)
            assert lib.pfk_set_tuning(11, -1) == 0
        return out
    import ctypes as C
    import os
    import torch

    class Raw:      # a raw device pointer as a CUDA array (to look at what pf_device_malloc returned)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    assert lib.pfk_set_tuning(11, 0x3F) == 0           # the hook is live: a new allocation comes back filled
    ptr = C.c_void_p()
    assert lib.pf_device_malloc(C.byref(ptr), C.c_size_t(8 * 4096)) == 0
    seen = torch.as_tensor(Raw(ptr.value, 4096), device="cuda").cpu().numpy().copy()
    assert lib.pf_device_free(ptr) == 0 and lib.pfk_set_tuning(11, -1) == 0
    assert (seen.view(np.uint8) == 0x3F).all()
    ref = run(-1)
    for fill in (0x3F, 0xFF):
        got = run(fill)
        for (kw, c, d0, d1), (_, c2, e0, e1) in zip(ref, got):
            np.testing.assert_array_equal(c, c2, err_msg="fill 0x%02X %r" % (fill, kw))
            np.testing.assert_array_equal(d1, e1, err_msg="fill 0x%02X %r" % (fill, kw))
            if d0 is not None:
                np.testing.assert_array_equal(d0, e0, err_msg="fill 0x%02X %r" % (fill, kw))
    del rng


def test_fem_be_does_not_depend_on_what_the_memory_held_before(lib):
    """A BE-parity run must not depend on the history of the process.  Round 4: the same BM2 run (200 intervals, dt = 0.01,
    t = 0.02) returned F = 5405.5 in a fresh process and 4088.6 (same C) after explicit-FD handles and three smaller BE
    handles of the same process -- the create-time hipMemsets of csrc/fem_be.hip run on the legacy default stream,
    asynchronously to the host, and one of them landed AFTER the initial-condition kernel that the caller launches on the
    handle's non-blocking stream (eta_2..4 = 0).  fembe_create now waits for the device before it returns.  (Filling every
    array of the mode with garbage changes nothing: it reads no memory it did not write, profiles/r04/fem_be_order_dependence.log.)
    The run before and the run after a batch of other handles agree to 1e-12 relative for BM2 and for BM1 (the library GEMMs of
    the dense levels are not bitwise reproducible from call to call: 1 ulp differences occur; the defect was 25 %)."""
    from pfhubbenchmarks_amd import verification as V
    ts = (0.02,)
    before = V.multi_energy("bm2", "fem_be", 200, 0.01, ts)
    with PhaseFieldSolver(dim=2, n=201, h=1.0, bc="mirror", scheme="fem_be") as s:
        s.set_ic_bm1()
        s.step(0.1, 1, check=True)
        b1 = np.array(s.diagnostics())
    V.multi_fd_limit("bm2", ts)                                   # explicit-FD handles on 200^2 / 400^2 lattices, created and freed
    for d in (0.01, 0.005, 0.0025):                               # ... and the sequence that exposed it
        V.multi_energy("bm2", "fem_be", 100, d, ts)
    for n in (256, 384):
        with PhaseFieldSolver(dim=3, n=(n, 64, 32), h=1.0, scheme="fd") as s:
            s.set_c(np.ones((32, 64, n)))
            s.step(1e-3, 2)
    with PhaseFieldSolver(dim=2, n=512, h=1.0, scheme="spectral") as s:
        s.set_c(np.ones((512, 512)))
        s.step(1e-2, 2)
    after = V.multi_energy("bm2", "fem_be", 200, 0.01, ts)
    np.testing.assert_allclose(after, before, rtol=1e-12, atol=0.0)
    with PhaseFieldSolver(dim=2, n=201, h=1.0, bc="mirror", scheme="fem_be") as s:
        s.set_ic_bm1()
        s.step(0.1, 1, check=True)
        np.testing.assert_allclose(np.array(s.diagnostics()), b1, rtol=1e-12, atol=1e-12)
    assert abs(before[0, 0] - 5405.51233352) <= 1e-8 * 5405.5     # (the fresh-process value, profiles/r04/fem_be_order_dependence.log)


def test_bm2_fd_scheme_converges_to_the_reference_algorithm(lib):
    """BM2's explicit FD scheme pinned to the reference's algorithm during the fast initial transient (F falls from 6514 to
    5334 by t = 0.02): GPU fem_be (bench2.py's own discretisation, reproduces results/bench2_out.csv to 3e-10) Richardson-
    extrapolated in dt (0.01 / 0.005 / 0.0025 at h = 2; 0.01 / 0.005 at h = 1: 242 406 unknowns) and h, against the FD scheme
    extrapolated in dt and h (1 / 0.5).  Measured (tools/multi_convergence_probe.py bm2 full): 3.8e-5 at t = 0.02, 3.8e-6
    at t = 0.04; the reference's own mesh (h = 2) alone is 1.2e-3 from that limit."""
    from pfhubbenchmarks_amd import verification as V
    ts, dt = (0.02,), 0.01
    fd, per_h = V.multi_fd_limit("bm2", ts)
    fem2, runs = V.multi_fem_dt_limit("bm2", ts, dt)
    f1 = V.multi_energy("bm2", "fem_be", 200, dt, ts)
    f2 = V.multi_energy("bm2", "fem_be", 200, dt / 2, ts)
    _, b2dt2 = V.quad_extrapolate(*runs)
    g1 = 2.0 * (f2 - b2dt2 / 4.0) - (f1 - b2dt2)          # h = 1, dt -> 0 (dt^2 term taken from the h = 2 runs)
    star = (4.0 * g1 - fem2) / 3.0                         # (h, dt) -> 0
    relF = np.abs(fd[:, 0] - star[:, 0]) / np.abs(star[:, 0])
    assert (relF <= 1e-4).all(), (fd, star)
    assert (np.abs(fd[:, 1] - star[:, 1]) <= 1e-5 * star[:, 1]).all()      # total solute: trapezoid vs P1 functional of the IC
    assert (np.abs(fem2[:, 0] - star[:, 0]) > 5e-4 * star[:, 0]).all()     # ... and the h = 2 mesh alone is visibly off
    assert abs(per_h[400][0, 0] - per_h[200][0, 0]) < 5e-5 * star[0, 0]    # the FD scheme itself has converged in h


def test_bm3_fd_scheme_against_the_reference_algorithm_on_its_mesh(lib):
    """BM3: the reference's mesh (h = 960/350 = 2.74) under-resolves its own initial interface (width 1, bench3.py:54), so
    its trajectory is mesh-dependent and a 1e-4 pin at that resolution does not exist: at t = 0.5 the explicit FD scheme and
    the reference's algorithm (GPU fem_be, dt -> 0) on the SAME mesh agree to 2e-5 in F -- but F is dominated by the
    constant energy of the undercooled melt; the solid fraction, which does move, differs by ~9 %.  Asserted: both release
    energy and solidify, solid fractions within 20 %; refinement moves the FD result (no mesh convergence at this h)."""
    from pfhubbenchmarks_amd import verification as V
    ts = (0.5,)
    fd_h = V.multi_energy("bm3", "fd", 350, 0.5 / 8, ts)
    fd_h2 = V.multi_energy("bm3", "fd", 700, 0.5 / 32, ts)
    f = [V.multi_energy("bm3", "fem_be", 350, d, ts) for d in (0.25, 0.125)]
    fem = 2.0 * f[1] - f[0]                                  # dt -> 0, first order
    assert abs(fd_h[0, 0] - fem[0, 0]) <= 5e-5 * fem[0, 0]
    ic = {}
    for scheme in ("fd", "fem_be"):                          # each scheme's own functional of the same interpolated IC
        with PhaseFieldSolver(dim=2, n=351, h=960.0 / 350, bc="mirror", scheme=scheme, model="bm3") as s0:
            s0.set_ic_bm3()
            ic[scheme] = s0.diagnostics()[:2]
    assert abs(ic["fem_be"][0] - 2122262.9930579877) < 1e-3  # = oracle/fem_multi.py at t = 0 (oracle/logs/fem_multi_bm3.log)
    assert fd_h[0, 0] < ic["fd"][0] and fem[0, 0] < ic["fem_be"][0]                  # both release energy ...
    assert fd_h[0, 1] > ic["fd"][1] and fem[0, 1] > ic["fem_be"][1]                  # ... and solidify
    assert abs(fd_h[0, 1] - fem[0, 1]) < 0.2 * fem[0, 1]                             # measured: 9 % apart
    assert fd_h2[0, 1] != fd_h[0, 1]                         # not mesh-converged at the reference's resolution


def test_drivers_save_solution_and_process_bench1(lib, golden_dir, tmp_path):
    """the drivers' per-step field dump (bench1.py:116-119,190-191) read back by the counterpart of
    dolfin/process_bench1.py:9-43 with stats.csv and re-emitted as a PVD series -- BE-parity mode (crossed mesh) and a
    grid scheme."""
    import os
    from pfhubbenchmarks_amd import io as pio
    from pfhubbenchmarks_amd import postprocess
    from pfhubbenchmarks_amd.drivers import run_bench1, run_fem_be
    fields = np.load(os.path.join(golden_dir, "bm1_fields.npz"))
    out = str(tmp_path / "be")
    rows, _ = run_fem_be("bench1", "fixture", out_dir=out, verbose=False, max_rows=4, save_solution=True)
    mesh, times, cs, stats = postprocess.process_bench1(os.path.join(out, "bench1"))
    assert mesh["kind"] == "crossed" and len(cs) == 4 and stats.shape == (4, 3)
    np.testing.assert_allclose(times, fields["times"][:4], atol=1e-12)
    np.testing.assert_allclose(stats[:, 1], rows[:, 1], rtol=1e-9)
    for i in range(4):
        assert np.abs(cs[i] - fields["c"][i]).max() < 5e-9                    # the reference's own snapshots
    files = postprocess.write_series(os.path.join(out, "bench1"), mesh, times, cs)
    got = pio.read_vtu_pointdata(files[3])["c"]
    np.testing.assert_allclose(got, cs[3], rtol=1e-15, atol=0)
    out2 = str(tmp_path / "fd")
    rows2, _ = run_bench1(intervals=100, scheme="fd", end_time=0.5, out_dir=out2, save_solution=True, verbose=False)
    mesh2, times2, cs2, stats2 = postprocess.process_bench1(os.path.join(out2, "bench1"))
    assert mesh2["kind"] == "grid" and mesh2["h"] == 2.0 and len(cs2) == rows2.shape[0] == 3
    assert cs2[0].shape == (101, 101) and abs(cs2[2].mean() - 0.5126) < 3e-3
    files2 = postprocess.write_series(os.path.join(out2, "bench1"), mesh2, times2, cs2)
    np.testing.assert_array_equal(pio.read_vtu_pointdata(files2[1])["c"].reshape(101, 101), cs2[1])


def test_fem_be_bm2_against_reference_rows_and_oracle(lib, golden_dir):
    """PFHub BM2 (dolfin/bench2.py: Cahn-Hilliard + 4 Allen-Cahn fields, P1^6 on the crossed mesh,
    backward Euler, Newton) through the generic multi-field kernels of the BE-parity mode.  (a) t = 0: fields equal the
    oracle's interpolated initial condition, F and C equal the oracle's; (b) the first rows of the reference's committed
    results/bench2_out.csv: F within 1e-8, C within 1e-9 (the CPU oracle oracle/fem_multi.py reproduces these rows to
    1e-11: oracle/logs/fem_multi_bm2_cp.log); (c) rollback / failed-solve semantics."""
    import os
    from oracle import fem_multi
    csv = np.loadtxt(os.path.join(golden_dir, "bench2_out.csv"), delimiter=",", skiprows=1)
    o = fem_multi.MultiFieldBE("bm2", newton_max=100)
    with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", model="bm2", max_newton=100) as s:
        s.set_ic_bm2()
        for f, name in enumerate(o.m.fields):
            assert np.abs(s.get_field(name) - o.u[f]).max() < 1e-14, name
        F, C, _ = s.diagnostics()
        Fo, Co = o.diagnostics()
        assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)
        assert abs(C - 20504.4690550850) < 1e-8
        fields = np.load(os.path.join(golden_dir, "bm2_fields.npz"))     # the reference's own VTU snapshots, rows 0..3
        tprev = 0.0
        for i in range(10):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok and s.last_iters <= 10, (i, s.last_iters)
            tprev = csv[i, 0]
            F, C, _ = s.diagnostics()
            assert abs(F - csv[i, 1]) <= 1e-8 * csv[i, 1], (i, F, csv[i, 1])
            assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2], (i, C, csv[i, 2])
            if i < 4:
                assert abs(fields["times"][i] - csv[i, 0]) < 1e-9
                for name in ("c", "eta1", "eta2", "eta3", "eta4"):
                    err = np.abs(s.get_field(name) - fields[name][i]).max()
                    assert err < 1e-7, (i, name, err)
        before = {n: s.get_field(n) for n in ("c", "mu", "eta3")}
        ok, _, _ = s.step(csv[10, 0] - tprev, 1, check=True)
        assert ok
        s.rollback()
        for n, v in before.items():
            np.testing.assert_array_equal(s.get_field(n), v)
        # one oracle step from the GPU state: the two Newton solves land on the same root
        o.u = np.stack([s.get_field(n) for n in o.m.fields])
        its, ok_o = o.step(csv[10, 0] - tprev)
        ok, _, _ = s.step(csv[10, 0] - tprev, 1, check=True)
        assert ok and ok_o and s.last_iters == its
        for f, name in enumerate(o.m.fields):
            assert np.abs(s.get_field(name) - o.u[f]).max() < 1e-9, name


@pytest.mark.parametrize("model,shape", [("bm2", (40, 64)), ("bm2", (6, 10, 34)), ("bm3", (48, 96)), ("bm3", (5, 12, 20)),
                                         ("bm2", (3, 2)), ("bm3", (2, 1, 4)), ("bm2", (12, 16, 128)), ("bm3", (9, 32, 256)),
                                         ("bm2", (4, 48, 384)), ("bm3", (70, 16, 128)), ("bm2", (5, 512, 512))])
def test_multifield_fd_schemes_bit_exact_vs_numpy_oracle(lib, model, shape):
    """PF_SCHEME_FD_EXPLICIT for PF_MODEL_BM2 (c + 4 order parameters: mu pass + update pass) and PF_MODEL_BM3 (U, phi):
    periodic 2-D / 3-D boxes, random fields, several steps: BIT-identical to oracle/multi_fd.py (same operation order, no
    fma on either side); diagnostics to 1e-13; rollback; blow-up guard.  The last five shapes tile (x a multiple of 128, y
    of 16, >= 4 planes; 512 x 512 x 5 = full-width planes of the bench workloads) and run on the streaming LDS-tiled kernels (mfd_stream_kernel: several z-chunks, chunk ends, wraps)."""
    from oracle import multi_fd
    dim = len(shape)
    n = shape[::-1]
    rng = np.random.default_rng(sum(shape) + (2 if model == "bm2" else 3))
    names = ("c", "eta1", "eta2", "eta3", "eta4") if model == "bm2" else ("U", "phi")
    if model == "bm2":
        u = np.stack([0.5 + 0.1 * rng.standard_normal(shape)] + [0.3 + 0.3 * rng.random(shape) for _ in range(4)])
        dt, h, step = 2e-3, 1.3, multi_fd.bm2_step
    else:
        u = np.stack([-0.3 + 0.05 * rng.standard_normal(shape), np.clip(rng.standard_normal(shape), -1.0, 1.0)])
        dt, h, step = 5e-3, 0.9, multi_fd.bm3_step
    u3 = u.reshape((len(names),) + ((1,) + shape if dim == 2 else shape))
    with PhaseFieldSolver(dim=dim, n=n, h=h, scheme="fd", model=model) as s:
        assert s.status.startswith("fd: explicit multi-field")
        assert ("streaming" in s.status) == (dim == 3 and shape[2] % 128 == 0 and shape[1] % 16 == 0 and shape[0] >= 4)
        for f, name in enumerate(names):
            s.set_field(name, u[f])
        F, C, _ = s.diagnostics()
        dom = float(np.prod(n)) * h ** dim
        Fo, Co = multi_fd.diagnostics(model, u3, h, dim, domain=dom)
        assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)
        ref = u3
        for k in range(1, 8):
            ok, umin, umax = s.step(dt, 1, check=True)
            ref = step(ref, dt, h)
            assert ok and umin == ref.min() and umax == ref.max()
            if k in (1, 4, 7):
                for f, name in enumerate(names):
                    np.testing.assert_array_equal(s.get_field(name), ref[f].reshape(shape), err_msg="%s step %d" % (name, k))
        F, C, _ = s.diagnostics()
        Fo, Co = multi_fd.diagnostics(model, ref, h, dim, domain=dom)
        assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)
        before = s.get_field(names[-1])
        s.step(dt, 3)
        s.rollback()
        np.testing.assert_array_equal(s.get_field(names[-1]), step(step(ref, dt, h), dt, h)[-1].reshape(shape))
        with pytest.raises(L.PfhipError):
            s.rollback()
        del before
        ok, _, _ = s.step(50.0 * dt if model == "bm2" else 400.0 * dt, 60, check=True)     # far beyond the stability limit
        assert not ok
        with pytest.raises(L.PfhipError):
            s.get_field("mu")                                                          # never stored by the FD scheme


@pytest.mark.parametrize("model,shape,nranks", [("bm2", (12, 16, 128), 1), ("bm2", (12, 16, 128), 3), ("bm3", (9, 32, 256), 2),
                                                ("bm3", (7, 6, 10), 3), ("bm2", (9, 5, 12), 2), ("bm2", (24, 16, 128), 4)])
def test_multifield_fd_slabs_bit_exact_vs_single_box(lib, model, shape, nranks):
    """BM2 / BM3 explicit FD decomposed along z (HipMultiFieldSlabEngine: ghost = 2 / 1 planes per side of every field,
    refreshed before each step -- pf_field_halo_layout): `nranks` slab handles on this one GPU, the ring exchange done by
    copying the planes between their buffers exactly as MultiFieldSlabSolver's isend / irecv pairs would (uneven partitions
    included), against oracle/multi_fd.py on the whole periodic box: BIT-identical fields after every step; the sum of the
    ranks' pf_diagnostics_local = the box's diagnostics to 1e-13.  Shapes that tile run the streaming LDS-tiled kernels
    (the one-pass BM2 kernel included) on the ghosted local box.  Equations: dolfin/bench2.py:76-113, bench3.py:63-97."""
    import torch
    from oracle import multi_fd
    from pfhubbenchmarks_amd.solver import HipMultiFieldSlabEngine, MultiFieldSlabSolver, slab_partition
    nz, ny, nx = shape
    rng = np.random.default_rng(sum(shape) + nranks)
    names = ("c", "eta1", "eta2", "eta3", "eta4") if model == "bm2" else ("U", "phi")
    if model == "bm2":
        u = np.stack([0.5 + 0.1 * rng.standard_normal(shape)] + [0.3 + 0.3 * rng.random(shape) for _ in range(4)])
        dt, h, step = 2e-3, 1.3, multi_fd.bm2_step
    else:
        u = np.stack([-0.3 + 0.05 * rng.standard_normal(shape), np.clip(rng.standard_normal(shape), -1.0, 1.0)])
        dt, h, step = 5e-3, 0.9, multi_fd.bm3_step
    engines = [HipMultiFieldSlabEngine(model, (nx, ny, nz), h, nranks, r, 0) for r in range(nranks)]
    try:
        for e in engines:
            for f, name in enumerate(names):
                e.set_local(name, u[f, e.z0:e.z0 + e.nz])

        def exchange():
            for e in engines:
                e.sync()
            for r, e in enumerate(engines):
                g, n = e.ghost, e.nz
                lo, hi = engines[e.rank_lo], engines[e.rank_hi]
                b = e.buffers[e.cur]
                b[:, 0:g].copy_(lo.buffers[lo.cur][:, lo.nz:lo.nz + g])            # my low ghosts = low neighbour's last planes
                b[:, n + g:n + 2 * g].copy_(hi.buffers[hi.cur][:, hi.ghost:2 * hi.ghost])
            torch.cuda.synchronize()

        ref = u
        dom = float(nx * ny * nz) * h ** 3
        for k in range(1, 6):
            if k % 2 == 0:                                  # the overlapped protocol (MultiFieldSlabSolver.step): the planes that
                for e in engines:                           # need no ghosts run first, on STALE ghost planes
                    e.step_begin(dt)
                exchange()
                for e in engines:
                    e.step_finish()
            else:
                exchange()
                for e in engines:
                    e.step_local(dt)
            ref = step(ref, dt, h)
            for e in engines:
                for f, name in enumerate(names):
                    np.testing.assert_array_equal(e.get_local(name), ref[f, e.z0:e.z0 + e.nz], err_msg="%s step %d" % (name, k))
        exchange()                                         # the energy's forward differences reach one plane up
        tot = sum(e.diag_local() for e in engines)
        Fo, Co = multi_fd.diagnostics(model, ref, h, 3, domain=dom)
        assert abs(tot[0] - Fo) <= 1e-13 * abs(Fo) and abs(tot[1] - Co) <= 1e-13 * abs(Co)
        if nranks == 1:                                    # the solver class itself, the rank being its own neighbour
            s = MultiFieldSlabSolver(engines[0])
            s.step(dt, 2)
            ref2 = step(step(ref, dt, h), dt, h)
            np.testing.assert_array_equal(engines[0].get_local(names[0]), ref2[0])
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("model,n,nranks", [("bm2", (9, 7, 9), 2), ("bm3", (65, 9, 7), 2), ("bm2", (65, 9, 7), 3), ("bm3", (6, 5, 13), 4)])
def test_multifield_fd_mirror_slabs_match_the_single_gpu_box(lib, model, n, nranks):
    """BM2 / BM3 explicit FD, the reference's no-flux boxes (dolfin/bench2.py:66-69, bench3.py:59-61: natural boundary
    conditions) on several slabs: as on one GPU the box lives on its even extension along all three axes, the ring of slab
    handles runs over the 2 (nz - 1) lattice planes.  Same initial condition (pf_set_ic_bm2 / bm3), five steps: the
    physical-node corner of the gathered lattice stack equals the single-GPU handle's fields BIT for bit; summed local
    diagnostics = the single handle's to 1e-13."""
    import torch
    from pfhubbenchmarks_amd.solver import HipMultiFieldSlabEngine
    nx, ny, nz = n
    names = ("c", "eta1", "eta2", "eta3", "eta4") if model == "bm2" else ("U", "phi")
    dt, h = (2e-3, 1.3) if model == "bm2" else (5e-3, 0.9)
    engines = [HipMultiFieldSlabEngine(model, n, h, nranks, r, 0, bc="mirror") for r in range(nranks)]
    try:
        for e in engines:
            e.set_ic()

        def exchange():
            for e in engines:
                e.sync()
            for e in engines:
                g, m = e.ghost, e.nz
                lo, hi = engines[e.rank_lo], engines[e.rank_hi]
                b = e.buffers[e.cur]
                b[:, 0:g].copy_(lo.buffers[lo.cur][:, lo.nz:lo.nz + g])
                b[:, m + g:m + 2 * g].copy_(hi.buffers[hi.cur][:, hi.ghost:2 * hi.ghost])
            torch.cuda.synchronize()

        with PhaseFieldSolver(dim=3, n=n, h=h, bc="mirror", scheme="fd", model=model) as s:
            (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
            for k in range(5):
                exchange()
                for e in engines:
                    e.step_local(dt)
                s.step(dt, 1)
            for name in names:
                stack = np.concatenate([e.get_local(name) for e in engines], axis=0)
                assert stack.shape == (2 * (nz - 1), 2 * (ny - 1), 2 * (nx - 1))
                np.testing.assert_array_equal(stack[:nz, :ny, :nx], s.get_field(name), err_msg=name)
                np.testing.assert_array_equal(stack[nz:, :ny, :nx], s.get_field(name)[nz - 2:0:-1], err_msg=name + " (mirror image)")
            exchange()
            tot = sum(e.diag_local() for e in engines)
            F, C, _ = s.diagnostics()
            assert abs(tot[0] - F) <= 1e-13 * abs(F) and abs(tot[1] - C) <= 1e-13 * abs(C)
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("model", ["bm2", "bm3"])
def test_multifield_fd_no_flux_box_is_the_even_extension(lib, model):
    """PF_BC_MIRROR (the reference's natural boundary condition, bench2.py:113, bench3.py:100) for the multi-field FD
    schemes: nodes in, even extension inside, nodes out -- equal to the oracle on the extended lattice; the device
    initial condition equals the oracle's formula at the nodes; diagnostics carry the 2^-d volume factor."""
    from oracle import multi_fd
    n = 21
    h = 200.0 / (n - 1) if model == "bm2" else 960.0 / 350
    names = ("c", "eta1", "eta2", "eta3", "eta4") if model == "bm2" else ("U", "phi")
    with PhaseFieldSolver(dim=2, n=n, h=h, bc="mirror", scheme="fd", model=model) as s:
        (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
        ic = (multi_fd.ic_bm2 if model == "bm2" else multi_fd.ic_bm3)(n, n, h)
        got = np.stack([s.get_field(nm) for nm in names])
        assert np.abs(got - ic[:, 0]).max() < 1e-14
        for f, nm in enumerate(names):                    # identical start -> bit-exact evolution
            s.set_field(nm, ic[f, 0])
        ext = multi_fd.even_extend(ic)
        dt = 1e-3 if model == "bm2" else 2e-2
        step = multi_fd.bm2_step if model == "bm2" else multi_fd.bm3_step
        for _ in range(5):
            ext = step(ext, dt, h)
        s.step(dt, 5)
        for f, nm in enumerate(names):
            np.testing.assert_array_equal(s.get_field(nm), ext[f, 0, :n, :n])
        F, C, _ = s.diagnostics()
        Fo, Co = multi_fd.diagnostics(model, ext, h, 2, mirror=True, domain=((n - 1) * h) ** 2)
        assert abs(F - Fo) <= 1e-13 * abs(Fo) and abs(C - Co) <= 1e-13 * abs(Co)


@pytest.mark.parametrize("model,n,h", [("bm2", 41, 2.0), ("bm3", 61, 960.0 / 350), ("bm6", 41, 1.0), ("bm1", 41, 2.0)])
def test_fem_be_static_condensation_equals_the_full_block_solve(lib, model, n, h, monkeypatch):
    """The generic BE-parity path eliminates the cell-centre unknowns cell by cell before the block-tridiagonal solve
    (gen_cell_jacobian_kernel / gen_condense_kernel / gen_backsub_kernel).  PFHIP_FEM_CONDENSE=0 keeps them in the blocks
    (the first implementation, pinned to the reference CSVs in round 2; for BM1 / BM6 that is the c / mu / phi kernels with
    BM6's Dirichlet rows): same Newton iterates up to rounding -- same
    iteration counts, fields to 1e-10, incl. the cp line search (BM2).  Also against the other solver variants kept behind
    switches: the sequential block Thomas solve, dense kernels on the first reduction level, one stream, row exchanges in
    every factorisation (the default factors the small batches of the dense levels without them)."""
    out = {}
    for cond in ("1", "0", "thomas", "dense0", "onestream", "allpivot"):
        monkeypatch.setenv("PFHIP_FEM_CONDENSE", "0" if cond == "0" else "1")
        monkeypatch.setenv("PFHIP_FEM_SOLVER", "thomas" if cond == "thomas" else "bcr")     # sequential block solve
        monkeypatch.setenv("PFHIP_FEM_BAND0", "0" if cond == "dense0" else "1")             # dense kernels on level 0
        monkeypatch.setenv("PFHIP_FEM_STREAMS", "1" if cond == "onestream" else "0")        # no side streams
        monkeypatch.setenv("PFHIP_FEM_PIVOT", "1" if cond == "allpivot" else "auto")        # row exchanges in every LU
        with PhaseFieldSolver(dim=2, n=n, h=h, bc="mirror", scheme="fem_be", model=model, max_newton=100) as s:
            {"bm1": s.set_ic_bm1, "bm2": s.set_ic_bm2, "bm3": s.set_ic_bm3, "bm6": s.set_ic_bm6}[model]()
            its = []
            for dt in (0.01, 0.04, 0.08):
                ok, _, _ = s.step(dt, 1, check=True)
                assert ok
                its.append(s.last_iters)
            names = {"bm1": ("c", "mu"), "bm2": ("c", "mu", "eta1", "eta4"), "bm3": ("U", "phi"),
                     "bm6": ("c", "mu", "phi")}[model]
            out[cond] = (its, [s.get_field(k) for k in names], s.diagnostics())
    assert max(out["1"][0]) >= 2, out["1"][0]
    for other in ("0", "thomas", "dense0", "onestream", "allpivot"):      # every variant lands on the same Newton iterates
        assert out["1"][0] == out[other][0], (other, out["1"][0], out[other][0])
        for a, b in zip(out["1"][1], out[other][1]):
            assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), other
        assert abs(out["1"][2][0] - out[other][2][0]) <= 1e-11 * abs(out[other][2][0]), other


def _bm23(model):
    if model == "bm2":
        return dict(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", model="bm2", max_newton=100)
    return dict(dim=2, n=351, h=960.0 / 350, bc="mirror", scheme="fem_be", model="bm3", max_newton=100)


@pytest.mark.parametrize("model,rows", [("bm2", 9), ("bm3", 7)])
def test_fem_be_pivot_policies_agree_at_production_block_sizes(lib, golden_dir, monkeypatch, model, rows):
    """The default factors the small batches of the dense reduction levels WITHOUT row exchanges when the blocks hold 400+
    unknowns -- which only the production meshes do (BM2: 101 nodes per row x 6 fields = 606, BM3: 351 x 2 = 702; the
    small meshes of the variant test above never take that branch).  On those meshes, along the reference's committed time
    grid (dolfin/bench2.py:226-262, bench3.py:202-258; BM2's row 5 needs the cp line search): PFHIP_FEM_PIVOT=1 (row
    exchanges everywhere) and the default give identical Newton iteration counts, fields within 1e-10 and the same CSV
    rows; pf_get_stat shows that the default really took the un-pivoted branch and never needed the repeat."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench%s_out.csv" % model[-1]), delimiter=",", skiprows=1)
    names = ("c", "mu", "eta1", "eta4") if model == "bm2" else ("U", "phi")
    out = {}
    for mode in ("auto", "1"):
        monkeypatch.setenv("PFHIP_FEM_PIVOT", mode)
        with PhaseFieldSolver(**_bm23(model)) as s:
            (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
            its, npvt, tprev = [], [], 0.0
            for i in range(rows):
                ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
                assert ok, (mode, i)
                tprev = csv[i, 0]
                its.append(s.last_iters)
                npvt.append(s.stat(L.PF_STAT_FEM_NPVT_LEVELS))
                assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1, (mode, i)
                F = s.diagnostics()[0]
                assert abs(F - csv[i, 1]) <= 1e-8 * abs(csv[i, 1]), (mode, i, F, csv[i, 1])
            out[mode] = (its, npvt, [s.get_field(k) for k in names])
    assert min(out["auto"][1]) >= 3, out["auto"][1]          # the dense levels with <= 32 blocks: un-pivoted by default
    assert max(out["1"][1]) == 0, out["1"][1]
    assert out["auto"][0] == out["1"][0], (out["auto"][0], out["1"][0])
    assert max(out["auto"][0]) >= 3
    for a, b in zip(out["auto"][2], out["1"][2]):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("model,n", [("bm2", 68), ("bm2", 75), ("bm2", 101), ("bm2", 120), ("bm3", 201), ("bm3", 260), ("bm3", 351)])
def test_fem_be_own_dense_kernels_match_the_library_path(lib, monkeypatch, model, n):
    """The dense reduction levels on the repository's kernels (lu_npvt_coop_kernel: cooperative un-pivoted LU,
    lu_solve_mfma_kernel: both substitutions on fp64 MFMA tiles, gemv_sub_kernel) against the same solve on rocSOLVER /
    rocBLAS (PFHIP_FEM_GETRF=rocsolver PFHIP_FEM_TRSM=rocblas: substitutions and matrix-vector products), at block sizes that exercise ragged
    last tiles and every instantiation in use: 6 x 69 = 414, 6 x 76 = 456, 606, 6 x 121 = 726 (BM2), 2 x 202 = 404, 522, 702
    (BM3) unknowns per block -- 26..46 tiles of 16, i.e. 4, 5 and 6 tiles per wave.  Same Newton iteration counts, fields
    within 1e-10, one attempt per step (no factorisation reported singular, no partner lost).  The reference side of the
    computation is dolfin/bench2.py:126-158 / bench3.py:111-140 (one Newton solve per backward-Euler step)."""
    kw = dict(_bm23(model))
    kw["n"] = n
    kw["h"] = (200.0 if model == "bm2" else 960.0) / (n - 1)
    names = ("c", "mu", "eta1", "eta4") if model == "bm2" else ("U", "phi")
    dts = (0.01, 0.02, 0.04) if model == "bm2" else (0.1, 0.2, 0.4)
    out = {}
    for mode in ("own", "library"):
        for k in ("PFHIP_FEM_GETRF", "PFHIP_FEM_TRSM"):
            monkeypatch.delenv(k, raising=False)
        if mode == "library":
            monkeypatch.setenv("PFHIP_FEM_GETRF", "rocsolver")
            monkeypatch.setenv("PFHIP_FEM_TRSM", "rocblas")
        with PhaseFieldSolver(**kw) as s:
            (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
            its = []
            for dt in dts:
                ok, _, _ = s.step(dt, 1, check=True)
                assert ok, (mode, dt)
                assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1, (mode, dt)
                assert s.stat(L.PF_STAT_FEM_NPVT_LEVELS) >= 3, (mode, dt)      # the branch under test was taken
                its.append(s.last_iters)
            out[mode] = (its, [s.get_field(k) for k in names])
    assert out["own"][0] == out["library"][0], (out["own"][0], out["library"][0])
    for a, b in zip(out["own"][1], out["library"][1]):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())


def test_fem_be_cooperative_lu_without_its_workgroups_is_reported_and_switched_off(lib, monkeypatch):
    """lu_npvt_coop_kernel needs G resident workgroups per matrix on one XCD (tickets by HW_REG_XCC_ID).  On a part whose XCD
    count or CU share breaks that, matrices are left without partners: the kernel must neither hang (bounded spins) nor return
    a half-factored matrix silently.  PFHIP_FEM_TEST_LU_STARVE=1 (test-only) launches half of the workgroups: the first Newton
    solve fails with the 'not my matrices' code, is repeated with row exchanges on the library path (two attempts, correct
    result), the handle switches the cooperative kernel off, and the following steps take one attempt each on rocSOLVER's
    un-pivoted LU -- same fields as an undisturbed handle to 1e-10.  (dolfin/bench2.py:126-158 is the solve this stands for.)"""
    dts = (0.01, 0.02, 0.04)
    ref = []
    with PhaseFieldSolver(**_bm23("bm2")) as s:
        s.set_ic_bm2()
        for dt in dts:
            ok, _, _ = s.step(dt, 1, check=True)
            assert ok and s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1
            ref.append((s.last_iters, s.get_field("c"), s.get_field("eta3")))
    monkeypatch.setenv("PFHIP_FEM_TEST_LU_STARVE", "1")
    with PhaseFieldSolver(**_bm23("bm2")) as s:
        s.set_ic_bm2()
        for k, dt in enumerate(dts):
            ok, _, _ = s.step(dt, 1, check=True)
            assert ok
            assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == (2 if k == 0 else 1), k
            assert (s.stat(L.PF_STAT_FEM_NPVT_LEVELS) == 0) == (k == 0), k      # the repeat pivots everywhere; later: library npvt
            # ... and the handle SAYS so (pf_status_string), it does not only count attempts
            assert "cooperative LU kernel" in s.describe() and "switched off" in s.describe(), s.describe()
            assert s.last_iters == ref[k][0]
            for a, b in zip((s.get_field("c"), s.get_field("eta3")), ref[k][1:]):
                assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
        assert "own kernels" in s.status and "switched off" not in s.status      # (the string at creation)


def test_fem_be_failed_unpivoted_solve_is_repeated_with_row_exchanges(lib, monkeypatch):
    """fembe_step's safety net, reached on purpose: PFHIP_FEM_TEST_POISON_NPVT=1 (test-only switch) spoils the first Newton
    direction of any attempt that contained an un-pivoted factorisation.  The step must then (i) restore the state,
    (ii) repeat the solve with row exchanges in EVERY factorisation -- including the first reduction level, which leaves the
    banded block-Thomas kernels for the dense pivoted ones -- and (iii) land exactly where an always-pivoting handle lands:
    same iteration count, same fields, ok = 1, two attempts on record."""
    dts = (0.01, 0.02, 0.04)          # the first steps of the committed grid (rows 0..2 of bench2_out.csv)
    monkeypatch.setenv("PFHIP_FEM_PIVOT", "auto")
    monkeypatch.setenv("PFHIP_FEM_TEST_POISON_NPVT", "1")
    with PhaseFieldSolver(**_bm23("bm2")) as s:
        s.set_ic_bm2()
        its = []
        for dt in dts:
            before = s.get_field("c")
            ok, _, _ = s.step(dt, 1, check=True)
            assert ok
            assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == 2 and s.stat(L.PF_STAT_FEM_NPVT_LEVELS) == 0
            assert np.abs(s.get_field("c") - before).max() > 0.0
            its.append(s.last_iters)
        got = [s.get_field(k) for k in ("c", "mu", "eta2")]
        s.rollback()                                      # the pre-step state of the LAST step, not of a failed attempt
        np.testing.assert_array_equal(s.get_field("c"), before)
    monkeypatch.delenv("PFHIP_FEM_TEST_POISON_NPVT")
    with PhaseFieldSolver(always_pivot=True, **_bm23("bm2")) as s:          # PF_FLAG_FEM_ALWAYS_PIVOT
        s.set_ic_bm2()
        ref_its = []
        for dt in dts:
            ok, _, _ = s.step(dt, 1, check=True)
            assert ok and s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1 and s.stat(L.PF_STAT_FEM_NPVT_LEVELS) == 0
            ref_its.append(s.last_iters)
        ref = [s.get_field(k) for k in ("c", "mu", "eta2")]
    assert its == ref_its, (its, ref_its)
    for a, b in zip(got, ref):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())


def test_fem_be_rejected_step_costs_one_solve_when_every_factorisation_pivoted(lib, golden_dir, monkeypatch):
    """A step the Newton cap rejects (the reference then halves dt, bench1.py:164-177) is reported after ONE solve whenever
    that solve pivoted throughout: under PF_FLAG_FEM_ALWAYS_PIVOT (what the drivers set with the reference's dt controller)
    for BM1 as for BM2.  Only a failed solve that really contained an un-pivoted factorisation is repeated (the default
    policy: two attempts on record, state untouched).  Along the way: BM1's first 21 committed steps take one attempt each
    under either policy, with the same Newton iteration counts."""
    import os
    monkeypatch.setenv("PFHIP_FEM_PIVOT", "auto")
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    its = {}
    for always, attempts in ((True, 1), (False, 2)):
        with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", always_pivot=always) as s:       # cap 10
            s.set_ic_bm1(0.5, 0.05)
            tprev, its[always] = 0.0, []
            for i in range(21):
                ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
                assert ok and s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1
                assert (s.stat(L.PF_STAT_FEM_NPVT_LEVELS) == 0) == always
                its[always].append(s.last_iters)
                tprev = csv[i, 0]
            before = s.get_c()
            ok, _, _ = s.step(csv[21, 0] - tprev, 1, check=True)                            # needs 24 iterations
            assert not ok and s.last_iters == 10 and s.stat(L.PF_STAT_FEM_ATTEMPTS) == attempts
            np.testing.assert_array_equal(s.get_c(), before)
    assert its[True] == its[False]
    for always, attempts in ((True, 1), (False, 2)):
        kw = dict(_bm23("bm2"), max_newton=2)
        with PhaseFieldSolver(always_pivot=always, **kw) as s:
            s.set_ic_bm2()
            before = s.get_field("eta1")
            ok, _, _ = s.step(0.64, 1, check=True)        # 2 iterations are not enough at this dt (row 7 takes 4+)
            assert not ok and s.last_iters == 2
            assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == attempts
            np.testing.assert_array_equal(s.get_field("eta1"), before)


@pytest.mark.parametrize("model", ["bm2", "bm3"])
def test_fem_be_full_reference_trajectory_bm2_bm3(lib, golden_dir, monkeypatch, model):
    """The WHOLE committed runs of dolfin/bench2.py (120 accepted steps, dt 0.01 ... 163.84; bench2.py:226-262) and
    dolfin/bench3.py (46 steps; bench3.py:202-258) through the GPU BE-parity mode with its DEFAULT pivot policy: every row's
    F within 1e-8 of the reference's CSV (C / solid fraction to the CSV's precision) -- the large-dt rows are where an
    un-pivoted Schur complement would be most at risk; none of them may need the pivoted repeat."""
    import os
    monkeypatch.setenv("PFHIP_FEM_PIVOT", "auto")
    csv = np.loadtxt(os.path.join(golden_dir, "bench%s_out.csv" % model[-1]), delimiter=",", skiprows=1)
    assert csv.shape == ((120, 3) if model == "bm2" else (46, 3))
    worst, its = 0.0, []
    with PhaseFieldSolver(**_bm23(model)) as s:
        (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
        tprev = 0.0
        for i in range(len(csv)):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok, (i, s.last_iters)
            assert s.stat(L.PF_STAT_FEM_ATTEMPTS) == 1, i
            its.append(s.last_iters)
            tprev = csv[i, 0]
            F, C, _ = s.diagnostics()
            worst = max(worst, abs(F - csv[i, 1]) / abs(csv[i, 1]))
            assert abs(F - csv[i, 1]) <= 1e-8 * abs(csv[i, 1]), (i, F, csv[i, 1])
            if model == "bm2":
                assert abs(C - csv[i, 2]) <= 1e-9 * csv[i, 2], (i, C, csv[i, 2])
            else:
                assert abs(C - csv[i, 2]) <= 2e-10 + 1e-8 * abs(csv[i, 2]), (i, C, csv[i, 2])
    print("%s full trajectory: worst rel F %.3g, Newton its %s" % (model, worst, its))


def test_fem_be_bm3_against_reference_rows(lib, golden_dir):
    """PFHub BM3 (dolfin/bench3.py: U, phi on the 350 x 350 crossed mesh, 491 402 unknowns) through the same generic
    kernels: t = 0 known answers and the first rows of results/bench3_out.csv (F within 1e-8, solid fraction to the
    CSV's printed precision)."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench3_out.csv"), delimiter=",", skiprows=1)
    with PhaseFieldSolver(dim=2, n=351, h=960.0 / 350, bc="mirror", scheme="fem_be", model="bm3", max_newton=100) as s:
        s.set_ic_bm3()
        U, phi = s.get_field("U"), s.get_field("phi")
        assert U.shape == (351 * 351 + 350 * 350,) and (U == -0.3).all()
        assert phi.max() == 1.0 and phi.min() == -1.0 and (phi > -1.0).sum() < 40       # the seed of radius 8 at the origin
        tprev = 0.0
        for i in range(6):
            ok, _, _ = s.step(csv[i, 0] - tprev, 1, check=True)
            assert ok and s.last_iters <= 8, (i, s.last_iters)
            tprev = csv[i, 0]
            F, S, _ = s.diagnostics()
            assert abs(F - csv[i, 1]) <= 1e-8 * csv[i, 1], (i, F, csv[i, 1])
            assert abs(S - csv[i, 2]) <= 6e-11, (i, S, csv[i, 2])        # the CSV prints 10 decimals of a 5e-5 number


def _lockstep(engs, op, dt=0.0):
    """Run the distributed state machine of all slab handles on the one GPU, doing by hand (tensor copies) the
    all-to-all / halo exchanges that FFTSlabSolver does over RCCL."""
    P = len(engs)
    for e in engs:
        e.dist_begin(op, dt)
    while True:
        reqs = [e.dist_advance() for e in engs]
        torch.cuda.synchronize()
        kinds = {r[0] for r in reqs}
        assert len(kinds) == 1
        kind = kinds.pop()
        if kind == "done":
            return
        if kind == "alltoall":
            for q in range(P):
                dst = reqs[q][1].view(P, -1)
                for p in range(P):
                    dst[p].copy_(reqs[p][2].view(P, -1)[q])
        else:
            nb = len(reqs[0][1])
            for b in range(nb):
                for r, e in enumerate(engs):
                    lo, hi = engs[(r - 1) % P], engs[(r + 1) % P]
                    mine = reqs[r][1][b]
                    mine[0:2].copy_(reqs[(r - 1) % P][1][b][lo.nz:lo.nz + 2])
                    mine[e.nz + 2:e.nz + 4].copy_(reqs[(r + 1) % P][1][b][2:4])
        torch.cuda.synchronize()


@pytest.mark.parametrize("nodes,P", [((9, 7, 5), 2), ((9, 7, 5), 4), ((65, 65, 65), 2), ((33, 9, 17), 4), ((9, 7, 5), 1)])
def test_bm6_reference_boundary_conditions_on_slabs(lib, nodes, P):
    """BM6 with the REFERENCE's boundary conditions across ranks (dolfin/bench6.py:77-90 under `mpirun`, README.md:22: phi = 0 /
    sin(y/7) on x = 0 / Lx, no flux elsewhere): HipFFTSlabEngine(bc="mirror", model="bm6") -- the no-flux box lives on its even
    extension along all three axes, the slabs form a ring over the 2 (nz - 1) lattice planes, and the periodic slab FFT
    transforms the odd-in-x / even-in-y,z extension of the right-hand side (= the sine x cosine x cosine transform of the
    physical problem; Dirichlet data moved to node N - 1, boundary values written back after the solve).  P rank handles on
    the one GPU, collectives emulated by copies, against (a) the single-GPU solver with the same boundary conditions (its
    Poisson solve runs the sine / cosine passes on the PHYSICAL nodes: a different route to the same discrete problem) and
    (b) the numpy oracle.  (65, 65, 65): 128^3 lattice on the hand-written passes; the others: rocFFT."""
    from oracle import bm6_fd
    from oracle.multi_fd import even_extend
    from pfhubbenchmarks_amd.solver import HipFFTSlabEngine
    npx, npy, npz = nodes
    rng = np.random.default_rng(sum(nodes) + P)
    phys = 0.5 + 0.05 * rng.standard_normal((npz, npy, npx))
    h, dt, nsteps = 1.0, 1e-3, 3
    engs = [HipFFTSlabEngine(nodes, h, P, r, 0, scheme="fd", model="bm6", bc="mirror") for r in range(P)]
    try:
        for e in engs:
            e.set_global(phys)
        _lockstep(engs, L.PF_DIST_OP_REFRESH)
        d0 = np.sum([e.diag_local() for e in engs], 0)
        for _ in range(nsteps):
            _lockstep(engs, L.PF_DIST_OP_STEP, dt)
        _lockstep(engs, L.PF_DIST_OP_REFRESH)
        d1 = np.sum([e.diag_local() for e in engs], 0)
        lat = np.concatenate([e.get_local() for e in engs], 0)           # this ring's lattice planes, physical x, y nodes
        phi_lat = np.concatenate([e.phi[2:2 + e.nz].cpu().numpy() for e in engs], 0)
    finally:
        for e in engs:
            e.close()
    assert lat.shape == (2 * (npz - 1), npy, npx)
    # (a) the single-GPU solver
    with PhaseFieldSolver(dim=3, n=nodes, h=h, bc="mirror", model="bm6") as s:
        s.set_c(phys)
        w0 = np.array(s.diagnostics())
        s.step(dt, nsteps)
        w1 = np.array(s.diagnostics())
        whole, whole_phi = s.get_c(), s.get_phi()
    assert np.abs(lat[:npz] - whole).max() <= 1e-12
    assert np.abs(lat[npz:] - whole[-2:0:-1]).max() <= 1e-12              # the mirrored planes of the ring
    np.testing.assert_allclose(d0, w0, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(d1, w1, rtol=1e-11, atol=1e-13)
    # (b) the oracle on the whole lattice
    full = even_extend(phys)
    full = np.concatenate([full, full[-2:0:-1]], axis=0)
    o = bm6_fd.BM6FD(full, h, mirror_nodes=(npx, npy))
    o.step(dt, nsteps)
    assert np.abs(lat - o.c[:, :npy, :npx]).max() <= 1e-12
    assert np.abs(phi_lat - o.phi()).max() <= 1e-11 and np.abs(whole_phi - o.phi()[:npz, :npy, :npx]).max() <= 1e-11
    assert np.all(phi_lat[:, :, 0] == 0.0)                                # the Dirichlet plane x = 0, exactly


@pytest.mark.parametrize("n,P,chunk", [((64, 24, 16), 2, None), ((128, 128, 128), 2, None), ((128, 256, 512), 4, None),
                                       ((512, 128, 128), 1, None), ((128, 256, 512), 4, "24,2"), ((128, 128, 128), 2, "7,3")])
@pytest.mark.parametrize("mode", ["spectral", "bm6"])
def test_fft_slab_modes_on_one_gpu(lib, monkeypatch, mode, n, P, chunk):
    """slab FFT modes (pf_dist_begin / pf_dist_advance): P rank handles on the one GPU, collectives emulated by
    copies, against the single-domain numpy oracles.  (64, 24, 16): rocFFT + pack / unpack kernels; the power-of-two
    boxes: the hand-written LDS-FFT passes with the all-to-all layout written / read by the y pass itself
    (fusedslab_*: radix-2^2 and radix-8 column kernels, 2 / 4 ranks and the single-rank ring).  chunk: the rank-local x / y
    passes chunk of planes by chunk of planes on side streams (PFHIP_FFT3D_CHUNK; the default at production sizes: 512^3
    over 8 ranks = 64 planes per rank in chunks of 31)."""
    from oracle import bm6_fd, ch_spectral
    from pfhubbenchmarks_amd.solver import HipFFTSlabEngine
    if chunk:
        monkeypatch.setenv("PFHIP_FFT3D_CHUNK", chunk)
    rng = np.random.default_rng(31)
    full = 0.5 + 0.05 * rng.standard_normal(n[::-1])
    engs = [HipFFTSlabEngine(n, 1.0, P, r, 0, scheme="spectral" if mode == "spectral" else "fd",
                             model="bm6" if mode == "bm6" else "bm1") for r in range(P)]
    for e in engs:
        e.set_local(full[e.z0:e.z0 + e.nz])
    dt = 1e-2 if mode == "spectral" else 1e-3
    if mode == "spectral":
        o = ch_spectral.SpectralCH(full, h=1.0)
    else:
        o = bm6_fd.BM6FD(full, 1.0)
    _lockstep(engs, 2)
    d = np.sum([e.diag_local() for e in engs], 0)
    ref = o.diagnostics()
    assert abs(d[0] - ref[0]) <= 1e-11 * abs(ref[0]) and abs(d[1] - ref[1]) <= 1e-13 * abs(ref[1])
    for _ in range(3):
        _lockstep(engs, 1, dt)
    o.step(dt, 3)
    got = np.concatenate([e.get_local() for e in engs], 0)
    assert np.abs(got - o.c).max() <= 1e-12
    _lockstep(engs, 2)
    d = np.sum([e.diag_local() for e in engs], 0)
    ref = o.diagnostics()
    assert abs(d[0] - ref[0]) <= 1e-11 * abs(ref[0]) and abs(d[1] - ref[1]) <= 1e-13 * abs(ref[1])
    for e in engs:
        e.close()


@pytest.mark.parametrize("shape", [(34, 18), (50, 130), (512, 512), (7, 258), (200, 400)])
def test_2d_multistep_launches_are_bit_identical_to_single_steps(lib, orc, shape):
    """2-D path: up to 4 time steps per launch with the tile resident in LDS (csrc: ch_fd2d_multistep_kernel) must equal
    the oracle's single steps bit for bit, for every split of nsteps into 4 / 2 / 1-step launches."""
    ny, nx = shape
    rng = np.random.default_rng(nx + ny)
    c = 0.5 + 0.1 * rng.standard_normal(shape)
    for kmax in (4, 2, 1):
        assert lib.pfk_set_tuning(3, kmax) == 0
        with PhaseFieldSolver(dim=2, n=(nx, ny), h=1.0, kernel="fused") as s:
            s.set_c(c)
            ref = c
            done = 0
            for n in (1, 2, 3, 5, 12):
                s.step(1e-3, n)
                for _ in range(n):
                    ref = orc.fd_step(ref, 1e-3)
                done += n
                np.testing.assert_array_equal(s.get_c(), ref, err_msg="kmax %d after %d steps" % (kmax, done))
            before = s.get_c()
            s.step(1e-3, 9)
            s.rollback()                      # the state before the LAST step, also after multi-step launches
            for _ in range(8):
                before = orc.fd_step(before, 1e-3)
            np.testing.assert_array_equal(s.get_c(), before)
    lib.pfk_set_tuning(3, 4)


def test_api_error_paths_return_status_codes(lib):
    """programmer errors come back as negative pf_status + message, never as exceptions / crashes across the C ABI"""
    assert lib.pf_destroy(None) == 0
    with PhaseFieldSolver(dim=2, n=64, h=1.0) as s:
        h = s._h
        buf = np.zeros(10)
        assert lib.pf_set_field(h, L.PF_FIELD_C, buf.ctypes.data_as(C.c_void_p), 10) == L.PF_ERR_INVALID
        assert b"element count" in lib.pf_last_error(h)
        assert lib.pf_get_field(h, L.PF_FIELD_MU, buf.ctypes.data_as(C.c_void_p), 10) == L.PF_ERR_UNSUPPORTED
        assert lib.pf_rollback(h) == L.PF_ERR_STATE
        assert lib.pf_step(h, C.c_double(-1.0), 1, None) == L.PF_ERR_INVALID
        assert lib.pf_step(h, C.c_double(1e-3), 0, None) == 0
        assert lib.pf_step_begin(h, C.c_double(1e-3)) == L.PF_ERR_STATE          # not a slab handle
        assert lib.pf_dist_begin(h, L.PF_DIST_OP_STEP, C.c_double(1e-3)) == L.PF_ERR_STATE
        req = L.PfDistRequest()
        assert lib.pf_dist_advance(h, C.byref(req)) == L.PF_ERR_STATE
        lay = L.PfHaloLayout()
        assert lib.pf_halo_layout_get(h, C.byref(lay)) == L.PF_ERR_STATE
        assert lib.pfk_set_tuning(3, 3) == L.PF_ERR_INVALID and lib.pfk_set_tuning(99, 1) == L.PF_ERR_INVALID
    e = HipSlabEngine((64, 16, 8), 1.0, 2, 0, 0)
    out = (C.c_double * 3)()
    assert lib.pf_step(e._h, C.c_double(1e-3), 1, None) == L.PF_ERR_STATE          # slab handles step via begin/finish
    assert lib.pf_diagnostics(e._h, out) == L.PF_ERR_STATE
    assert lib.pf_step_finish(e._h) == L.PF_ERR_STATE
    assert lib.pf_step_begin(e._h, C.c_double(1e-3)) == 0
    assert lib.pf_step_begin(e._h, C.c_double(1e-3)) == L.PF_ERR_STATE            # already open
    assert lib.pf_diagnostics_local(e._h, out) == L.PF_ERR_STATE
    assert lib.pf_step_finish(e._h) == 0
    e.close()
    cfg = L.default_config(3, 30, 1.0)      # slab FFT modes need divisibility
    cfg.nranks, cfg.rank, cfg.scheme = 4, 0, L.PF_SCHEME_SPECTRAL_SI
    hh = C.c_void_p()
    assert lib.pf_create(C.byref(cfg), C.byref(hh)) == L.PF_ERR_UNSUPPORTED and not hh


def test_nccl_path_single_rank(lib):
    """the production multi-GPU host path (RCCL through torch.distributed) with world size 1 on the one GPU"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "nccl_single_rank_worker.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "NCCL_SINGLE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("world,bc,mode", [(2, "periodic", "split"), (3, "periodic", "split"), (3, "mirror", "split"),
                                           (2, "periodic", "fused"), (3, "periodic", "fused"), (3, "mirror", "fused"),
                                           (2, "periodic", "p2p"), (3, "mirror", "p2p"),
                                           (2, "periodic", "p2p_wide"), (3, "mirror", "p2p_wide"),
                                           (3, "periodic", "split_wide"), (3, "mirror", "split_wide")])
def test_multi_process_slabs_on_one_gpu_with_the_peer_copy_transport(lib, orc, world, bc, mode, tmp_path):
    """2 and 3 ranks as separate processes sharing the GPU: HipSlabEngine + SlabSolver(transport="ipc") -- ghost planes
    pushed into the neighbour's buffer through CUDA IPC, flag-ordered (pfk_push_planes / pfk_wait_flag); must equal the
    whole-domain oracle bit for bit (27 steps, so every buffer parity and sequence number is exercised).
    mode "fused": one launch per step, the boundary-strip workgroups poll the arrival flags inside the kernel
    (pf_step_slab_fused).  mode "p2p": the DEFAULT exchange code (batch_isend_irecv of the GPU ghost planes, the path
    RCCL serves on a multi-GPU node) over gloo -- two ranks, so both neighbours are the same peer and the message
    order matters.  mode "p2p_wide": the same with PF_FLAG_WIDE_HALO engines (4 ghost planes every second step)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    out = str(tmp_path / "res.npz")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "ipc_slab_worker.py"), out, bc, mode],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
    res = np.load(out)
    c = res["full"]
    nz, ny, nx = c.shape
    mirror = bc == "mirror"
    e = orc.even_extend(c) if mirror else c
    F0, C0, _ = orc.diagnostics(e, h=1.0, mirror=mirror)
    np.testing.assert_allclose(res["d0"][:2], [F0, C0], rtol=1e-13)
    for _ in range(25):
        e = orc.fd_step(e, 1e-3)
    F1, C1, _ = orc.diagnostics(e, h=1.0, mirror=mirror)
    np.testing.assert_allclose(res["d1"][:2], [F1, C1], rtol=1e-13)
    for _ in range(2):
        e = orc.fd_step(e, 1e-3)
    np.testing.assert_array_equal(res["field"], e[:nz, :ny, :nx])


@pytest.mark.parametrize("mode", ["spectral", "bm6", "bm6_elim", "spectral_mirror", "bm6_mirror"])
def test_multi_process_fft_slab_modes_on_one_gpu(lib, mode):
    """the slab-FFT modes with TWO ranks as separate processes sharing the GPU: real engines, the library's request
    protocol (pf_dist_begin / pf_dist_advance) served by gloo collectives on the GPU tensors (all_to_all_single + ghost
    exchange); rank 0 compares the gathered field with the single-domain solver (1e-12)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "fft_slab_gpu_worker.py"), mode],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs) and "FFT_SLAB_GPU_OK" in logs[0], "\n".join(l[-2000:] for l in logs)


def test_multi_process_slabs_unequal_planes_per_rank(lib, orc, tmp_path):
    """20 planes over 3 ranks (7 + 7 + 6): the neighbours' buffers have different sizes and time-level offsets, which
    the peer-copy transport must take from the neighbour's own layout (fused single-launch step)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    out = str(tmp_path / "res")
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PF_TEST_NZ="20")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "ipc_slab_worker.py"), out,
                                       "periodic", "fused"], env=env, cwd=root, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
    parts = [np.load(out + ".rank%d.npz" % r) for r in range(3)]
    e = parts[0]["full"]
    assert [int(p["z0"]) for p in parts] == [0, 7, 14] and [p["local"].shape[0] for p in parts] == [7, 7, 6]
    for _ in range(27):
        e = orc.fd_step(e, 1e-3)
    for p in parts:
        z0 = int(p["z0"])
        np.testing.assert_array_equal(p["local"], e[z0:z0 + p["local"].shape[0]])


@pytest.mark.parametrize("world,workload,grid,scaling", [
    (2, "bm1_fd_512c", [512, 512, 1024], "weak"),          # the driver's default series (weak: 512^3 per rank)
    (4, "bm1_fd_1024c", [1024, 1024, 1024], "strong"),     # BASELINE.json config 4: 1024^3 split into z-slabs
    (2, "bm6_fd_256c", [256, 256, 512], "weak"),           # BM6: slab-FFT Poisson (2 all-to-alls) + ghost exchange of c, phi
    (2, "bm1_spectral_512c", [512, 512, 512], "strong"),   # FFT modes at N > 1: ONE 512^3 box over the ranks (BASELINE config 5's form)
    (2, "bm3_fd_512c", [512, 512, 1024], "weak"),          # multi-field slabs: in-place ghost exchange overlapped with the interior planes
])
def test_bench_multi_rank_launch_contract_rehearsal(world, workload, grid, scaling):
    """The driver's N > 1 command line (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) on the
    1-GPU box: all ranks on device 0 over gloo (PFHIP_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device; at most 6
    processes may share the card, so the 8-rank case is rehearsed on CPU in tests/test_dist_gloo.py).
    Checks the launch contract only: env parsing, one JSON line from rank 0, whole-job value, grid and scaling mode."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, PFHIP_BENCH_REHEARSAL="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", str(world), "--workload", workload, "--steps", "4", "--warmup", "2",
                        "--preheat-s", "0.05", "--repeats", "2", "--verbose"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == scaling
    assert d["config"]["grid"] == grid and "REHEARSAL" in d["config"]["parallelism"]
    cells = grid[0] * grid[1] * grid[2]
    assert abs(d["value"] - cells * 4 / (d["ms_per_step"] * 1e-3 * 4)) <= 1e-6 * d["value"]
    if not workload.startswith("bm3"):                      # (BM3's second column is the solid fraction: it grows)
        assert d["check"]["C_rel_drift"] < 1e-12
    assert d["check"]["F_after"] < d["check"]["F_before"]
    assert "cpu_baseline" not in d and d["roofline"]["traffic"] is None
    assert d["repeats"] == 2 and len(d["block_ms_per_step"]) == 2 and d["preheat_steps"] >= 5
    if workload.startswith("bm1_spectral") or workload.startswith("bm6"):      # the line says which transform path ran
        assert "slab" in d["config"]["status"] and "hand-written LDS-FFT passes" in d["config"]["status"], d["config"]["status"]
    # the process group's own account of who ran: N ranks, one entry (rank, device, pid) each, distinct processes
    rk = d["ranks"]
    assert rk["world_size"] == world and rk["backend"] == "gloo"
    assert sorted(r["rank"] for r in rk["ranks"]) == list(range(world)) and len({r["pid"] for r in rk["ranks"]}) == world


def test_bench_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """One command, like the reference's `mpirun -np N python dolfin/bench1.py` (README.md:22): `python bench.py --gpus 2`
    with no WORLD_SIZE in the environment starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a child
    process (before this process touches the GPU), relays rank 0's JSON line and returns the child's exit code.  Rehearsal
    mode (both ranks on device 0 over gloo)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PFHIP_BENCH_REHEARSAL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--preheat-s", "0.05", "--repeats", "2", "--no-cpu-baseline"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "torch.distributed.run" in p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["config"]["grid"] == [512, 512, 1024]
    # a mismatch that is NOT "no launcher" is still an error (exit code != 0, nothing measured)
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env2, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr


@pytest.mark.parametrize("workload", ["bm1_fem_be", "bm2_fem_be"])
def test_bench_fem_be_workloads_standalone(workload):
    """`python bench.py --workload bm1_fem_be` (BASELINE.json config 1) on its own -- the round-2 form crashed on its
    warm-up default -- and the BM2 counterpart: one JSON line, rc 0, node-updates/s on the committed time grid."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["unit"] == "node-updates/s" and d["steps"] == 4 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"]["workload"] == workload and d["config"]["newton_iterations"] >= 4
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "bm1_fem_be"], cwd=root,
                       capture_output=True, text=True, timeout=600) if workload == "bm1_fem_be" else None
    if p is not None:       # the documented defaults (100 steps are capped by the 73-row grid; warm-up 10)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
        assert d["warmup"] == 10 and d["steps"] == 63 and d["cpu_baseline"]["kind"] == "port" and d["preheat_ms"] > 0


ALSO_KEYS = ("bm2_fem_be", "bm3_fem_be", "bm1_fem_be", "bm2_fd_512c", "bm3_fd_512c", "bm6_fd_512c_elim", "bm1_fd_512s",
             "bm1_spectral_512s", "bm6_fd_512c", "bm1_fd_1024c", "bm1_spectral_512c")


def test_bench_contract_json_line():
    """bench.py prints exactly one JSON line with the fields the driver reads (short run of the default workload), and the
    line is COMPACT: the driver keeps the last 8 KB of stdout, round 3's 15 KB line lost five of its eleven side
    measurements that way -- <= 6000 characters, every `also` entry recoverable from the last 6000 characters of stdout,
    the BASELINE-config entries last."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    assert len(lines[0]) <= 6000, len(lines[0])
    tail = p.stdout[-6000:]
    d = json.loads(tail[tail.index("{"):])                 # the whole line sits inside the tail a reader keeps
    assert tuple(d["also"].keys()) == ALSO_KEYS            # ... with the BASELINE-config lines at its end
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "cell-updates/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "f64"
    assert d["config"]["workload"] == "bm1_fd_512c" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["config"]["grid"] == [512, 512, 512]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0.2 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 2.0e9          # HBM bytes per launch >= algorithmic 2.15 GB
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "cell-updates/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert abs(d["value"] - 512 ** 3 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-5 * d["value"]
    assert d["check"]["C_rel_drift"] < 1e-12
    # timed-region bookkeeping: declared pre-heat, repeated K-step blocks (the blocks themselves are --verbose only)
    assert d["preheat_ms"] >= 500.0 and d["repeats"] == 25 and "block_ms_per_step" not in d
    # every side entry: value, time per step, steps, steady, roofline {frac, achieved, traffic, bytes}, check, workload
    for k in ALSO_KEYS:
        e = d["also"][k]
        assert e["value"] > 0 and e["steps"] > 0 and ("ms_per_step" in e or "us_per_step" in e) and e["config"]["workload"] == k
        assert "check" in e
        if not k.endswith("_fem_be"):
            assert set(e["roofline"]) == {"achieved", "frac", "traffic", "bytes_per_cell_update"} and "steady" in e
    # north_star's target configuration and the reference's own algorithm ride in the same line
    big = d["also"]["bm1_fd_1024c"]
    assert 0.2 < big["roofline"]["frac"] < 1.0
    assert abs(big["value"] - 1024 ** 3 / (big["ms_per_step"] * 1e-3)) < 1e-5 * big["value"]
    assert big["check"]["C_rel_drift"] < 1e-12 and big["check"]["F_after"] < big["check"]["F_before"]
    sp3 = d["also"]["bm1_spectral_512c"]
    assert sp3["roofline"]["bytes_per_cell_update"] == 72.0
    assert 0.1 < sp3["roofline"]["frac"] < 1.0 and sp3["check"]["C_rel_drift"] < 1e-12
    fb = d["also"]["bm1_fem_be"]
    assert fb["unit"] == "node-updates/s" and fb["cpu_baseline"]["kind"] == "port"
    assert abs(fb["check"]["F"] - 190.1699) < 1e-3      # row t = 11.1 of the reference's bench1_out.csv
    # BASELINE.json config 5 (BM6 at 512^3, one GPU's share) in both forms, and the two extra models of SURVEY 8f next-4
    b6, b6e = d["also"]["bm6_fd_512c"], d["also"]["bm6_fd_512c_elim"]
    assert b6["roofline"]["bytes_per_cell_update"] == 72.0 and b6e["roofline"]["bytes_per_cell_update"] == 16.0
    assert 0.05 < b6["roofline"]["frac"] < 1.0 and 0.2 < b6e["roofline"]["frac"] < 1.0
    assert b6["check"]["C_rel_drift"] < 1e-12 and b6e["check"]["C_rel_drift"] < 1e-12
    for m, nodes in (("bm2", 20201), ("bm3", 245701)):
        fm = d["also"]["%s_fem_be" % m]
        assert fm["unit"] == "node-updates/s" and abs(fm["value"] - nodes / (fm["ms_per_step"] * 1e-3)) < 1e-5 * fm["value"]
        assert fm["cpu_baseline"]["kind"] == "port-extrapolated" and fm["cpu_baseline"]["value"] > 0
    assert abs(d["also"]["bm2_fem_be"]["check"]["F"] - 3621.6143739566) < 1e-5       # row t = 0.63 of bench2_out.csv
    for nm, bpc in (("bm2_fd_512c", 80.0), ("bm3_fd_512c", 32.0)):                   # the same models on the stencil design
        fd = d["also"][nm]
        assert fd["roofline"]["bytes_per_cell_update"] == bpc and 0.3 < fd["roofline"]["frac"] < 1.0
    assert abs(r["frac"] - d["value"] * 16.0 / 1e9 / 8000.0) < 1e-6                  # the wall-clock figure IS the headline
    assert r["frac_hip_events"] >= r["frac"] * 0.98 and d["ranks"]["world_size"] == 1


def test_bench_verbose_line_keeps_the_bookkeeping():
    """--verbose restores what the compact line drops: every timed block (the reported one is the median), the pre-heat step
    count, pf_status_string, the spectral scheme's field-store mode, notes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "bm1_spectral_256c", "--steps", "4",
                        "--warmup", "1", "--verbose", "--no-cpu-baseline", "--repeats", "5"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0])
    assert d["repeats"] == len(d["block_ms_per_step"]) == 5 and sorted(d["block_ms_per_step"])[2] == d["ms_per_step"]
    assert d["preheat_ms"] >= 500.0 and d["preheat_steps"] > 100
    assert d["config"]["field_store"] == "last two steps of each pf_step call" and d["config"]["status"]
    assert d["config"]["grid"] == [256, 256, 256] and "traffic_source" in d["roofline"]


def test_b13d_driver_3d_extrusion_invariants(lib, tmp_path):
    """3-D BM1 driver (dolfin/b13d.py semantics): z-extruded data -> F_3D = L_z F_2D and C_3D = L_z C_2D at every row
    (SURVEY a14), with the 2-D run at the same h and dt."""
    from pfhubbenchmarks_amd.drivers import run_b13d, run_bench1
    r3, _ = run_b13d(intervals=50, end_time=1.0, dt=0.005, out_dir=str(tmp_path), verbose=False)
    r2, _ = run_bench1(intervals=50, L=100.0, scheme="fd", dt=0.005, end_time=1.0, out_dir=str(tmp_path / "two"),
                       verbose=False)
    assert r3.shape == r2.shape == (4, 3)
    np.testing.assert_allclose(r3[:, 1], 100.0 * r2[:, 1], rtol=1e-12)
    np.testing.assert_allclose(r3[:, 2], 100.0 * r2[:, 2], rtol=1e-12)


def test_fused_kernel_fuzz_shapes_ranges_phi_with_nan_guards(lib, orc):
    """Randomised shapes / plane ranges / ghost mode / phi coupling for the stateless kernel entry, bit-compared with
    the oracle.  Every device buffer sits between NaN-filled guard bands and the outputs outside the requested plane
    range must stay untouched: an out-of-range read that matters, or a stray write, shows up as a mismatch."""
    rng = np.random.default_rng(2024)
    guard = 4096
    for case in range(48):
        nx = 2 * int(rng.integers(1, 150))
        ny = int(rng.integers(1, 40))
        nz = int(rng.integers(1, 12))
        slab = bool(rng.integers(0, 2)) and nz >= 1
        ghost, zwrap = (2, 0) if slab else (0, 1)
        zlo = int(rng.integers(0, nz))
        zhi = int(rng.integers(zlo + 1, nz + 1))
        with_phi = bool(rng.integers(0, 2))
        variant = int(rng.choice([0, 3, 6]))
        shape = (nz + 2 * ghost, ny, nx)
        c = 0.5 + 0.1 * rng.standard_normal(shape)
        phi = rng.standard_normal(shape) if with_phi else None
        sentinel = -7.25
        ref = np.full(shape, sentinel)
        orc.fd_step(c, 1e-3, phi=phi, k_phi=0.09 if with_phi else 0.0, ghost=ghost, zwrap=zwrap, zlo=zlo, zhi=zhi, out=ref)

        def guarded(a, fill=np.nan):
            t = torch.full((a.size + 2 * guard,), fill, dtype=torch.float64, device="cuda")
            t[guard:guard + a.size] = torch.from_numpy(np.ascontiguousarray(a).ravel()).cuda()
            return t
        tc = guarded(c)
        tp = guarded(phi) if with_phi else None
        to = guarded(np.full(shape, sentinel), fill=sentinel)
        p = L.PfkChParams(0.3, 0.7, 10.0, 2.0, 1e-3 * 5.0, 0.09 if with_phi else 0.0)
        assert lib.pfk_set_tuning(0, variant) == 0
        rc = lib.pfk_ch_fd_step(C.c_void_p(tc.data_ptr() + 8 * guard), C.c_void_p(to.data_ptr() + 8 * guard),
                                C.c_void_p(tp.data_ptr() + 8 * guard) if with_phi else None, nx, ny, nz, ghost, zwrap,
                                zlo, zhi, C.byref(p), L.PF_KERNEL_FUSED, None)
        assert rc == 0, lib.pf_last_error(None)
        torch.cuda.synchronize()
        got = to.cpu().numpy()
        msg = "case %d: nx %d ny %d nz %d slab %s z [%d,%d) phi %s variant %d" % (case, nx, ny, nz, slab, zlo, zhi,
                                                                                  with_phi, variant)
        np.testing.assert_array_equal(got[guard:-guard].reshape(shape), ref, err_msg=msg)
        assert np.all(got[:guard] == sentinel) and np.all(got[-guard:] == sentinel), msg   # no stray writes
    lib.pfk_set_tuning(0, 0)


def test_diagnostics_fuzz(lib, orc):
    """diagnostics reduction (streaming kernel for even nx, generic kernel for odd nx) on random 2-D / 3-D shapes,
    periodic and mirror, against the oracle's compensated sums"""
    rng = np.random.default_rng(77)
    for case in range(24):
        dim = int(rng.integers(2, 4))
        n = [int(rng.integers(2, 140)), int(rng.integers(1, 30)), int(rng.integers(1, 9))][:dim]
        mirror = bool(rng.integers(0, 2)) and all(v >= 2 for v in n)
        if not mirror and n[0] % 2 == 1 and rng.integers(0, 2):
            n[0] += 1
        shape = tuple(reversed(n))
        c = 0.5 + 0.1 * rng.standard_normal(shape)
        kern = "auto"
        with PhaseFieldSolver(dim=dim, n=tuple(n), h=0.7, bc="mirror" if mirror else "periodic", kernel=kern) as s:
            s.set_c(c)
            F, Ctot, _ = s.diagnostics()
        lat = orc.even_extend(c) if mirror else c
        Fo, Co, _ = orc.diagnostics(lat, h=0.7, mirror=mirror)
        assert abs(F - Fo) <= 2e-13 * abs(Fo) and abs(Ctot - Co) <= 2e-13 * abs(Co), (case, n, mirror, F, Fo)


@pytest.mark.parametrize("shape", [(12, 20, 128), (96, 256)])
def test_bm6_phi_elimination(lib, shape):
    """PF_FLAG_BM6_ELIMINATE_PHI (periodic box): lap_h(k phi) = -(k^2/eps)(c - mean c) exactly, so the step needs no Poisson
    solve.  (a) bit-exact against the oracle restating the same step when both use the same mean; (b) equal to the
    explicit-phi path up to rounding; (c) phi / f_elec still available for diagnostics."""
    from oracle import bm6_fd
    rng = np.random.default_rng(sum(shape))
    c = 0.5 + 0.04 * rng.standard_normal(shape)
    dim, n = len(shape), shape[::-1]
    cbar = float(np.mean(c))
    o_elim = bm6_fd.BM6FD(c, 1.0, eliminate_phi=True, cbar=cbar)
    o_phi = bm6_fd.BM6FD(c, 1.0)
    with PhaseFieldSolver(dim=dim, n=n, h=1.0, model="bm6", eliminate_phi=True) as s:
        s.set_c(c)
        _lib_check = L.check(lib.pf_set_mean_c(s._h, C.c_double(cbar)), s._h)
        s.step(5e-4, 9)
        o_elim.step(5e-4, 9)
        np.testing.assert_array_equal(s.get_c(), o_elim.c)                    # (a)
        o_phi.step(5e-4, 9)
        assert np.abs(s.get_c() - o_phi.c).max() <= 1e-12                    # (b)
        F, Ctot, E = s.diagnostics()                                         # (c) Poisson solve on demand
        Fo, Co, Eo = o_elim.diagnostics()
        assert abs(F - Fo) <= 1e-12 * abs(Fo) and abs(E - Eo) <= 1e-10 * abs(Eo) and E != 0.0
        assert np.abs(s.get_phi() - o_elim.phi()).max() <= 1e-12
    with PhaseFieldSolver(dim=dim, n=n, h=1.0, model="bm6", eliminate_phi=True) as s:   # mean computed by the library
        s.set_c(c)
        s.step(5e-4, 9)
        assert np.abs(s.get_c() - o_elim.c).max() <= 1e-14
    with pytest.raises(L.PfhipError):                                         # only meaningful for the periodic box
        PhaseFieldSolver(dim=2, n=33, h=1.0, bc="mirror", model="bm6", eliminate_phi=True)
