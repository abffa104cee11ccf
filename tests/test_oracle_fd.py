"""Properties of the CPU restatement of the FD scheme (oracle/ch_fd.c) and its link to the FEM oracle."""
import numpy as np

from oracle import fem_be


def test_mass_conservation_and_energy_decay(orc):
    c = orc.ic(96, 64, 1)[0]
    F0, C0, _ = orc.diagnostics(c)
    Fs = [F0]
    for _ in range(200):
        c = orc.fd_step(c, 2e-3)
        Fs.append(orc.diagnostics(c)[0])
    F1, C1, _ = orc.diagnostics(c)
    assert abs(C1 - C0) / C0 < 1e-13
    assert all(b <= a + 1e-12 * abs(a) for a, b in zip(Fs, Fs[1:]))


def test_extruded_3d_equals_2d_bitwise(orc):
    c2 = orc.ic(64, 48, 1)[0]
    c3 = np.repeat(c2[None], 6, 0)
    for _ in range(5):
        c2 = orc.fd_step(c2, 1e-3)
        c3 = orc.fd_step(c3, 1e-3)
    for z in range(6):
        np.testing.assert_array_equal(c3[z], c2)


def test_slab_with_ghosts_equals_whole(orc):
    rng = np.random.default_rng(7)
    c = 0.5 + 0.1 * rng.standard_normal((12, 10, 16))
    whole = orc.fd_step(c, 1e-3)
    for z0, nzl in [(0, 6), (6, 6), (3, 5)]:
        idx = np.arange(z0 - 2, z0 + nzl + 2) % 12
        slab = np.ascontiguousarray(c[idx])
        out = orc.fd_step(slab, 1e-3, ghost=2, zwrap=0)
        np.testing.assert_array_equal(out[2:2 + nzl], whole[z0:z0 + nzl])
        # split into interior + boundary launches like pf_step_begin / pf_step_finish
        o2 = np.zeros_like(slab)
        orc.fd_step(slab, 1e-3, ghost=2, zwrap=0, zlo=2, zhi=nzl - 2, out=o2)
        orc.fd_step(slab, 1e-3, ghost=2, zwrap=0, zlo=0, zhi=2, out=o2)
        orc.fd_step(slab, 1e-3, ghost=2, zwrap=0, zlo=nzl - 2, zhi=nzl, out=o2)
        np.testing.assert_array_equal(o2[2:2 + nzl], whole[z0:z0 + nzl])


def test_mirror_extension_keeps_symmetry_and_halves_integrals(orc):
    n = 21
    x = np.arange(n) * 2.0
    c = fem_be.ic_bm1(x[None, :], x[:, None])
    e = orc.even_extend(c)
    assert e.shape == (40, 40)
    for _ in range(20):
        e = orc.fd_step(e, 0.02, h=2.0)
    np.testing.assert_array_equal(e[1:20, :], e[39:20:-1, :])     # mirror symmetry is preserved bitwise
    np.testing.assert_array_equal(e[:, 1:20], e[:, 39:20:-1])
    F, C, _ = orc.diagnostics(e, h=2.0, mirror=True)
    w = np.ones(n)
    w[0] = w[-1] = 0.5
    Ctrap = 4.0 * np.einsum("i,j,ij", w, w, e[:n, :n])            # trapezoid rule on the physical nodes
    assert abs(C - Ctrap) / Ctrap < 1e-13


def test_fd_scheme_tracks_the_reference_algorithm_at_early_time(orc, golden_dir):
    """Physical cross-check, NOT the parity gate: the explicit FD scheme (h = 2, dt = 0.01, even extension of the
    200 x 200 no-flux domain) and the reference's FEM backward-Euler trajectory (fixture row t = 0.1) are two
    discretisations of the same PDE; at t = 0.1 they agree to ~1e-3 in F (-6.6e-4 for a
    converged solution, the rest is h = 2 spatial error)."""
    import os
    csv = np.loadtxt(os.path.join(golden_dir, "bench1_out.csv"), delimiter=",", skiprows=1)
    n = 101
    x = np.arange(n) * 2.0
    e = orc.even_extend(fem_be.ic_bm1(x[None, :], x[:, None]))
    for _ in range(10):
        e = orc.fd_step(e, 0.01, h=2.0)
    F, C, _ = orc.diagnostics(e, h=2.0, mirror=True)
    assert abs(F - csv[0, 1]) / csv[0, 1] < 3e-3
    assert abs(C - csv[0, 2]) / csv[0, 2] < 1e-4      # trapezoid vs P1 mass functional


def test_spectral_oracle_conserves_mass_and_matches_fd_for_smooth_data():
    """The two GPU schemes' oracles agree with each other on smooth periodic data (O(h^2) apart)."""
    from oracle import ch_fd, ch_spectral
    n = 64
    x = np.arange(n) * (2 * np.pi / n)
    c = 0.5 + 0.05 * np.cos(x)[None, :] * np.cos(2 * x)[:, None]
    h = 1.0
    sp = ch_spectral.SpectralCH(c, h=h)
    F0, C0 = sp.diagnostics()
    sp.step(1e-3, 50)
    F1, C1 = sp.diagnostics()
    assert abs(C1 - C0) < 1e-12 * abs(C0) and F1 < F0
    ch_fd.load()
    cf = c.copy()
    for _ in range(50):
        cf = ch_fd.fd_step(cf, 1e-3, h=h)
    assert np.abs(cf - sp.c).max() < 2e-4          # FD's O(h^2) dispersion error on k = 1, 2 modes at n = 64
    Ff, Cf, _ = ch_fd.diagnostics(cf, h=h)
    assert abs(Ff - F1) / F1 < 1e-3


def test_bm6_spectral_oracle_agrees_with_the_fd_phi_eliminated_oracle():
    """two restatements of BM6 in a periodic box -- phi eliminated in Fourier space (oracle/ch_spectral.py) and through
    the discrete Laplacian (oracle/bm6_fd.py) -- solve the same PDE: on a smooth field they differ by the spatial
    discretisation error only, and both conserve mass"""
    from oracle import bm6_fd, ch_spectral
    n = 64
    x = np.arange(n)
    c = 0.5 + 0.04 * np.cos(2 * np.pi * x[None, :] / n) * np.cos(4 * np.pi * x[:, None] / n)
    a = ch_spectral.SpectralCH(c, h=1.0, bm6=True)
    b = bm6_fd.BM6FD(c, 1.0, eliminate_phi=True)
    a.step(1e-4, 200)
    b.step(1e-4, 200)
    assert np.abs(a.c - b.c).max() < 2e-6
    Fa, Ca, Ea = a.diagnostics()
    Fb, Cb, Eb = b.diagnostics()
    assert abs(Ca - Cb) < 1e-9 and abs(Fa - Fb) < 1e-4 * abs(Fb) and abs(Ea - Eb) < 1e-2 * abs(Eb)


def test_oracle_under_asan_ubsan():
    """SURVEY.md section 5 (sanitizers on the CPU build): oracle/sanitize_check.c drives every entry point of
    oracle/ch_fd.c over the edge shapes on exactly-sized heap buffers, built with -fsanitize=address,undefined; it must
    run clean and print the same checksum as the plain build of the same driver."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    a = subprocess.run([os.path.join(root, "oracle", "sanitize_check_asan")], capture_output=True, text=True, env=env,
                       timeout=300)
    p = subprocess.run([os.path.join(root, "oracle", "sanitize_check_plain")], capture_output=True, text=True,
                       timeout=300)
    assert a.returncode == 0 and a.stderr.strip() == "", a.stderr[-3000:]
    assert p.returncode == 0
    assert a.stdout.startswith("SANITIZE_OK cases=121") and a.stdout == p.stdout, (a.stdout, p.stdout)
