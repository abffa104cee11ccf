"""CPU suite: the numpy restatement of the explicit multi-field FD schemes (oracle/multi_fd.py, PFHub BM2 / BM3) --
conservation, dissipation, fixed points, symmetry, and its initial energy against the FEM functional of the reference
(oracle/fem_multi.py, which is pinned to the reference's committed CSVs)."""
import numpy as np
import pytest

from oracle import fem_multi, multi_fd


def test_bm2_conserves_solute_and_dissipates_energy():
    u = multi_fd.ic_bm2(64, 48, 2.0)
    h, dt = 2.0, 5e-3
    F0, C0 = multi_fd.diagnostics("bm2", u, h, 2)
    Fs = [F0]
    for _ in range(40):
        u = multi_fd.bm2_step(u, dt, h)
        Fs.append(multi_fd.diagnostics("bm2", u, h, 2)[0])
    F1, C1 = multi_fd.diagnostics("bm2", u, h, 2)
    assert abs(C1 - C0) <= 1e-13 * abs(C0)                 # c_t = M lap_h(mu): the lattice sum of lap_h is zero
    assert all(b < a for a, b in zip(Fs, Fs[1:]))          # gradient flow of the discrete energy, dt below the limit
    assert F1 < 0.9 * F0


def test_bm3_fixed_points_and_growth():
    h = 960.0 / 350
    liquid = np.stack([np.full((1, 8, 8), -0.3), np.full((1, 8, 8), -1.0)])
    np.testing.assert_array_equal(multi_fd.bm3_step(liquid, 0.05, h), liquid)      # dfdp(phi = -1) = 0, lap = 0
    u = multi_fd.even_extend(multi_fd.ic_bm3(40, 40, h))
    dom = (39 * h) ** 2
    S = [multi_fd.diagnostics("bm3", u, h, 2, mirror=True, domain=dom)[1]]
    for _ in range(60):
        u = multi_fd.bm3_step(u, 0.05, h)
    S.append(multi_fd.diagnostics("bm3", u, h, 2, mirror=True, domain=dom)[1])
    assert S[1] > S[0] > 0.0                               # the undercooled melt solidifies: the seed grows
    assert u[1].max() <= 1.0 + 1e-6 and u[1].min() >= -1.0 - 1e-6


@pytest.mark.parametrize("model", ["bm2", "bm3"])
def test_even_extension_is_preserved_bitwise(model):
    """the no-flux boxes of the reference run on their even extension: the symmetric lattice stays symmetric bit for bit
    (the stencil sums are commutative pairwise), so restricting to the nodes is exact"""
    n, h = 17, 3.0
    ic = (multi_fd.ic_bm2 if model == "bm2" else multi_fd.ic_bm3)(n, n, h)
    ext = multi_fd.even_extend(ic)
    assert ext.shape[-2:] == (2 * (n - 1), 2 * (n - 1))
    step = multi_fd.bm2_step if model == "bm2" else multi_fd.bm3_step
    for _ in range(6):
        ext = step(ext, 2e-3 if model == "bm2" else 2e-2, h)
    nodes = ext[..., :n, :n]
    np.testing.assert_array_equal(multi_fd.even_extend(nodes), ext)


def test_bm2_fd_energy_of_the_initial_condition_is_the_fem_functional_up_to_h2():
    """same physics as the pinned FEM oracle: the FD functional (trapezoid rule on the even extension, forward
    differences) and the P1 / Strang-Fix functional of the same interpolated IC differ by the discretisation error only"""
    o = fem_multi.MultiFieldBE("bm2")
    F_fem, C_fem = o.diagnostics()
    Fs = []
    for N in (100, 200, 400):
        h = 200.0 / N
        u = multi_fd.even_extend(multi_fd.ic_bm2(N + 1, N + 1, h))
        F, C = multi_fd.diagnostics("bm2", u, h, 2, mirror=True)
        Fs.append(F)
        assert abs(C - C_fem) <= 1e-5 * C_fem
    # the FD functional converges (6531.96, 6533.04, 6533.31 -> 6533.4); the P1 value on the reference's h = 2 mesh
    # (6514.19) is 2.9e-3 below that limit: its own spatial error
    assert abs(Fs[2] - Fs[1]) < 0.3 * abs(Fs[1] - Fs[0])
    assert 2e-3 * F_fem < Fs[2] - F_fem < 4e-3 * F_fem
