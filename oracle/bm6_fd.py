"""ORACLE (test infrastructure, not product code) -- CPU restatement of the BM6 (Cahn-Hilliard + Poisson) explicit
FD path of libpfhip: Poisson solve by numpy FFT with the 5/7-point Laplacian's eigenvalues (csrc/poisson.hip), then the
C oracle's FD step with the k*phi coupling (oracle/ch_fd.c).

Reference physics: dolfin/bench6.py:61-74 (f_elec = k c phi / 2, dfdc += k phi, lap phi = -k c / eps),
:77-90 (phi = 0 on x = 0, phi = sin(y/7) on x = Lx, no-flux elsewhere), pfbase.py:410-421.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

from . import ch_fd


def _lam(shape, h):
    lam = np.zeros([s if i < len(shape) - 1 else s // 2 + 1 for i, s in enumerate(shape)])
    for ax, n in enumerate(shape):
        m = np.arange(n // 2 + 1) if ax == len(shape) - 1 else np.arange(n)
        sh = [1] * len(shape)
        sh[ax] = len(m)
        lam = lam + (2.0 * np.cos(2.0 * np.pi * m / n) - 2.0).reshape(sh)
    return lam / (h * h)


def poisson_periodic(c, h, k=0.09, eps=90.0):
    """zero-mean phi with lap_h phi = -(k/eps) (c - mean c) on a periodic lattice"""
    lam = _lam(c.shape, h)
    ch = np.fft.rfftn(c)
    with np.errstate(divide="ignore", invalid="ignore"):
        ph = np.where(lam != 0.0, -(k / eps) * ch / lam, 0.0)
    return np.fft.irfftn(ph, s=c.shape, axes=tuple(range(c.ndim)))


def poisson_dirichlet_x(c_ext, npx, npy, h, k=0.09, eps=90.0):
    """c_ext: even extension (2(npy-1), 2(npx-1)) of the no-flux domain.  Returns the EVEN-in-x extension of phi with
    phi = 0 on x = 0 and phi = sin(y/7) on x = Lx (same output as poisson_solve in csrc/poisson.hip).  2-D, or
    3-D (nz_ext, ny_ext, nx_ext) with z a no-flux axis like y (the boundary data depends on y only)."""
    ny, nx = c_ext.shape[-2:]
    N = npx - 1
    x = np.arange(nx)
    xr = np.where(x <= N, x, 2 * N - x)
    y = np.arange(ny)
    yr = np.where(y < npy, y, 2 * (npy - 1) - y)
    g = np.sin(yr * h / 7.0)
    r = -(k / eps) * c_ext.copy()
    r[..., xr == N - 1] -= (g / (h * h))[:, None]
    sgn = np.where((xr == 0) | (xr == N), 0.0, np.where(x > N, -1.0, 1.0))
    r = r * sgn
    lam = _lam(r.shape, h)
    rh = np.fft.rfftn(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        ph = np.where(lam != 0.0, rh / lam, 0.0)
    phi = np.fft.irfftn(ph, s=r.shape, axes=tuple(range(r.ndim)))
    out = phi * np.where(x > N, -1.0, 1.0)
    out[..., x == 0] = 0.0
    out[..., x == N] = g[:, None]
    return out


class BM6FD:
    def __init__(self, c, h, mirror_nodes=None, k=0.09, eps=90.0, eliminate_phi=False, cbar=None):
        """c: lattice array (2-D or 3-D).  mirror_nodes = (npx, npy) selects the reference's boundary conditions.
        eliminate_phi (periodic box only): lap_h(k phi) = -(k^2/eps)(c - mean c) exactly, so the step needs no Poisson
        solve: cnew += -dt M k^2/eps (c - cbar) (PF_FLAG_BM6_ELIMINATE_PHI); cbar defaults to the mean of the start field."""
        self.c = np.ascontiguousarray(c, dtype=np.float64)
        self.h, self.k, self.eps = h, k, eps
        self.mirror_nodes = mirror_nodes
        assert not (eliminate_phi and mirror_nodes is not None)
        self.eliminate_phi = eliminate_phi
        self.cbar = float(np.mean(self.c)) if cbar is None else float(cbar)

    def phi(self):
        if self.mirror_nodes is None:
            return poisson_periodic(self.c, self.h, self.k, self.eps)
        return poisson_dirichlet_x(self.c, self.mirror_nodes[0], self.mirror_nodes[1], self.h, self.k, self.eps)

    def step(self, dt, nsteps=1):
        for _ in range(nsteps):
            if self.eliminate_phi:
                c3 = self.c if self.c.ndim == 3 else self.c[None]
                out = ch_fd.fd_step(c3, dt, h=self.h, elim=(self.k, self.eps, self.cbar))
                self.c = out if self.c.ndim == 3 else out[0]
                continue
            phi = np.ascontiguousarray(self.phi())
            c3 = self.c if self.c.ndim == 3 else self.c[None]
            p3 = phi if phi.ndim == 3 else phi[None]
            out = ch_fd.fd_step(c3, dt, h=self.h, phi=p3, k_phi=self.k)
            self.c = out if self.c.ndim == 3 else out[0]
        return self.c

    def diagnostics(self):
        phi = np.ascontiguousarray(self.phi())
        c3 = self.c if self.c.ndim == 3 else self.c[None]
        p3 = phi if phi.ndim == 3 else phi[None]
        return ch_fd.diagnostics(c3, h=self.h, dim=self.c.ndim, phi=p3, k=self.k, mirror=self.mirror_nodes is not None)
