"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's FEM / backward-Euler
algorithm for PFHub BM1 and BM6.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates (paths relative to the reference tree):
  * dolfin/bench1.py:21-23   mesh: RectangleMesh(0,0 -> L,L, N, N, 'crossed'), L=200, N=100
  * dolfin/bench1.py:39-41   space: P1 x P1 (c, mu)              (bench6.py:42-46: P1 x P1 x P1 (c, mu, phi))
  * dolfin/bench1.py:14-16   quadrature_degree = 3  -> 6-point Strang-Fix rule for the nonlinear terms
  * dolfin/bench1.py:63-65   f_chem = rho_s (c-c_alpha)^2 (c_beta-c)^2, dfdc = d f_chem / dc
  * dolfin/pfbase.py:361-383 cahn_hilliard_weak_form (backward Euler, monolithic in c and mu)
  * dolfin/pfbase.py:410-421 poisson_weak_form;  dolfin/bench6.py:61-90 coupling k*phi, Dirichlet phi
  * dolfin/bench1.py:85-88   SNES: absolute_tolerance 1e-6 on ||R||_2 (we take plain full Newton steps)
  * dolfin/bench1.py:121-125 diagnostics total_solute / total_free_energy (bench6.py:155-165 adds f_elec)
  * dolfin/pfbase.py:177-193 / :322-339  initial conditions

The arithmetic of the reference lives in FEniCS/PETSc (not vendored, unpinned: README.md:10), which are absent
here; this restatement is pinned instead against the reference's committed outputs (tests/golden/):
results/bench1_out.csv (73 rows), results/bench6_out.csv (14 rows), results/bench1/conc00000{0-5}.vtu and
results/bench6/{conc,phi}00000{0-5}.vtu.  See tests/test_oracle_fem.py for the tolerances reached.

Time grid: taken from column 1 of the fixture CSV (the reference's controller, bench1.py:180-183, depends on
its inexact GMRES/SOR iteration counts and cannot be re-derived).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# Strang-Fix 6-point degree-3 rule on the reference triangle: all permutations of (a, b, c), weights 1/6.
_SF = (0.659027622374092, 0.231933368553031, 0.109039009072877)
_SF_PERMS = np.array([
    [_SF[0], _SF[1], _SF[2]], [_SF[0], _SF[2], _SF[1]],
    [_SF[1], _SF[0], _SF[2]], [_SF[1], _SF[2], _SF[0]],
    [_SF[2], _SF[0], _SF[1]], [_SF[2], _SF[1], _SF[0]],
])
_SF_W = np.full(6, 1.0 / 6.0)


class Params:
    """Model constants; defaults = dolfin/bench1.py:32-36 (BM6 adds k, eps_r: bench6.py:38-39)."""

    def __init__(self, c_alpha=0.3, c_beta=0.7, rho_s=5.0, kappa=2.0, M=5.0, k=0.09, eps_r=90.0):
        self.c_alpha, self.c_beta, self.rho_s, self.kappa, self.M = c_alpha, c_beta, rho_s, kappa, M
        self.k, self.eps_r = k, eps_r

    def f(self, c):
        return self.rho_s * (c - self.c_alpha) ** 2 * (self.c_beta - c) ** 2

    def df(self, c):
        a, b = c - self.c_alpha, self.c_beta - c
        return 2.0 * self.rho_s * a * b * (b - a)

    def d2f(self, c):
        a, b = c - self.c_alpha, self.c_beta - c
        return 2.0 * self.rho_s * (b * b - 4.0 * a * b + a * a)


def ic_bm1(x, y, c0=0.5, eps=0.05):
    """dolfin/pfbase.py:187-189 (amplitudes from dolfin/bench1.py:48-49)."""
    return c0 + eps * (np.cos(0.105 * x) * np.cos(0.11 * y)
                       + (np.cos(0.13 * x) * np.cos(0.087 * y)) ** 2
                       + np.cos(0.025 * x - 0.15 * y) * np.cos(0.07 * x - 0.02 * y))


def ic_bm6(x, y, c0=0.5, c1=0.04):
    """dolfin/pfbase.py:332-334 (amplitudes from dolfin/bench6.py:53-54)."""
    return c0 + c1 * (np.cos(0.2 * x) * np.cos(0.11 * y)
                      + (np.cos(0.13 * x) * np.cos(0.087 * y)) ** 2
                      + np.cos(0.025 * x - 0.15 * y) * np.cos(0.07 * x - 0.02 * y))


class CrossedMesh:
    """'crossed' RectangleMesh: corners numbered i + (N+1) j (x fastest), then cell centres (N+1)^2 + i + N j.
    Each square -> 4 triangles {corner, corner, centre}; order as in results/bench1/conc000000.vtu."""

    def __init__(self, L=200.0, N=100):
        self.L, self.N = float(L), int(N)
        h = self.h = self.L / self.N
        n1 = N + 1
        ii, jj = np.meshgrid(np.arange(n1), np.arange(n1), indexing="xy")  # jj varies along rows
        xc = (ii * h).ravel()
        yc = (jj * h).ravel()
        ci, cj = np.meshgrid(np.arange(N), np.arange(N), indexing="xy")
        xm = ((ci + 0.5) * h).ravel()
        ym = ((cj + 0.5) * h).ravel()
        self.x = np.concatenate([xc, xm])
        self.y = np.concatenate([yc, ym])
        self.n_corner = n1 * n1
        self.n = self.x.size
        sw = (ci + n1 * cj).ravel()            # south-west corner of each square
        se, nw, ne = sw + 1, sw + n1, sw + n1 + 1
        ctr = self.n_corner + (ci + N * cj).ravel()
        tri = np.stack([
            np.stack([sw, se, ctr], 1), np.stack([sw, nw, ctr], 1),
            np.stack([se, ne, ctr], 1), np.stack([nw, ne, ctr], 1)], 1)   # (N*N, 4, 3)
        self.tri = tri.reshape(-1, 3)
        self.ntri = self.tri.shape[0]
        self._assemble_linear()

    def _assemble_linear(self):
        t = self.tri
        x, y = self.x[t], self.y[t]                       # (ntri, 3)
        # gradients of barycentric functions
        b = np.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], 1)
        c = np.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], 1)
        det = x[:, 0] * b[:, 0] + x[:, 1] * b[:, 1] + x[:, 2] * b[:, 2]     # 2 * signed area
        area = 0.5 * np.abs(det)
        self.area = area
        gx = b / det[:, None]
        gy = c / det[:, None]
        self.gx, self.gy = gx, gy
        Ke = area[:, None, None] * (gx[:, :, None] * gx[:, None, :] + gy[:, :, None] * gy[:, None, :])
        Me = area[:, None, None] / 12.0 * (np.ones((3, 3)) + np.eye(3))[None]
        rows = np.repeat(t, 3, axis=1).ravel()            # i index repeated over j
        cols = np.tile(t, (1, 3)).ravel()
        self._rows, self._cols = rows, cols
        n = self.n
        self.K = sp.csr_matrix((Ke.ravel(), (rows, cols)), shape=(n, n))
        self.Mass = sp.csr_matrix((Me.ravel(), (rows, cols)), shape=(n, n))
        self.K.sum_duplicates()
        self.Mass.sum_duplicates()
        # scatter operator for element 3x3 blocks with quadrature-weighted coefficients (for G(c)):
        # local basis products Lam[q,i]*Lam[q,j] -> (6, 9)
        lam = _SF_PERMS
        self._lam = lam
        self._lamlam = (lam[:, :, None] * lam[:, None, :]).reshape(6, 9)
        # element -> node scatter for vectors (g(c))
        self._S = sp.csr_matrix((np.ones(t.size), (t.ravel(), np.arange(t.size))), shape=(n, t.size))
        # element-block -> matrix scatter with fixed pattern
        pat = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n)).tocsr()
        pat.sum_duplicates()
        self._pat = pat
        # map each (row, col) pair to its CSR slot
        keys_pat = (pat.indices.astype(np.int64) + n * np.repeat(np.arange(n, dtype=np.int64), np.diff(pat.indptr)))
        keys = cols.astype(np.int64) + n * rows.astype(np.int64)
        slot = np.searchsorted(keys_pat, keys)
        assert np.all(keys_pat[slot] == keys)
        self._P = sp.csr_matrix((np.ones(slot.size), (slot, np.arange(slot.size))), shape=(pat.nnz, slot.size))

    # ---- nonlinear pieces (6-point rule) --------------------------------------------------------------
    def quad_values(self, u):
        """u at the 6 quadrature points of every triangle: (ntri, 6)."""
        return u[self.tri] @ self._lam.T

    def g_vec(self, p: Params, c):
        """g_i = int f'(c_h) lambda_i  (degree-3 rule)."""
        cq = self.quad_values(c)
        w = (self.area[:, None] * _SF_W[None, :]) * p.df(cq)           # (ntri, 6)
        ge = w @ self._lam                                              # (ntri, 3)
        return self._S @ ge.ravel()

    def G_mat(self, p: Params, c):
        """G_ij = int f''(c_h) lambda_i lambda_j  (degree-3 rule), CSR with the K/M sparsity pattern."""
        cq = self.quad_values(c)
        w = (self.area[:, None] * _SF_W[None, :]) * p.d2f(cq)          # (ntri, 6)
        Ge = w @ self._lamlam                                           # (ntri, 9)
        data = self._P @ Ge.ravel()
        return sp.csr_matrix((data, self._pat.indices, self._pat.indptr), shape=self._pat.shape)

    # ---- diagnostics (dolfin/bench1.py:121-125; bench6.py:155-165) -------------------------------------
    def total_solute(self, c):
        return float(np.sum(self.area * c[self.tri].mean(axis=1)))

    def total_free_energy(self, p: Params, c, phi=None):
        cq = self.quad_values(c)
        fch = np.sum(self.area[:, None] * _SF_W[None, :] * p.f(cq))
        ce = c[self.tri]
        gxc = np.sum(self.gx * ce, axis=1)
        gyc = np.sum(self.gy * ce, axis=1)
        fgr = 0.5 * p.kappa * np.sum(self.area * (gxc * gxc + gyc * gyc))
        fel = 0.0
        if phi is not None:
            fel = 0.5 * p.k * float(c @ (self.Mass @ phi))
        return float(fch + fgr + fel)


class FemBE:
    """Backward-Euler Newton solver on the crossed P1 mesh (BM1: fields c, mu; BM6: c, mu, phi)."""

    def __init__(self, model="bm1", L=None, N=100, params: Params | None = None, newton_atol=1e-6, newton_max=10):
        assert model in ("bm1", "bm6")
        self.model = model
        if L is None:
            L = 200.0 if model == "bm1" else 100.0       # bench1.py:21 / bench6.py:22
        self.mesh = CrossedMesh(L, N)
        self.p = params or Params()
        self.atol, self.newton_max = newton_atol, newton_max
        m = self.mesh
        n = m.n
        if model == "bm1":
            self.c = ic_bm1(m.x, m.y)
        else:
            self.c = ic_bm6(m.x, m.y)
        self.mu = np.zeros(n)
        self.phi = None
        if model == "bm6":
            # Dirichlet data (bench6.py:77-90): phi=0 on x=0, phi=sin(y/7) on x=Lx, all boundary nodes incl. corners
            tol = 1e-12
            left = np.where(np.abs(m.x) < tol)[0]
            right = np.where(np.abs(m.x - m.L) < tol)[0]
            self.bc_idx = np.concatenate([left, right])
            self.bc_val = np.concatenate([np.zeros(left.size), np.sin(m.y[right] / 7.0)])
            self.phi = np.zeros(n)
            self.phi[self.bc_idx] = self.bc_val
            free = np.ones(n, bool)
            free[self.bc_idx] = False
            self._free_diag = sp.diags(free.astype(float))
            self._bc_diag = sp.diags((~free).astype(float))
        self.t = 0.0
        self.last_newton_iters = 0

    # residual / Jacobian ---------------------------------------------------------------------------------
    def _residual(self, c, mu, phi, c0, dt):
        m, p = self.mesh, self.p
        Rc = m.Mass @ (c - c0) / dt + p.M * (m.K @ mu)
        Rmu = m.Mass @ mu - m.g_vec(p, c) - p.kappa * (m.K @ c)
        if self.model == "bm1":
            return np.concatenate([Rc, Rmu])
        Rmu = Rmu - p.k * (m.Mass @ phi)
        Rphi = -(m.K @ phi) + (p.k / p.eps_r) * (m.Mass @ c)
        Rphi[self.bc_idx] = phi[self.bc_idx] - self.bc_val
        return np.concatenate([Rc, Rmu, Rphi])

    def _jacobian(self, c, dt):
        m, p = self.mesh, self.p
        G = m.G_mat(p, c)
        A11 = m.Mass / dt
        A12 = p.M * m.K
        A21 = -(G + p.kappa * m.K)
        A22 = m.Mass
        if self.model == "bm1":
            return sp.bmat([[A11, A12], [A21, A22]], format="csc")
        A23 = -p.k * m.Mass
        A31 = self._free_diag @ ((p.k / p.eps_r) * m.Mass)
        A33 = self._free_diag @ (-m.K) + self._bc_diag
        return sp.bmat([[A11, A12, None], [A21, A22, A23], [A31, None, A33]], format="csc")

    def step(self, dt):
        """One backward-Euler step with plain (undamped) Newton; returns (iterations, converged)."""
        n = self.mesh.n
        c0 = self.c.copy()
        c, mu = self.c.copy(), self.mu.copy()
        phi = None if self.phi is None else self.phi.copy()
        converged = False
        its = 0
        for its in range(self.newton_max + 1):
            R = self._residual(c, mu, phi, c0, dt)
            if np.linalg.norm(R) < self.atol:
                converged = True
                break
            if its == self.newton_max:
                break
            J = self._jacobian(c, dt)
            d = spla.splu(J).solve(-R)
            c = c + d[:n]
            mu = mu + d[n:2 * n]
            if phi is not None:
                phi = phi + d[2 * n:]
        self.last_newton_iters = its
        if converged:
            self.c, self.mu, self.phi = c, mu, phi
            self.t += dt
        return its, converged

    def diagnostics(self):
        m = self.mesh
        return m.total_free_energy(self.p, self.c, self.phi), m.total_solute(self.c)

    def run_on_time_grid(self, times, callback=None):
        """Advance through the given accepted-step times (fixture column 1); returns rows [t, F, C]."""
        rows = []
        tprev = 0.0
        for tn in times:
            dt = tn - tprev
            its, ok = self.step(dt)
            if not ok:
                raise RuntimeError("Newton failed at t=%g dt=%g after %d iterations" % (tn, dt, its))
            self.t = tn
            tprev = tn
            F, C = self.diagnostics()
            rows.append([tn, F, C])
            if callback is not None:
                callback(self, its)
        return np.array(rows)


def _main():
    import argparse
    import os
    import time
    ap = argparse.ArgumentParser(description="run the FEM-BE oracle on a fixture time grid")
    ap.add_argument("--model", default="bm1", choices=["bm1", "bm6"])
    ap.add_argument("--rows", type=int, default=6)
    ap.add_argument("--golden", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    a = ap.parse_args()
    csv = np.loadtxt(os.path.join(a.golden, "bench1_out.csv" if a.model == "bm1" else "bench6_out.csv"),
                     delimiter=",", skiprows=1)
    s = FemBE(a.model, newton_max=100)      # the committed time grid needs up to 24 plain-Newton iterations
    F0, C0 = s.diagnostics()
    print("t=0: F=%.10f C=%.10f" % (F0, C0))
    t0 = time.time()

    def cb(solver, its):
        i = cb.i
        F, C = solver.diagnostics()
        print("row %2d t=%9.4f its=%2d F=%.10f relerr=%.2e C=%.10f relerr=%.2e  [%.1fs]" % (
            i, solver.t, its, F, abs(F - csv[i, 1]) / csv[i, 1], C, abs(C - csv[i, 2]) / csv[i, 2], time.time() - t0))
        cb.i += 1
    cb.i = 0
    s.run_on_time_grid(csv[:a.rows, 0], cb)


if __name__ == "__main__":
    _main()
