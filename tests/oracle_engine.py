"""Test-only compute engine with the interface of pfhubbenchmarks_amd.solver.HipSlabEngine, backed by the CPU
oracle (oracle/ch_fd.c) and CPU torch tensors, so the slab / halo-exchange logic of SlabSolver can run under the
gloo backend without a GPU.  Never imported by product code."""
import contextlib

import numpy as np
import torch

from oracle import ch_fd
from pfhubbenchmarks_amd.solver import slab_partition


class OracleSlabEngine:
    ghost = 2

    def __init__(self, n, h, nranks, rank, bc="periodic", **params):
        nx, ny, nz = (n, n, n) if isinstance(n, int) else n
        self.nx, self.ny, self.nz_global = nx, ny, nz
        self.h = h
        self.bc = bc
        self.z0, self.nz = slab_partition(nz, nranks, rank)
        lx, ly = (2 * (nx - 1), 2 * (ny - 1)) if bc == "mirror" else (nx, ny)
        self.buffers = [torch.zeros((self.nz + 4, ly, lx), dtype=torch.float64) for _ in range(2)]
        self._cur = 0
        if bc == "mirror":        # a line of slabs over the physical planes; walls mirror locally (pfhip.h: nranks)
            assert self.nz >= 3
            self.rank_lo = rank - 1 if rank > 0 else -1
            self.rank_hi = rank + 1 if rank < nranks - 1 else -1
        else:
            self.rank_lo = (rank - 1) % nranks
            self.rank_hi = (rank + 1) % nranks
        self._open = None

    def _reflect(self):
        if self.bc != "mirror":
            return
        b, nz = self.buffers[self._cur], self.nz
        if self.rank_lo < 0:
            b[1], b[0] = b[3].clone(), b[4].clone()
        if self.rank_hi < 0:
            b[nz + 2], b[nz + 3] = b[nz].clone(), b[nz - 1].clone()

    @property
    def cur(self):
        return self._cur

    def stream_context(self):
        return contextlib.nullcontext()

    def set_local(self, arr):
        a = np.ascontiguousarray(arr)
        if self.bc == "mirror":   # even extension in y and x of every owned plane
            a = np.stack([ch_fd.even_extend(p) for p in a])
        self.buffers[self._cur][2:2 + self.nz] = torch.from_numpy(np.ascontiguousarray(a))

    def get_local(self):
        return self.buffers[self._cur][2:2 + self.nz, :self.ny, :self.nx].numpy().copy()

    def _launch(self, dt, zlo, zhi):
        if zhi <= zlo:
            return
        ch_fd.fd_step(self.buffers[self._cur].numpy(), dt, h=self.h, ghost=2, zwrap=0, zlo=zlo, zhi=zhi,
                      out=self.buffers[1 - self._cur].numpy())

    def step_begin(self, dt):
        assert self._open is None
        self._reflect()
        self._launch(dt, 2, self.nz - 2)
        self._open = dt

    def step_finish(self):
        dt, nz = self._open, self.nz
        if nz - 2 > 2:
            self._launch(dt, 0, 2)
            self._launch(dt, nz - 2, nz)
        else:
            self._launch(dt, 0, nz)
        self._cur ^= 1
        self._open = None

    def diag_local(self):
        if self.bc == "mirror":
            return self._diag_mirror()
        F, C, E = ch_fd.diagnostics(self.buffers[self._cur].numpy(), h=self.h, dim=3, ghost=2, zwrap=0)
        return [F, C, E]

    def _diag_mirror(self, rho=5.0, ca=0.3, cb=0.7, kappa=2.0):
        """Trapezoid weights along z (1/2 on the wall planes), no z-difference out of the top wall; x and y on the
        even extension (factor 1/4) -- the weighting of csrc/diag_kernels.hip (zends)."""
        self._reflect()
        b = self.buffers[self._cur].numpy()
        own, up = b[2:2 + self.nz], b[3:3 + self.nz]
        wz = np.ones(self.nz)
        wd = np.ones(self.nz)
        if self.rank_lo < 0:
            wz[0] = 0.5
        if self.rank_hi < 0:
            wz[-1] = 0.5
            wd[-1] = 0.0
        f = rho * ((own - ca) * (cb - own)) ** 2
        gxy = (np.roll(own, -1, 2) - own) ** 2 + (np.roll(own, -1, 1) - own) ** 2
        gz = (up - own) ** 2
        W = wz[:, None, None]
        vol = self.h ** 3 * 0.25
        F = vol * (np.sum(W * f) + 0.5 * kappa / self.h ** 2 * (np.sum(W * gxy) + np.sum(wd[:, None, None] * gz)))
        return [F, vol * np.sum(W * own), 0.0]

    def sync(self):
        pass


class OracleWideSlabEngine(OracleSlabEngine):
    """CPU mirror of HipSlabEngine(wide=True) -- PF_FLAG_WIDE_HALO (include/pfhip.h): 4 ghost planes exchanged every second
    step.  Step A (ghosts fresh) computes real planes [-2, nz+2): begin = the interior [2, nz-2), finish = the two 4-plane
    strips; step B computes [0, nz) in one go, no exchange.  The oracle sees the buffer as nz + 4 owned planes with 2
    ghost planes (virtual plane = real plane + 2), exactly as the library hands it to the fused kernel."""
    ghost = 4
    wide = True

    def __init__(self, n, h, nranks, rank, bc="periodic", **params):
        super().__init__(n, h, nranks, rank, bc=bc, **params)
        assert self.nz >= (5 if bc == "mirror" else 4)
        shape = (self.nz + 8,) + tuple(self.buffers[0].shape[1:])
        self.buffers = [torch.zeros(shape, dtype=torch.float64) for _ in range(2)]
        self.phase = 0

    def needs_exchange(self):
        return self.phase == 0

    def _reflect(self):
        if self.bc != "mirror":
            return
        b, nz, g = self.buffers[self._cur], self.nz, 4
        for k in range(1, g + 1):
            if self.rank_lo < 0:
                b[g - k] = b[g + k].clone()
            if self.rank_hi < 0:
                b[g + nz - 1 + k] = b[g + nz - 1 - k].clone()

    def set_local(self, arr):
        a = np.ascontiguousarray(arr)
        if self.bc == "mirror":
            a = np.stack([ch_fd.even_extend(p) for p in a])
        self.buffers[self._cur][4:4 + self.nz] = torch.from_numpy(np.ascontiguousarray(a))
        self.phase = 0

    def get_local(self):
        return self.buffers[self._cur][4:4 + self.nz, :self.ny, :self.nx].numpy().copy()

    def step_begin(self, dt):
        assert self._open is None
        self._reflect()
        nz = self.nz
        if self.phase == 0:
            self._launch(dt, 4, nz)                # real [2, nz - 2)
        else:
            self._launch(dt, 2, nz + 2)            # real [0, nz)
        self._open = dt

    def step_finish(self):
        dt, nz = self._open, self.nz
        if self.phase == 0:
            if nz > 4:
                self._launch(dt, 0, 4)             # real [-2, 2)
                self._launch(dt, nz, nz + 4)       # real [nz - 2, nz + 2)
            else:
                self._launch(dt, 0, nz + 4)
        self._cur ^= 1
        self.phase ^= 1
        self._open = None

    def diag_local(self):
        if self.bc == "mirror":
            self._reflect()
            b = self.buffers[self._cur]
            narrow = OracleSlabEngine.__new__(OracleSlabEngine)       # reuse the 2-ghost trapezoid sums on a view
            narrow.__dict__.update(self.__dict__)
            narrow.buffers = [b[2:-2], b[2:-2]]
            narrow._cur = 0
            return OracleSlabEngine._diag_mirror(narrow)
        F, C, E = ch_fd.diagnostics(self.buffers[self._cur].numpy(), h=self.h, dim=3, ghost=4, zwrap=0)
        return [F, C, E]


class OracleFFTSlabEngine(OracleSlabEngine):
    """CPU mirror of HipFFTSlabEngine: the same distributed state machine (pf_dist_begin / pf_dist_advance in
    csrc/pfhip_api.hip) restated with numpy FFTs, so FFTSlabSolver's collectives run under gloo without a GPU."""

    def __init__(self, n, h, nranks, rank, scheme="fd", model="bm1", k=0.09, eps=90.0, kappa=2.0, M=5.0,
                 eliminate_phi=False, dirichlet=None):
        """dirichlet = (npx, npy): BM6 with the reference's boundary conditions (bench6.py:77-90) -- n is then the LATTICE of
        the even extension (2 (np - 1) points per axis), the slabs a ring over its planes, and the Poisson solve transforms
        the odd-in-x right-hand side (csrc/pfhip_api.hip pf_dist_advance, g.mirror branches)"""
        super().__init__(n, h, nranks, rank)
        self.dirichlet = dirichlet
        self.eliminate_phi = eliminate_phi
        self.cbar = None
        self.P, self.rank = nranks, rank
        self.scheme, self.model = scheme, model
        self.k, self.eps, self.kappa, self.M = k, eps, kappa, M
        nx, ny, nz = self.nx, self.ny, self.nz_global
        assert ny % nranks == 0 and nz % nranks == 0
        self.nyl, self.nxh = ny // nranks, nx // 2 + 1
        na = 2 * self.nz * ny * self.nxh
        self.a2a = [torch.zeros(na, dtype=torch.float64) for _ in range(2)]
        self.phi = torch.zeros((self.nz + 4, ny, nx), dtype=torch.float64)
        self.phi_valid = False
        self.chat = None
        self.op = 0

    def set_mean_c(self, v):
        self.cbar = float(v)

    def _launch(self, dt, zlo, zhi):          # FD slab step; with phi eliminated it carries the gq (c - cbar) term
        if zhi <= zlo:
            return
        elim = (self.k, self.eps, self.cbar) if self.eliminate_phi else None
        ch_fd.fd_step(self.buffers[self._cur].numpy(), dt, h=self.h, ghost=2, zwrap=0, zlo=zlo, zhi=zhi,
                      out=self.buffers[1 - self._cur].numpy(), elim=elim)

    # complex views of the all-to-all buffers
    def _A(self):
        return self.a2a[0].numpy().view(np.complex128).reshape(self.P, self.nz, self.nyl, self.nxh)

    def _B(self):
        return self.a2a[1].numpy().view(np.complex128).reshape(self.nz_global, self.nyl, self.nxh)

    def _owned(self, t):
        return t[2:2 + self.nz].numpy()

    def _fwd_local(self, real):
        tmp = np.fft.rfft2(real, axes=(1, 2))
        self._A()[...] = tmp.reshape(self.nz, self.P, self.nyl, self.nxh).transpose(1, 0, 2, 3)

    def _inv_local(self, out):
        tmp = self._A().transpose(1, 0, 2, 3).reshape(self.nz, self.ny, self.nxh)
        out[...] = np.fft.irfft2(tmp, s=(self.ny, self.nx), axes=(1, 2))

    def _kidx(self):
        kz = np.arange(self.nz_global)[:, None, None]
        ky = (self.rank * self.nyl + np.arange(self.nyl))[None, :, None]
        kx = np.arange(self.nxh)[None, None, :]
        return kx, ky, kz

    def dist_begin(self, op, dt=0.0):
        assert self.op == 0
        self.op, self.phase, self.dt = op, 0, dt

    def _done(self):
        self.op = 0
        return ("done",)

    def dist_advance(self):
        nx, ny, nz, h = self.nx, self.ny, self.nz_global, self.h
        cur = self.buffers[self._cur]
        if self.scheme == "spectral":
            while True:
                if self.phase == 0:
                    if self.chat is not None:
                        self.phase = 2
                        continue
                    self._fwd_local(self._owned(cur))
                    self.phase = 1
                    return ("alltoall", self.a2a[1], self.a2a[0])
                if self.phase == 1:
                    self.chat = np.fft.fft(self._B(), axis=0)
                    self.phase = 2
                    continue
                if self.phase == 2:
                    if self.op == 2:
                        return self._done()
                    from oracle import ch_spectral
                    self._fwd_local(ch_spectral.fprime(self._owned(cur)))
                    self.phase = 3
                    return ("alltoall", self.a2a[1], self.a2a[0])
                if self.phase == 3:
                    gh = np.fft.fft(self._B(), axis=0)
                    kx, ky, kz = self._kidx()
                    ky = np.where(2 * ky > ny, ky - ny, ky)
                    kz = np.where(2 * kz > nz, kz - nz, kz)
                    k2 = ((2 * np.pi / (nx * h) * kx) ** 2 + (2 * np.pi / (ny * h) * ky) ** 2) + (2 * np.pi / (nz * h) * kz) ** 2
                    self.chat = (self.chat - (self.dt * self.M) * k2 * gh) / (1.0 + (self.dt * self.M * self.kappa) * k2 ** 2)
                    self._B()[...] = np.fft.ifft(self.chat, axis=0)
                    self.phase = 4
                    return ("alltoall", self.a2a[0], self.a2a[1])
                self._inv_local(self._owned(self.buffers[1 - self._cur]))
                self._cur ^= 1
                return self._done()
        if self.model == "bm6":
            if self.phase == 0:
                if self.phi_valid and self.op == 2:
                    return self._done()
                if self.dirichlet:      # rhs_dirichlet_kernel on this rank's lattice planes
                    npx, npy = self.dirichlet
                    N = npx - 1
                    x, y = np.arange(nx), np.arange(ny)
                    xr = np.where(x <= N, x, 2 * N - x)
                    yr = np.where(y < npy, y, 2 * (npy - 1) - y)
                    r = -(self.k / self.eps) * self._owned(cur).copy()
                    r[..., xr == N - 1] -= (np.sin(yr * h / 7.0) / (h * h))[:, None]
                    r = r * np.where((xr == 0) | (xr == N), 0.0, np.where(x > N, -1.0, 1.0))
                    self._fwd_local(r)
                else:
                    self._fwd_local(self._owned(cur))
                self.phase = 1
                return ("alltoall", self.a2a[1], self.a2a[0])
            if self.phase == 1:
                ch = np.fft.fft(self._B(), axis=0)
                kx, ky, kz = self._kidx()
                lam = ((2 * np.cos(2 * np.pi * kx / nx) - 2) + (2 * np.cos(2 * np.pi * ky / ny) - 2)
                       + (2 * np.cos(2 * np.pi * kz / nz) - 2)) / (h * h)
                with np.errstate(divide="ignore", invalid="ignore"):
                    ph = np.where(lam != 0.0, (1.0 if self.dirichlet else -(self.k / self.eps)) * ch / lam, 0.0)
                self._B()[...] = np.fft.ifft(ph, axis=0)
                self.phase = 2
                return ("alltoall", self.a2a[0], self.a2a[1])
            if self.phase == 2:
                self._inv_local(self._owned(self.phi))
                if self.dirichlet:      # fixup_dirichlet_kernel: even-in-x extension with the boundary values written
                    npx, npy = self.dirichlet
                    N = npx - 1
                    x, y = np.arange(nx), np.arange(ny)
                    yr = np.where(y < npy, y, 2 * (npy - 1) - y)
                    p = self._owned(self.phi)
                    p *= np.where(x > N, -1.0, 1.0)
                    p[..., x == 0] = 0.0
                    p[..., x == N] = np.sin(yr * h / 7.0)[:, None]
                self.phase = 3
                return ("halo", [cur, self.phi])
            self.phi_valid = True
            if self.op == 1 and self.eliminate_phi:
                self._launch(self.dt, 0, self.nz)
                self._cur ^= 1
                self.phi_valid = False
            elif self.op == 1:
                ch_fd.fd_step(cur.numpy(), self.dt, h=h, phi=self.phi.numpy(), k_phi=self.k, ghost=2, zwrap=0,
                              out=self.buffers[1 - self._cur].numpy())
                self._cur ^= 1
                self.phi_valid = False
            return self._done()
        if self.phase == 0:
            self.phase = 1
            return ("halo", [cur])
        if self.op == 1:
            self._launch(self.dt, 0, self.nz)
            self._cur ^= 1
        return self._done()

    def step_finish(self):
        super().step_finish()
        self.phi_valid = False

    def diag_local(self):
        cur = self.buffers[self._cur].numpy()
        if self.scheme == "spectral":
            c = cur[2:2 + self.nz]
            f = 5.0 * ((c - 0.3) * (0.7 - c)) ** 2
            kx, ky, kz = self._kidx()
            w = np.where((kx == 0) | (2 * kx == self.nx), 1.0, 2.0)
            ky = np.where(2 * ky > self.ny, ky - self.ny, ky)
            kz = np.where(2 * kz > self.nz_global, kz - self.nz_global, kz)
            h = self.h
            k2 = ((2 * np.pi / (self.nx * h) * kx) ** 2 + (2 * np.pi / (self.ny * h) * ky) ** 2) + (
                2 * np.pi / (self.nz_global * h) * kz) ** 2
            g = np.sum(w * k2 * np.abs(self.chat) ** 2) / (self.nx * self.ny * self.nz_global)
            vol = h ** 3
            return [vol * (f.sum() + 0.5 * self.kappa * g), vol * c.sum(), 0.0]
        phi = self.phi.numpy() if self.model == "bm6" else None
        F, C, E = ch_fd.diagnostics(cur, h=self.h, dim=3, ghost=2, zwrap=0, phi=phi, k=self.k)
        if self.dirichlet:      # sums over the even extension along all three axes: the physical box is 1/8 of it
            F, C, E = F / 8.0, C / 8.0, E / 8.0
        return [F, C, E]


class OracleMultiFieldSlabEngine:
    """CPU stand-in for HipMultiFieldSlabEngine (same interface: buffers, cur, nf, ghost, nz, rank_lo / rank_hi, step_local,
    diag_local): the local box INCLUDING its ghost planes is stepped as a periodic box by oracle/multi_fd.py -- exactly what
    the HIP kernels do with it (the wrap only pollutes the ghost planes, which the next exchange overwrites).  Test
    infrastructure only: lets the gloo jobs drive pfhubbenchmarks_amd.solver.MultiFieldSlabSolver on CPU."""

    device = None

    def __init__(self, model, n, h, nranks, rank):
        import torch
        from oracle import multi_fd
        from pfhubbenchmarks_amd.solver import slab_partition
        self.torch = torch
        self.mf = multi_fd
        self.model, self.h = model, float(h)
        self.nx, self.ny, self.nz_global = n
        self.nf = 5 if model == "bm2" else 2
        self.ghost = 2 if model == "bm2" else 1
        self.z0, self.nz = slab_partition(n[2], nranks, rank)
        self.rank_lo, self.rank_hi = (rank - 1) % nranks, (rank + 1) % nranks
        shape = (self.nf, self.nz + 2 * self.ghost, self.ny, self.nx)
        self.buffers = [torch.zeros(shape, dtype=torch.float64) for _ in range(2)]
        self.cur = 0
        self.nranks = nranks

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def sync(self):
        pass

    def set_local(self, u):
        """u: (nf, nz_local, ny, nx)"""
        self.buffers[self.cur][:, self.ghost:self.ghost + self.nz] = self.torch.as_tensor(np.ascontiguousarray(u))

    def get_local(self):
        return self.buffers[self.cur][:, self.ghost:self.ghost + self.nz].numpy().copy()

    def step_local(self, dt):
        step = self.mf.bm2_step if self.model == "bm2" else self.mf.bm3_step
        new = step(self.buffers[self.cur].numpy(), dt, self.h)
        self.cur = 1 - self.cur
        self.buffers[self.cur][...] = self.torch.as_tensor(np.ascontiguousarray(new))

    def step_begin(self, dt):
        """the planes that need no ghosts, computed with the ghost planes POISONED (NaN): a protocol that read a ghost plane
        before the exchange has delivered would show up as a NaN in the result"""
        g, nz = self.ghost, self.nz
        step = self.mf.bm2_step if self.model == "bm2" else self.mf.bm3_step
        tmp = self.buffers[self.cur].numpy().copy()
        tmp[:, :g] = np.nan
        tmp[:, nz + g:] = np.nan
        with np.errstate(invalid="ignore"):
            new = step(tmp, dt, self.h)
        self._interior = new[:, 2 * g:nz].copy()
        assert np.isfinite(self._interior).all()
        self._dt = dt

    def step_finish(self):
        """the owned planes next to the ghost layers (fresh ghosts needed); swap"""
        g, nz = self.ghost, self.nz
        step = self.mf.bm2_step if self.model == "bm2" else self.mf.bm3_step
        new = step(self.buffers[self.cur].numpy(), self._dt, self.h)
        new[:, 2 * g:nz] = self._interior
        self.cur = 1 - self.cur
        self.buffers[self.cur][...] = self.torch.as_tensor(np.ascontiguousarray(new))

    def diag_local(self):
        """two linear functionals of the owned planes (the energy itself is checked on the GPU against the oracle; what the
        gloo jobs check is MultiFieldSlabSolver's all-reduce of whatever the engine returns)"""
        u = self.buffers[self.cur][:, self.ghost:self.ghost + self.nz].numpy()
        return np.array([u[0].sum(), (u[-1] * u[-1]).sum(), 0.0])
