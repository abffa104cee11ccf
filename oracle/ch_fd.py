"""ORACLE (test infrastructure, not product code) -- ctypes wrapper of oracle/ch_fd.c (liborc_fd.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Build: `make -C oracle` (done by __graft_entry__.build()).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcParams(C.Structure):
    _fields_ = [("c_alpha", C.c_double), ("c_beta", C.c_double), ("two_rho", C.c_double),
                ("kappa_over_h2", C.c_double), ("dtM_over_h2", C.c_double), ("k_phi", C.c_double),
                ("gq", C.c_double), ("cbar", C.c_double)]


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    name = "liborc_fd.so" if (_cpu_has("fma") and _cpu_has("avx2")) else "liborc_fd_generic.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise ImportError("%s not built: run `make -C oracle`" % path)
    lib = C.CDLL(path)
    P = C.c_void_p
    lib.orc_ch_fd_step.restype = C.c_int
    lib.orc_ch_fd_step.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(OrcParams)]
    lib.orc_ch_mu.restype = C.c_int
    lib.orc_ch_mu.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcParams)]
    lib.orc_ch_diag.restype = C.c_int
    lib.orc_ch_diag.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                C.c_double, C.POINTER(C.c_double)]
    lib.orc_set_threads.restype = C.c_int
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_ic.restype = C.c_int
    lib.orc_ic.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
    _lib = lib
    return lib


def host_cores():
    """CPU cores this process may really use: min(affinity, cgroup quota) -- os.cpu_count() over-reports on a
    GPU box whose container owns a 16-core share of a 256-thread host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def set_threads(n=0):
    """OpenMP threads for the C oracle; n = 0 -> host_cores().  Returns the count in effect."""
    return load().orc_set_threads(int(n) if n > 0 else host_cores())


def make_params(dt, h=1.0, rho_s=5.0, c_alpha=0.3, c_beta=0.7, kappa=2.0, M=5.0, k_phi=0.0, elim=None):
    """Same host arithmetic as libpfhip's make_args (pfhip_api.hip): two_rho = 2 rho, kappa/(h*h), dt*M/(h*h).
    elim = (k, eps, cbar): BM6 with phi eliminated -> gq = -(dt*M)*(k*k/eps)."""
    gq, cbar = (0.0, 0.0) if elim is None else (-(dt * M) * (elim[0] * elim[0] / elim[1]), elim[2])
    return OrcParams(c_alpha, c_beta, 2.0 * rho_s, kappa / (h * h), dt * M / (h * h), k_phi, gq, cbar)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fd_step(c, dt, h=1.0, phi=None, ghost=0, zwrap=1, zlo=None, zhi=None, out=None, **model):
    """One explicit FD step.  c: (nz [+2*ghost], ny, nx) or (ny, nx) float64.  Returns the new array (same shape;
    ghost planes of the result are left as in `out`/zeros)."""
    lib = load()
    c = np.ascontiguousarray(c, dtype=np.float64)
    two_d = c.ndim == 2
    c3 = c[None] if two_d else c
    nzb, ny, nx = c3.shape
    nz = nzb - 2 * ghost
    zlo = 0 if zlo is None else zlo
    zhi = nz if zhi is None else zhi
    o = np.zeros_like(c3) if out is None else out
    q = make_params(dt, h, **model)
    rc = lib.orc_ch_fd_step(_p(c3), _p(o), _p(phi), nx, ny, nz, ghost, zwrap, zlo, zhi, C.byref(q))
    if rc != 0:
        raise ValueError("orc_ch_fd_step: bad arguments")
    return o[0] if two_d else o


def diag_raw(c, phi=None, ghost=0, zwrap=1, rho_s=5.0, c_alpha=0.3, c_beta=0.7):
    lib = load()
    c = np.ascontiguousarray(c, dtype=np.float64)
    c3 = c[None] if c.ndim == 2 else c
    nzb, ny, nx = c3.shape
    out = (C.c_double * 6)()
    lib.orc_ch_diag(_p(c3), _p(phi), nx, ny, nzb - 2 * ghost, ghost, zwrap, rho_s, c_alpha, c_beta, out)
    return np.array(list(out))


def diagnostics(c, h=1.0, dim=None, phi=None, ghost=0, zwrap=1, rho_s=5.0, c_alpha=0.3, c_beta=0.7, kappa=2.0,
                k=0.09, mirror=False):
    """(total_free_energy, total_solute, f_elec) with the scaling of libpfhip's scale_diag."""
    c = np.asarray(c)
    dim = c.ndim if dim is None else dim
    raw = diag_raw(c, phi, ghost, zwrap, rho_s, c_alpha, c_beta)
    vol = h ** dim * (0.5 ** dim if mirror else 1.0)
    felec = 0.5 * k * raw[3] if phi is not None else 0.0
    F = vol * (raw[1] + 0.5 * kappa / (h * h) * raw[2] + felec)
    return F, vol * raw[0], vol * felec


def ic(nx, ny, nz=1, h=1.0, c0=0.5, amp=0.05, w0=0.105):
    lib = load()
    out = np.empty((nz, ny, nx), dtype=np.float64)
    lib.orc_ic(_p(out), nx, ny, nz, h, c0, amp, w0)
    return out


def even_extend(a):
    """(N+1)^d nodal array of a no-flux domain -> (2N)^d periodic lattice (what pf_set_field does for PF_BC_MIRROR)."""
    a = np.asarray(a)
    for ax in range(a.ndim):
        n = a.shape[ax]
        idx = np.concatenate([np.arange(n), np.arange(n - 2, 0, -1)])
        a = np.take(a, idx, axis=ax)
    return np.ascontiguousarray(a)
