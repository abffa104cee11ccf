"""Scheme-to-reference verification (SURVEY.md section 7.0-5 route (a); VERDICT r01 "next round" #3).

The committed reference trajectory (results/bench1_out.csv) carries the backward-Euler error of its own large steps
(dt = 1.6 at t = 7.9: 3.6e-2 in F), so the throughput schemes (explicit FD, semi-implicit spectral) cannot be compared
with it row by row.  What CAN be compared is the limit: the reference's ALGORITHM (P1 'crossed' FEM + backward Euler,
dolfin/pfbase.py:361-383, run here by the GPU BE-parity mode that reproduces the committed CSV to 5e-9) refined in
h and dt, against the FD and the spectral scheme refined the same way.  All three are Richardson-extrapolated:

  FEM-BE   F(h, dt) = F* + A h^2 + B dt + B2 dt^2 + C h^2 dt      5 runs: h = 2 at dt, dt/2, dt/4; h = 1 at dt, dt/2
  FD       F(h, dt) = F* + A h^2 + B dt + C h^2 dt                4 runs: h = 1, 0.5 at the stable dt and half of it
  spectral F(dt)    = F* + B dt + B2 dt^2   (spatially converged) 3 runs: dt, dt/2, dt/4 (+ a coarser lattice as a check)

Everything runs through the product API (PhaseFieldSolver over libpfhip); nothing here touches oracle/.
"""
from __future__ import annotations

import numpy as np

from .solver import PhaseFieldSolver, stable_dt

L_BM1 = 200.0          # dolfin/bench1.py:21
L_BM6 = 100.0          # dolfin/bench6.py:22


def _march(s, times, dt):
    """fixed-size steps to each report time (all times are multiples of dt to rounding); returns F at each"""
    out, t = [], 0.0
    for T in times:
        n = int(round((T - t) / dt))
        assert abs(n * dt - (T - t)) < 1e-9, (T, t, dt)
        s.step(dt, n)
        t = T
        out.append(s.diagnostics()[0])
    return np.array(out)


def fem_be_energy(intervals, dt, times, model="bm1"):
    """the reference's algorithm at mesh size h = L / intervals with fixed BE steps"""
    out, t = [], 0.0
    L = L_BM1 if model == "bm1" else L_BM6
    with PhaseFieldSolver(dim=2, n=intervals + 1, h=L / intervals, bc="mirror", scheme="fem_be", model=model) as s:
        (s.set_ic_bm1 if model == "bm1" else s.set_ic_bm6)()
        for T in times:
            n = int(round((T - t) / dt))
            assert abs(n * dt - (T - t)) < 1e-9
            for _ in range(n):
                ok, _, _ = s.step(dt, 1, check=True)
                if not ok:
                    raise RuntimeError("fem_be: Newton failed at t = %g (h = %g, dt = %g)" % (s.t, L / intervals, dt))
            t = T
            out.append(s.diagnostics()[0])
    return np.array(out)


def grid_energy(scheme, intervals, dt, times, model="bm1"):
    L = L_BM1 if model == "bm1" else L_BM6
    with PhaseFieldSolver(dim=2, n=intervals + 1, h=L / intervals, bc="mirror", scheme=scheme, model=model) as s:
        (s.set_ic_bm1 if model == "bm1" else s.set_ic_bm6)()
        return _march(s, times, dt)


def quad_extrapolate(f1, f2, f4):
    """limit dt -> 0 of f(dt) = f0 + b dt + b2 dt^2 from f(dt), f(dt/2), f(dt/4); also returns b2 dt^2"""
    # f1 - f2 = b dt/2 + 3/4 b2 dt^2 ; f2 - f4 = b dt/4 + 3/16 b2 dt^2
    b2dt2 = (8.0 / 3.0) * ((f1 - f2) - 2.0 * (f2 - f4))
    bdt = 2.0 * ((f1 - f2) - 0.75 * b2dt2)
    return f1 - bdt - b2dt2, b2dt2


def fem_be_limit(times, dt=0.1, log=None, model="bm1"):
    """F*(t) of the reference's algorithm, (h, dt) -> (0, 0)"""
    runs = {}
    L = L_BM1 if model == "bm1" else L_BM6
    for N, d in ((100, dt), (100, dt / 2), (100, dt / 4), (200, dt), (200, dt / 2)):
        runs[(N, d)] = fem_be_energy(N, d, times, model)
        if log:
            log("fem_be %s h = %g dt = %g: %s" % (model, L / N, d, np.array2string(runs[(N, d)], precision=6)))
    g2, b2dt2 = quad_extrapolate(runs[(100, dt)], runs[(100, dt / 2)], runs[(100, dt / 4)])      # h = 2, dt -> 0
    # h = 1: remove the dt^2 term found at h = 2, then linear extrapolation in dt
    a, b = runs[(200, dt)] - b2dt2, runs[(200, dt / 2)] - b2dt2 / 4.0
    g1 = 2.0 * b - a
    return (4.0 * g1 - g2) / 3.0, {"h2_dt0": g2, "h1_dt0": g1, "runs": runs}


def fd_limit(times, log=None):
    """explicit FD scheme, (h, dt) -> (0, 0): first order in dt, second order in h"""
    lim = {}
    for N in (200, 400):
        h = L_BM1 / N
        d = stable_dt(h, dim=2, safety=0.4)
        d = 0.1 / np.ceil(0.1 / d - 1e-9)      # a dt that divides every report time (all are multiples of 0.1)
        f1, f2 = grid_energy("fd", N, d, times), grid_energy("fd", N, d / 2, times)
        lim[N] = 2.0 * f2 - f1
        if log:
            log("fd h = %g dt = %.4g / half: %s / %s" % (h, d, np.array2string(f1, precision=6),
                                                           np.array2string(f2, precision=6)))
    return (4.0 * lim[400] - lim[200]) / 3.0, {"h1_dt0": lim[200], "h05_dt0": lim[400]}


def spectral_limit(times, intervals=256, dt=0.01, log=None):
    """semi-implicit spectral scheme on the even extension (2 * intervals lattice points), dt -> 0"""
    f = [grid_energy("spectral", intervals, d, times) for d in (dt, dt / 2, dt / 4)]
    if log:
        for d, v in zip((dt, dt / 2, dt / 4), f):
            log("spectral N = %d dt = %g: %s" % (intervals, d, np.array2string(v, precision=6)))
    lim, _ = quad_extrapolate(*f)
    return lim, {"runs": f}


def fd_limit_bm6(times, log=None):
    """BM6 (CH + Poisson with the reference's Dirichlet / no-flux phi, explicit coupling): FD scheme, (h, dt) -> (0, 0).
    h = 1, 0.5 on the 100 x 100 domain (bench6.py:22-24)."""
    lim = {}
    for N in (100, 200):
        h = L_BM6 / N
        d = stable_dt(h, dim=2, safety=0.4)
        d = 0.1 / np.ceil(0.1 / d - 1e-9)
        f1, f2 = grid_energy("fd", N, d, times, "bm6"), grid_energy("fd", N, d / 2, times, "bm6")
        lim[N] = 2.0 * f2 - f1
        if log:
            log("fd bm6 h = %g dt = %.4g / half: %s / %s" % (h, d, np.array2string(f1, precision=6),
                                                               np.array2string(f2, precision=6)))
    return (4.0 * lim[200] - lim[100]) / 3.0, {"h1_dt0": lim[100], "h05_dt0": lim[200]}


# ---- BM2 / BM3 (round 2): the explicit multi-field FD schemes against the reference's algorithm ----------------------
L_DOM = {"bm2": 200.0, "bm3": 960.0}          # bench2.py:21, bench3.py:21
N_REF = {"bm2": 100, "bm3": 350}              # bench2.py:22, bench3.py:22


def multi_fd_dt(model, h, safety=0.4):
    """forward-Euler limits: BM2 is bound by its Cahn-Hilliard part (M kappa_c lap_h^2), BM3 by the heat equation (D lap_h)"""
    if model == "bm2":
        return stable_dt(h, M=5.0, kappa=3.0, dim=2, safety=safety)
    return safety * h * h / (4.0 * 10.0)


def multi_energy(model, scheme, intervals, dt, times, max_newton=100):
    """F(t) and the second CSV column at the report times, fixed steps (all times multiples of dt)"""
    L = L_DOM[model]
    kw = dict(max_newton=max_newton) if scheme == "fem_be" else {}
    out, t = [], 0.0
    with PhaseFieldSolver(dim=2, n=intervals + 1, h=L / intervals, bc="mirror", scheme=scheme, model=model, **kw) as s:
        (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
        for T in times:
            n = int(round((T - t) / dt))
            assert abs(n * dt - (T - t)) < 1e-9 * max(1.0, T), (T, t, dt)
            if scheme == "fem_be":
                for _ in range(n):
                    ok, _, _ = s.step(dt, 1, check=True)
                    if not ok:
                        raise RuntimeError("%s fem_be: Newton failed at t = %g (dt = %g)" % (model, s.t, dt))
            else:
                ok, _, _ = s.step(dt, n, check=True)
                if not ok:
                    raise RuntimeError("%s fd: blow-up before t = %g (dt = %g)" % (model, T, dt))
            t = T
            out.append(s.diagnostics()[:2])
    return np.array(out)          # (ntimes, 2)


def multi_fd_limit(model, times, intervals=None, log=None):
    """explicit FD, (h, dt) -> (0, 0): first order in dt, second order in h; intervals = (coarse, fine = 2 coarse)"""
    n0 = intervals or (2 * N_REF[model] if model == "bm2" else N_REF[model])
    lim = {}
    for N in (n0, 2 * n0):
        h = L_DOM[model] / N
        d = multi_fd_dt(model, h)
        d = min(times) / np.ceil(min(times) / d - 1e-9)
        f1 = multi_energy(model, "fd", N, d, times)
        f2 = multi_energy(model, "fd", N, d / 2, times)
        lim[N] = 2.0 * f2 - f1
        if log:
            log("%s fd h = %g dt = %.4g / half: F %s / %s" % (model, h, d, np.array2string(f1[:, 0], precision=6),
                                                              np.array2string(f2[:, 0], precision=6)))
    return (4.0 * lim[2 * n0] - lim[n0]) / 3.0, lim


def multi_fem_dt_limit(model, times, dt, intervals=None, log=None):
    """the reference's algorithm at its own mesh (h fixed), dt -> 0 by quadratic extrapolation over dt, dt/2, dt/4"""
    N = intervals or N_REF[model]
    f = [multi_energy(model, "fem_be", N, d, times) for d in (dt, dt / 2, dt / 4)]
    if log:
        for d, v in zip((dt, dt / 2, dt / 4), f):
            log("%s fem_be h = %g dt = %g: F %s" % (model, L_DOM[model] / N, d, np.array2string(v[:, 0], precision=6)))
    lim, _ = quad_extrapolate(*f)
    return lim, f
