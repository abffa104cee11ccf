"""Benchmark drivers: the counterparts of the reference's dolfin/bench1.py and dolfin/bench6.py scripts.

Same contract (README.md:18-29): run from the repo root, hard-coded benchmark parameters (overridable by flags), rank-0
writes `results/bench<N>_out.csv` with header `time,total_free_energy,total_solute` and `fmt='%1.10f'`
(bench1.py:210-217; bench1 additionally writes results/bench1/stats.csv, the path today's bench1.py uses, and
results/bench1_out.csv, the path stats.jl:4 and b13d.py:200 use).

The time loop keeps the reference's shape (bench1.py:145-198): advance, on failure roll back and halve dt
(bench1.py:164-177), then diagnostics, then append a row.  Default scheme = "fem_be": the reference's own
discretisation and backward-Euler Newton solve on the GPU, one step per row on the committed run's time grid, so the
emitted CSV equals the reference's (<= 1e-8 BM1, <= 1e-6 BM6 in F).  With --scheme fd / spectral each row is reached by
explicit (or semi-implicit spectral) sub-steps of a stable size and rows are emitted at the reference run's accepted
times (pfhubbenchmarks_amd/data/bench<N>_times.txt) so the CSVs still line up row by row.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import io as pio
from .solver import PhaseFieldSolver, stable_dt

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CSV_HEADER = "time,total_free_energy,total_solute"


def report_times(bench):
    return np.loadtxt(os.path.join(_DATA, "%s_times.txt" % bench))


def write_csv(path, rows, header=CSV_HEADER):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    np.savetxt(path, np.array(rows), fmt="%1.10f", header=header, delimiter=",", comments="")


def advance_to(solver, t_target, dt_sub, dt_min):
    """Advance solver.t to t_target with sub-steps <= dt_sub; on a failed guard check restore the row's start state
    and halve dt_sub (bench1.py:164-177).  Returns the dt_sub that worked."""
    start_t = solver.t
    snapshot = None
    while True:
        span = t_target - solver.t
        if span <= 1e-14 * max(1.0, abs(t_target)):
            return dt_sub
        n = int(np.floor(span / dt_sub + 1e-12))
        if snapshot is None:        # the row's start state, whichever of the two step calls below comes first
            snapshot = solver.get_c()
        if n > 0:
            ok, _, _ = solver.step(dt_sub, n, check=True)
        else:
            ok = True
        rem = t_target - solver.t
        if ok and rem > 1e-14 * max(1.0, abs(t_target)):
            ok, _, _ = solver.step(rem, 1, check=True)
        if ok:
            solver.t = t_target
            return dt_sub
        if dt_sub <= dt_min:
            raise RuntimeError("step failed at dt_min = %g (t = %g)" % (dt_min, solver.t))
        dt_sub = max(0.5 * dt_sub, dt_min)
        print("REPEATING row: t = %g, dt_sub -> %g" % (start_t, dt_sub))
        solver.set_c(snapshot)
        solver.t = start_t


def run_bench1(intervals=200, L=200.0, scheme="fd", dt=None, end_time=1e3, times=None, out_dir="results",
               save_solution=False, device=0, verbose=True):
    """PFHub BM1 (dolfin/bench1.py): 200 x 200 no-flux square, c0 = 0.5, epsilon = 0.05 (bench1.py:48-49)."""
    h = L / intervals
    times = report_times("bench1") if times is None else np.asarray(times, dtype=float)
    # bench1.py:145: loop while t < end_time (+eps): the last accepted step overshoots end_time
    keep = [i for i, t in enumerate(times) if i == 0 or times[i - 1] < end_time + 1e-12]
    times = times[keep]
    if dt is None:
        dt = stable_dt(h, dim=2, safety=0.4) if scheme == "fd" else 1e-2
    dt_min = dt / 64.0
    rows, snaps = [], []
    store = None
    if save_solution:          # bench1.py:116-119: HDF5File(..., "results/bench1/conc.h5", "w"); outfile.write(mesh, "mesh")
        store = pio.FieldStore(os.path.join(out_dir, "bench1", "conc.npz"), "w")
        store.write_mesh("grid", h=h, shape=(intervals + 1, intervals + 1), L=L)
    t1 = time.time()
    with PhaseFieldSolver(dim=2, n=intervals + 1, h=h, bc="mirror", scheme=scheme, device=device) as s:
        s.set_ic_bm1(0.5, 0.05)
        for it, tn in enumerate(times):
            dt = advance_to(s, float(tn), dt, dt_min)
            F, C, _ = s.diagnostics()
            rows.append([float(tn), F, C])
            if verbose:
                print("Iteration #%d. Time: %g, C_total: %.10f, TFE: %.10f" % (it + 1, tn, C, F))
            if save_solution:      # bench1.py:190-191: outfile.write(c, "c", float(t)) -- one field per accepted step
                c = s.get_c()
                store.write(c, "c", float(tn))
                snaps.append(os.path.join(out_dir, "bench1", "conc%06d.vti" % it))
                pio.write_vti(snaps[-1], c, h, "c")
    if save_solution:
        store.close()
        pio.write_pvd(os.path.join(out_dir, "bench1", "conc.pvd"), times[:len(snaps)], snaps)
    spent = time.time() - t1
    print("Time spent is %s" % spent)
    write_csv(os.path.join(out_dir, "bench1_out.csv"), rows)
    write_csv(os.path.join(out_dir, "bench1", "stats.csv"), rows)
    return np.array(rows), spent


def run_fem_be(bench="bench1", controller="fixture", end_time=None, out_dir="results", device=0, verbose=True,
               max_rows=None, save_solution=False):
    """BE-parity mode: the reference's own discretisation (100 x 100 'crossed' P1 mesh) and backward-Euler Newton
    solve on the GPU (PF_SCHEME_FEM_BE).  controller = "fixture": one BE step per row on the committed run's time
    grid (rows comparable one to one with results/bench<N>_out.csv); "reference": the script's own rule --
    dt doubles when the solve took < 5 Newton iterations, else halves (bench1.py:180-183), a failed solve halves dt
    and retries (bench1.py:164-177); the iteration counts of a direct-LU Newton differ from the reference's inexact
    SNES/GMRES counts, so that time grid is this solver's own."""
    bm6 = bench == "bench6"
    # per benchmark: domain, intervals (bench<N>.py:21-23), model, end_time, first dt and dt_min of the script's controller
    # (bench1.py:138-141, bench2.py:208-211, bench3.py:192-195, bench6.py:178-181), third CSV column, output fields
    L, N, model, end_default, dt0, dt_min, col3, out_fields = {
        "bench1": (200.0, 100, "bm1", 1e3, 1e-1, 1e-2, "total_solute", ("c",)),
        "bench6": (100.0, 100, "bm6", 3.0, 1e-2, 1e-4, "total_solute", ("c", "phi")),
        "bench2": (200.0, 100, "bm2", 100.0, 1e-2, 1e-4, "total_solute", ("c", "eta1", "eta2", "eta3", "eta4")),
        "bench3": (960.0, 350, "bm3", 100.0, 1e-2, 1e-4, "solid_fraction", ("U", "phi")),
    }[bench]
    header = "time,total_free_energy," + col3
    multi = model in ("bm2", "bm3")
    end_time = end_default if end_time is None else end_time
    rows = []
    store = None
    if save_solution and bench == "bench1":      # bench1.py:116-119 (bench6.py writes PVD files only, :152-153)
        store = pio.FieldStore(os.path.join(out_dir, "bench1", "conc.npz"), "w")
        store.write_mesh("crossed", N=N, L=L)
    t1 = time.time()
    # Newton cap: the reference's 10 (bench1.py:88) under its own controller; the committed time grid with exact linear
    # solves needs up to 24 plain Newton iterations (rows 21, 37), so the fixture controller lifts the cap
    with PhaseFieldSolver(dim=2, n=N + 1, h=L / N, bc="mirror", scheme="fem_be", model=model,
                          device=device, max_newton=100 if controller == "fixture" else 10,
                          # the reference's dt rule reads the Newton iteration count (bench1.py:180-183): under it every
                          # factorisation pivots, so the time grid never depends on the optimistic un-pivoted solve
                          always_pivot=(controller == "reference")) as s:
        {"bm1": s.set_ic_bm1, "bm6": s.set_ic_bm6, "bm2": s.set_ic_bm2, "bm3": s.set_ic_bm3}[model]()
        if controller == "fixture":
            times = report_times(bench)
            times = times[[i for i, t in enumerate(times) if i == 0 or times[i - 1] < end_time + 1e-12]]
            if max_rows:
                times = times[:max_rows]
            tprev = 0.0
            for it, tn in enumerate(times):
                ok, _, _ = s.step(float(tn) - tprev, 1, check=True)
                if not ok:
                    write_csv(os.path.join(out_dir, "%s_out.partial.csv" % bench), rows, header)
                    raise RuntimeError("Newton failed at t = %g (row %d, dt = %g) after %d iterations; the rows so far "
                                       "are in %s_out.partial.csv" % (tn, it, float(tn) - tprev, s.last_iters, bench))
                tprev = float(tn)
                F, C, _ = s.diagnostics()
                rows.append([tprev, F, C])
                if save_solution and multi:   # bench2.py:267-272 / bench3.py:233-236: one PVD series per field
                    for fname in out_fields:
                        stem = "conc" if fname == "c" else fname
                        pio.write_vtu_crossed(os.path.join(out_dir, bench, "%s%06d.vtu" % (stem, it)),
                                              s.get_field(fname), N, L)
                elif save_solution:   # same mesh, node order and PointData layout as the reference's conc00000N.vtu
                    c = s.get_c()
                    pio.write_vtu_crossed(os.path.join(out_dir, bench, "conc%06d.vtu" % it), c, N, L)
                    if store is not None:
                        store.write(c, "c", tprev)                 # bench1.py:190-191
                    if bm6:                                        # bench6.py:227-229: file1 << (phi, t)
                        pio.write_vtu_crossed(os.path.join(out_dir, bench, "phi%06d.vtu" % it), s.get_phi(), N, L)
                if verbose:
                    print("Iteration #%d. Time: %g, niters: %d, C_total: %.10f, TFE: %.10f"
                          % (it + 1, tn, s.last_iters, C, F))
        else:
            t, dt, it = 0.0, dt0, 0
            while t < end_time + 3e-16 and (max_rows is None or len(rows) < max_rows):
                it += 1
                ok, _, _ = s.step(dt, 1, check=True)
                while not ok:
                    dt = max(0.5 * dt, dt_min)
                    if verbose:
                        print("REPEATING Iteration #%d. Time: %g, dt: %g" % (it, t + dt, dt))
                    ok, _, _ = s.step(dt, 1, check=True)
                t += dt
                dt = 2 * dt if s.last_iters < 5 else max(0.5 * dt, dt_min)
                F, C, _ = s.diagnostics()
                rows.append([t, F, C])
                if store is not None:
                    store.write(s.get_c(), "c", t)
                if verbose:
                    print("Iteration #%d. Time: %g, niters: %d, C_total: %.10f, TFE: %.10f" % (it, t, s.last_iters, C, F))
    spent = time.time() - t1
    print("Time spent is %s" % spent)
    if store is not None:
        store.close()
    if save_solution and controller == "fixture" and multi:
        for fname in out_fields:
            stem = "conc" if fname == "c" else fname
            pio.write_pvd(os.path.join(out_dir, bench, stem + ".pvd"), [r[0] for r in rows],
                          ["%s%06d.vtu" % (stem, i) for i in range(len(rows))])
    elif save_solution and controller == "fixture":
        pio.write_pvd(os.path.join(out_dir, bench, "conc.pvd"), [r[0] for r in rows],
                      ["conc%06d.vtu" % i for i in range(len(rows))])
        if bm6:
            pio.write_pvd(os.path.join(out_dir, bench, "phi.pvd"), [r[0] for r in rows],
                          ["phi%06d.vtu" % i for i in range(len(rows))])
    write_csv(os.path.join(out_dir, "%s_out.csv" % bench), rows, header)
    if bench == "bench1":
        write_csv(os.path.join(out_dir, "bench1", "stats.csv"), rows)
    return np.array(rows), spent


def run_bench6(intervals=100, L=100.0, dt=None, end_time=3.0, times=None, out_dir="results", save_solution=False,
               device=0, verbose=True):
    """PFHub BM6 (dolfin/bench6.py): 100 x 100 domain, c0 = 0.5, c1 = 0.04 (bench6.py:53-54), k = 0.09, eps = 90
    (:38-39), phi = 0 / sin(y/7) on x = 0 / Lx (:77-90).  Explicit coupling: each sub-step solves the Poisson
    problem for phi(c^n) (rocFFT) and takes one fused FD Cahn-Hilliard step with mu += k phi."""
    h = L / intervals
    times = report_times("bench6") if times is None else np.asarray(times, dtype=float)
    keep = [i for i, t in enumerate(times) if i == 0 or times[i - 1] < end_time + 1e-12]
    times = times[keep]
    if dt is None:
        dt = stable_dt(h, dim=2, safety=0.4)
    dt_min = dt / 64.0
    rows, snaps = [], []
    t1 = time.time()
    with PhaseFieldSolver(dim=2, n=intervals + 1, h=h, bc="mirror", model="bm6", device=device) as s:
        s.set_ic_bm6(0.5, 0.04)
        for it, tn in enumerate(times):
            dt = advance_to(s, float(tn), dt, dt_min)
            F, C, _ = s.diagnostics()
            rows.append([float(tn), F, C])
            if verbose:
                print("Iteration #%d. Time: %g, C_total: %.10f, TFE: %.10f" % (it + 1, tn, C, F))
            if save_solution:      # bench6.py:227-229: file0 << (c, t); file1 << (phi, t)
                snaps.append((os.path.join(out_dir, "bench6", "conc%06d.vti" % it),
                              os.path.join(out_dir, "bench6", "phi%06d.vti" % it)))
                pio.write_vti(snaps[-1][0], s.get_c(), h, "c")
                pio.write_vti(snaps[-1][1], s.get_phi(), h, "phi")
    if save_solution:
        pio.write_pvd(os.path.join(out_dir, "bench6", "conc.pvd"), times[:len(snaps)], [a for a, _ in snaps])
        pio.write_pvd(os.path.join(out_dir, "bench6", "phi.pvd"), times[:len(snaps)], [b for _, b in snaps])
    spent = time.time() - t1
    print("Time spent is %s" % spent)
    write_csv(os.path.join(out_dir, "bench6_out.csv"), rows)
    return np.array(rows), spent


def main_bench6(argv=None):
    ap = argparse.ArgumentParser(description="PFHub BM6 on MI355X (counterpart of dolfin/bench6.py)")
    ap.add_argument("--intervals", type=int, default=100, help="grid intervals per side (h = 100/intervals)")
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--end-time", type=float, default=3.0)
    ap.add_argument("--scheme", default="fem_be", choices=["fd", "fem_be"],
                    help="fem_be (default): the reference's own P1 backward-Euler discretisation on the GPU -- the CSV "
                         "matches the reference's committed results/bench6_out.csv to <= 1e-6; fd: explicit finite "
                         "differences + FFT Poisson solve (the throughput scheme; its own discretisation error)")
    ap.add_argument("--controller", default="fixture", choices=["fixture", "reference"])
    ap.add_argument("--out-dir", default="results")
    ap.add_argument("--save-solution", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    if a.scheme == "fem_be":
        run_fem_be("bench6", a.controller, a.end_time, a.out_dir, 0, not a.quiet, None, a.save_solution)
        return
    run_bench6(a.intervals, 100.0, a.dt, a.end_time, None, a.out_dir, a.save_solution, 0, not a.quiet)


def run_b13d(intervals=50, L=100.0, dt=None, end_time=50.0, out_dir="results", device=0, verbose=True):
    """3-D BM1 (dolfin/b13d.py): 100^3 no-flux box, 50^3 cells (b13d.py:24-26), the 2-D initial condition extruded in z
    (b13d.py:55 with pfbase.py:187-189 using x[0], x[1] only), end_time = 50 (b13d.py:125).  Rows at the accepted
    times of the committed 2-D run up to end_time.  The reference script writes results/bench1_out.csv
    (b13d.py:199-206); to keep the 2-D trajectory this driver writes results/b13d_out.csv."""
    h = L / intervals
    times = report_times("bench1")
    times = times[[i for i, t in enumerate(times) if i == 0 or times[i - 1] < end_time + 1e-12]]
    if dt is None:
        dt = stable_dt(h, dim=3, safety=0.4)
    rows = []
    t1 = time.time()
    with PhaseFieldSolver(dim=3, n=intervals + 1, h=h, bc="mirror", device=device) as s:
        s.set_ic_bm1(0.5, 0.05)
        for it, tn in enumerate(times):
            dt = advance_to(s, float(tn), dt, dt / 64.0)
            F, C, _ = s.diagnostics()
            rows.append([float(tn), F, C])
            if verbose:
                print("Iteration #%d. Time: %g, C_total: %.10f, TFE: %.10f" % (it + 1, tn, C, F))
    spent = time.time() - t1
    print("Time spent is %s" % spent)
    write_csv(os.path.join(out_dir, "b13d_out.csv"), rows)
    return np.array(rows), spent


def run_multi_fd(bench, end_time=None, out_dir="results", device=0, verbose=True, intervals=None, dt=None,
                 max_rows=None):
    """BM2 / BM3 with the explicit multi-field FD scheme (csrc/multi_fd.hip): rows at the accepted times of the reference's
    committed run, each reached by sub-steps of a stable size (the throughput counterpart of the BE-parity default, as
    `--scheme fd` is for bench1 / bench6)."""
    from .verification import L_DOM, N_REF, multi_fd_dt
    model = {"bench2": "bm2", "bench3": "bm3"}[bench]
    N = intervals or (2 * N_REF[model] if model == "bm2" else N_REF[model])       # BM2: h = 1, BM3: the reference's h
    L = L_DOM[model]
    h = L / N
    end_time = 100.0 if end_time is None else end_time
    times = report_times(bench)
    times = times[[i for i, t in enumerate(times) if i == 0 or times[i - 1] < end_time + 1e-12]]
    if max_rows:
        times = times[:max_rows]
    dt = multi_fd_dt(model, h) if dt is None else dt
    header = "time,total_free_energy," + ("total_solute" if model == "bm2" else "solid_fraction")
    rows = []
    t1 = time.time()
    with PhaseFieldSolver(dim=2, n=N + 1, h=h, bc="mirror", scheme="fd", model=model, device=device) as s:
        (s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3)()
        tprev = 0.0
        for it, tn in enumerate(times):
            n = max(1, int(np.ceil((float(tn) - tprev) / dt - 1e-12)))
            ok, _, _ = s.step((float(tn) - tprev) / n, n, check=True)
            if not ok:
                raise RuntimeError("%s fd: the state left the guard band before t = %g (dt = %g)" % (bench, tn, dt))
            tprev = float(tn)
            F, C, _ = s.diagnostics()
            rows.append([tprev, F, C])
            if verbose:
                print("Iteration #%d. Time: %g, second: %.10f, TFE: %.10f" % (it + 1, tn, C, F))
    spent = time.time() - t1
    print("Time spent is %s" % spent)
    write_csv(os.path.join(out_dir, "%s_out.csv" % bench), rows, header)
    return np.array(rows), spent


def _main_multi(bench, desc, argv=None):
    ap = argparse.ArgumentParser(description=desc)
    ap.add_argument("--scheme", default="fem_be", choices=["fem_be", "fd"],
                    help="fem_be (default): the reference's own discretisation -- the CSV matches the reference's committed one; "
                         "fd: explicit finite differences (the throughput scheme; its own discretisation error)")
    ap.add_argument("--intervals", type=int, default=None, help="--scheme fd: grid intervals per side")
    ap.add_argument("--controller", default="fixture", choices=["fixture", "reference"],
                    help="fixture: the committed run's time grid; reference: the script's own dt rule")
    ap.add_argument("--end-time", type=float, default=None)
    ap.add_argument("--max-rows", type=int, default=None)
    ap.add_argument("--out-dir", default="results")
    ap.add_argument("--save-solution", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    if a.scheme == "fd":
        run_multi_fd(bench, a.end_time, a.out_dir, 0, not a.quiet, a.intervals, None, a.max_rows)
        return
    run_fem_be(bench, a.controller, a.end_time, a.out_dir, 0, not a.quiet, a.max_rows, a.save_solution)


def main_bench2(argv=None):
    _main_multi("bench2", "PFHub BM2 (Ostwald ripening) on MI355X: the reference's P1 backward-Euler discretisation "
                          "(counterpart of dolfin/bench2.py)", argv)


def main_bench3(argv=None):
    _main_multi("bench3", "PFHub BM3 (dendritic growth) on MI355X: the reference's P1 backward-Euler discretisation "
                          "(counterpart of dolfin/bench3.py)", argv)


def main_b13d(argv=None):
    ap = argparse.ArgumentParser(description="3-D PFHub BM1 on MI355X (counterpart of dolfin/b13d.py)")
    ap.add_argument("--intervals", type=int, default=50)
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--end-time", type=float, default=50.0)
    ap.add_argument("--out-dir", default="results")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    run_b13d(a.intervals, 100.0, a.dt, a.end_time, a.out_dir, 0, not a.quiet)


def main_bench1(argv=None):
    ap = argparse.ArgumentParser(description="PFHub BM1 on MI355X (counterpart of dolfin/bench1.py)")
    ap.add_argument("--intervals", type=int, default=200, help="grid intervals per side (h = 200/intervals)")
    ap.add_argument("--scheme", default="fem_be", choices=["fd", "spectral", "fem_be"],
                    help="fem_be (default): the reference's own P1 backward-Euler discretisation on the GPU -- the CSV "
                         "matches the reference's committed results/bench1_out.csv to <= 1e-8 in every row; fd / spectral: "
                         "the throughput schemes bench.py measures (they converge to the same solution, not to the "
                         "committed run's backward-Euler error: results/CONVERGENCE.md)")
    ap.add_argument("--controller", default="fixture", choices=["fixture", "reference"])
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--end-time", type=float, default=1e3)
    ap.add_argument("--out-dir", default="results")
    ap.add_argument("--save-solution", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    if a.scheme == "fem_be":
        run_fem_be("bench1", a.controller, a.end_time, a.out_dir, 0, not a.quiet, None, a.save_solution)
        return
    run_bench1(a.intervals, 200.0, a.scheme, a.dt, a.end_time, None, a.out_dir, a.save_solution, 0, not a.quiet)
