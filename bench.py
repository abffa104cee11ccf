#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the PFHub BM1 Cahn-Hilliard hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--variant V]

A "step" is one explicit FD Cahn-Hilliard update of the whole grid (the fused HIP kernel of
pfhubbenchmarks_amd/csrc/ch_fd_kernels.hip).  Workloads:
  bm1_fd_512c   (default, BASELINE.json config 3)  512^3 per GPU, fp64, BM1 initial condition extruded in z;
                N > 1: each rank owns a 512 x 512 x 512 slab of a 512 x 512 x (512 N) periodic box (weak scaling),
                ghost planes exchanged over RCCL (torch.distributed "nccl") overlapped with the interior kernel; default
                --halo wide = PF_FLAG_WIDE_HALO: 4 ghost planes every second step (5.7 % instead of 11.8 % slab-path overhead
                on one GPU, bit-identical; --halo narrow = 2 ghost planes every step).
  bm1_fd_1024c  1024^3 on 1 GPU, or 1024 x 1024 x (1024/N) slabs on N GPUs (BASELINE.json config 4, strong)
  bm1_fd_512s   512^2 2-D (launch-latency bound; reported for completeness)
  bm1_spectral_512s / _256c / _512c   semi-implicit spectral scheme (BASELINE.json config 2); _512c also runs on N > 1
                GPUs: ONE 512^3 box in z-slabs (strong scaling; slab FFT: one RCCL all-to-all each way per transform);
                _512c_weak: 512 x 512 x 512 N
  bm6_fd_512c / _256c   BM6 (BASELINE.json config 5: "512^3, 8 x MI355X") in a periodic box: FFT Poisson solve + coupled
                fused FD step per step; N > 1: the 512^3 box in z-slabs (strong), slab FFT Poisson (2 all-to-alls) + ghost
                exchange of c and phi; bm6_fd_512c_weak: 512 x 512 x 512 N
  bm6_spectral_512c     BM6 with the semi-implicit spectral scheme (phi eliminated in Fourier space; 1 GPU)
  bm6_fd_512c_elim      the same physics with phi eliminated (lap_h(k phi) = -(k^2/eps)(c - mean c) exactly): the step is
                the fused kernel alone, no transform in the time loop (phi is solved only for diagnostics)
  bm2_fd_512c / bm3_fd_512c   the multi-field models (dolfin/bench2.py, bench3.py) by explicit FD on the stencil design: one-pass
                LDS-tiled BM2 kernel (80 B/cell-update), streaming BM3 kernel (32 B/cell-update); 1 GPU
  bm2_fem_be / bm3_fem_be     the same models in the BE-parity mode (the reference's own discretisation), node-updates/s
  bm1_fem_be    BASELINE.json config 1: the reference's own algorithm (100x100 crossed P1 mesh, backward Euler, Newton)
                on the GPU; a "step" is one accepted BE step of the committed run's time grid; metric node-updates/s;
                cpu_baseline = oracle/fem_be.py (numpy/scipy SuperLU) on the same rows; says whether FEniCS is present
Why config 3 and not config 2 is the default: BASELINE.json's metric is "cell-updates/s at 512^2 and 512^3, 1/2/4/8
GPUs" and its roofline target is the fused stencil.  Config 2 (512^2 spectral) is a 14 us, launch-latency-bound step
that does not shard (replicas only), so the N = 1, 2, 4, 8 series is run on the 512^3 stencil;
the same default run also times both 512^2 workloads, the 1024^3 stencil (north_star's roofline target), the 512^3
spectral step, BM6 at 512^3, BM2 / BM3 and the BE-parity mode with its CPU restatement, and reports them under "also" in
the same JSON line -- a COMPACT line (<= 6 KB; the driver keeps the last 8 KB of stdout), the BASELINE-config entries last;
--verbose restores every timed block, pf_status_string and the notes.  CPU legs run after all GPU legs.
Timed region (DESIGN.md section 6 "power transient"): the chip answers an HBM-heavy load that starts from
idle (>= 3-10 ms without work) by dropping its shader clock from 2.4 to ~1.7 GHz for ~25 ms (tools/ramp_probe.py,
profiles/r02/ramp_probe_512.log), which a 20-step / 9 ms timed region sits entirely inside.  So the run first does a
DECLARED, untimed pre-heat of the same step by wall time (--preheat-s, default 0.5 s; reported as preheat_ms /
preheat_steps), then the W counted warm-up steps, then times EXACTLY K steps between device syncs (+ barrier); when
K steps take less than 0.25 s that K-step block is repeated (`repeats`, at most 25) and the MEDIAN block is reported
(all block times are in `block_ms_per_step`).  --preheat-s 0 --repeats 1 gives the raw cold measurement.
Prints ONE JSON line (rank 0).  `value` counts the cell updates of all ranks; inputs are resident in HBM before the
timed region.  roofline.achieved = 16 B/cell-update x cells per launch / average kernel time from HIP events
recorded inside libpfhip around every step launch.  cpu_baseline = the CPU oracle (oracle/ch_fd.c, OpenMP) timed on
this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_UPDATE = 16.0     # read c^n once + write c^{n+1} once, fp64
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy rate


def cpu_baseline(nx, ny, nz_sample, dt, steps):
    """The oracle (port of the same algorithm) on the host cores: bounded sample = an nx x ny x nz_sample periodic
    block of the same initial condition."""
    import numpy as np
    from oracle import ch_fd
    ch_fd.load()
    threads = ch_fd.set_threads(int(os.environ.get("OMP_NUM_THREADS", "0")))   # default: the cores we really own
    c = ch_fd.ic(nx, ny, 1)
    c = np.repeat(c, nz_sample, 0)
    c = ch_fd.fd_step(c, dt)          # warm-up (page faults, OpenMP pool)
    t0 = time.perf_counter()
    for _ in range(steps):
        c = ch_fd.fd_step(c, dt)
    el = time.perf_counter() - t0
    return {"value": nx * ny * nz_sample * steps / el, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": "%dx%dx%d periodic block of the BM1 IC, %d steps, oracle/ch_fd.c (gcc -O2 -fopenmp), %.1f s"
                      % (nx, ny, nz_sample, steps, el)}


def cpu_baseline_spectral(n, dt, steps):
    """numpy (pocketfft, one thread) restatement of the same semi-implicit spectral step on the same grid."""
    from oracle import ch_fd, ch_spectral
    c = ch_fd.ic(n[0], n[1], n[2] if len(n) == 3 else 1)
    c = c if len(n) == 3 else c[0]
    sp = ch_spectral.SpectralCH(c, h=1.0)
    sp.step(dt, 2)
    t0 = time.perf_counter()
    sp.step(dt, steps)
    el = time.perf_counter() - t0
    cells = 1
    for v in n:
        cells *= v
    return {"value": cells * steps / el, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": "%s grid, %d steps, oracle/ch_spectral.py (numpy pocketfft), %.1f s" % ("x".join(map(str, n)),
                                                                                             steps, el)}


def _fem_timed_rows(s, set_ic, times, warmup, steps):
    """The timed region of the BE-parity workloads.  One UNTIMED pass over the same rows first (declared pre-heat: the first
    solves of a handle pay rocBLAS / code-object initialisation, and a GPU that idled during another workload's CPU leg
    starts at a reduced shader clock, DESIGN 6.1 -- round 3's driver line read 20.6 ms per BM1 step where the steady state
    is 9.4), then the state is reset to the initial condition and rows [warmup, warmup + steps) are timed.
    -> (seconds, Newton iterations, t_end, preheat_ms)"""
    import torch
    el = its = tprev = pre_ms = None
    for timed in (False, True):
        t_pass = time.perf_counter()
        set_ic()
        tprev, its = 0.0, 0
        for i in range(warmup):
            s.step(times[i] - tprev, 1, check=True)
            tprev = times[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            ok, _, _ = s.step(times[i] - tprev, 1, check=True)
            assert ok
            its += s.last_iters
            tprev = times[i]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if not timed:
            pre_ms = (time.perf_counter() - t_pass) * 1e3
    return el, its, tprev, pre_ms


def bench_fem_be(a, world, steps=None, warmup=None, ncpu_max=6):
    """config 1 (reference's own discretisation): accepted backward-Euler steps on the committed time grid.
    Returns the JSON dict (the caller prints it, or nests it under `also`); out["_cpu"] (when the CPU leg is wanted) is a
    callable that times oracle/fem_be.py on the same rows and fills out["cpu_baseline"] -- the caller runs it AFTER every
    GPU leg, so that no GPU measurement starts from a GPU that sat idle behind a CPU leg."""
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    import importlib.util
    from pfhubbenchmarks_amd.drivers import report_times
    from pfhubbenchmarks_amd.solver import PhaseFieldSolver
    if world != 1:
        sys.exit("bm1_fem_be is a single-GPU workload")
    times = report_times("bench1")
    steps = min(steps, len(times) - warmup)
    nodes = 20201
    with PhaseFieldSolver(dim=2, n=101, h=2.0, bc="mirror", scheme="fem_be", max_newton=100) as s:
        el, its, tprev, pre_ms = _fem_timed_rows(s, s.set_ic_bm1, times, warmup, steps)
        F, C, _ = s.diagnostics()
        status = getattr(s, "status", "")
    out = {"metric": "node-updates/sec on PFHub BM1, reference algorithm (P1 crossed mesh, backward Euler, Newton)",
           "value": nodes * steps / el, "unit": "node-updates/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "preheat_ms": pre_ms,
           "config": {"workload": "bm1_fem_be", "mesh": "100x100 crossed, 20201 nodes, 40402 dofs",
                      "time_grid": "rows %d..%d of results/bench1_out.csv" % (warmup, warmup + steps - 1),
                      "newton_iterations": its, "preheat": "one untimed pass over the same rows, then reset to the IC",
                      "linear_solver": "cell-centre unknowns condensed out, banded first reduction level, then block "
                                       "cyclic reduction on own LU / substitution kernels", "status": status},
           "roofline": None,
           "check": {"t": float(tprev), "F": F, "C": C},
           # (this repo has its own dolfin/ directory of command-line shims, so probe FEniCS's dependencies instead)
           "fenics_on_host": all(importlib.util.find_spec(m) is not None for m in ("ufl", "ffc", "petsc4py"))}

    def cpu_leg():
        from oracle import ch_fd, fem_be
        o = fem_be.FemBE("bm1", newton_max=100)
        tp = 0.0
        for i in range(warmup):
            o.step(times[i] - tp)
            tp = times[i]
        ncpu = min(steps, ncpu_max)
        t0 = time.perf_counter()
        for i in range(warmup, warmup + ncpu):
            o.step(times[i] - tp)
            tp = times[i]
        elc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nodes * ncpu / elc, "unit": "node-updates/s", "cores": ch_fd.host_cores(),
                               "kind": "port", "sample": "%d accepted BE steps (rows %d..%d), oracle/fem_be.py "
                               "(scipy SuperLU), %.1f s; FEniCS itself: %s" % (
                                   ncpu, warmup, warmup + ncpu - 1, elc,
                                   "present" if out["fenics_on_host"] else "unavailable on host")}
    if not a.no_cpu_baseline:
        out["_cpu"] = cpu_leg
    return out


def bench_fem_multi(a, model, steps=6, warmup=2, cpu=True):
    """BM2 / BM3 in the BE-parity mode: accepted backward-Euler steps of the reference's own
    discretisation (dolfin/bench2.py:76-113 on the 100 x 100 crossed mesh, 6 fields; bench3.py:63-97 on 350 x 350, 2
    fields) on the committed time grid; node-updates/s.  CPU leg (deferred like bench_fem_be's): ONE Newton iteration of
    oracle/fem_multi.py (scipy SuperLU; a whole step costs it 100-200 s here), extrapolated to the iterations the timed
    steps took -- for BM3 on a 120 x 120 mesh, scaled by nodes; kind = "port-extrapolated"."""
    from pfhubbenchmarks_amd.drivers import report_times
    from pfhubbenchmarks_amd.solver import PhaseFieldSolver
    bench = "bench2" if model == "bm2" else "bench3"
    N, L_dom = (100, 200.0) if model == "bm2" else (350, 960.0)
    times = report_times(bench)
    nodes = (N + 1) ** 2 + N * N
    nf = 6 if model == "bm2" else 2
    with PhaseFieldSolver(dim=2, n=N + 1, h=L_dom / N, bc="mirror", scheme="fem_be", model=model, max_newton=100) as s:
        el, its, tprev, pre_ms = _fem_timed_rows(s, s.set_ic_bm2 if model == "bm2" else s.set_ic_bm3, times, warmup, steps)
        F, C, _ = s.diagnostics()
        status = getattr(s, "status", "")
    ref_wall = {"bm2": "22 s on 32 cores for the whole 120-row run (dolfin/bench2.py:140 comment)",
                "bm3": "27 s on 128 cores for the whole 46-row run (dolfin/bench3.py:125 comment)"}[model]
    out = {"metric": "node-updates/sec on PFHub %s, reference algorithm (P1 crossed mesh, backward Euler, Newton)" % model.upper(),
           "value": nodes * steps / el, "unit": "node-updates/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "warmup": warmup, "dtype": "f64", "preheat_ms": pre_ms,
           "config": {"workload": "%s_fem_be" % model, "mesh": "%dx%d crossed, %d nodes x %d fields = %d dofs" % (N, N, nodes, nf, nodes * nf),
                      "time_grid": "rows %d..%d of results/%s_out.csv" % (warmup, warmup + steps - 1, bench),
                      "newton_iterations": its, "reference_wall_time": ref_wall,
                      "preheat": "one untimed pass over the same rows, then reset to the IC", "status": status},
           "check": {"t": float(tprev), "F": F, "second_column": C}}

    def cpu_leg():
        from oracle import ch_fd, fem_multi
        Ns = N if model == "bm2" else 120
        o = fem_multi.MultiFieldBE(model, N=Ns, newton_max=1)
        t0 = time.perf_counter()
        o.step(times[0])                       # exactly one Newton iteration: residual, Jacobian, sparse LU, solve
        t_it = time.perf_counter() - t0
        nodes_s = (Ns + 1) ** 2 + Ns * Ns
        out["cpu_baseline"] = {"value": nodes_s * steps / (t_it * its), "unit": "node-updates/s", "cores": ch_fd.host_cores(),
                               "kind": "port-extrapolated",
                               "sample": "1 Newton iteration (%.1f s) of oracle/fem_multi.py (scipy SuperLU) on the %dx%d "
                                         "crossed mesh, extrapolated to the %d iterations of the %d timed steps" % (
                                             t_it, Ns, Ns, its, steps)}
    if cpu:
        out["_cpu"] = cpu_leg
    return out


def run_cpu_legs(*outs):
    """Run the deferred CPU legs (after every GPU measurement of the process) and drop the callables."""
    for o in outs:
        leg = o.pop("_cpu", None)
        if leg is not None:
            leg()


def measured_traffic(workload, variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in
    separate runs, FETCH_SIZE x2 per MI355X_MICROARCH.md) of THIS workload with the default kernel variant;
    None when no such profile is committed."""
    if variant >= 0:
        return None
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, rnd, "summary_%s.json" % workload)
        if os.path.exists(f):
            with open(f) as fh:
                best = json.load(fh)
    return None if best is None else best.get("traffic_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 200 (bm1_fem_be: 100)")
    ap.add_argument("--warmup", type=int, default=None, help="counted untimed steps after the pre-heat; default 20 "
                                                             "(bm1_fem_be: 10)")
    ap.add_argument("--preheat-s", type=float, default=0.5,
                    help="declared untimed pre-heat of the same step, by wall time, before the counted warm-up: a load "
                         "starting from an idle GPU runs ~25 ms at a reduced shader clock (tools/ramp_probe.py)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed K-step blocks (median reported); 0 = as many as needed for 0.25 s of timed work, <= 25")
    ap.add_argument("--no-also", action="store_true", help="default workload only: skip the side measurements")
    ap.add_argument("--verbose", action="store_true",
                    help="keep the bulky bookkeeping in the JSON line (every timed block, pf_status_string, notes, the sources "
                         "of the traffic figures).  The default line is compact (<= 6 KB) so that the driver's 8 KB tail of "
                         "stdout holds ALL of it: round 3's 15 KB line lost five side measurements that way")
    ap.add_argument("--halo", default="wide", choices=["wide", "narrow"],
                    help="FD slab path (N > 1 or --slab): 'wide' = PF_FLAG_WIDE_HALO, 4 ghost planes exchanged every second "
                         "step (half the hand-offs, 8 instead of 12 redundant plane reads per step); 'narrow' = 2 ghost "
                         "planes every step.  Bit-identical results.")
    ap.add_argument("--strips", default="inline", choices=["side", "inline"],
                    help="FD slab path over RCCL: 'side' = the boundary strips of a step run on a stream of their own "
                         "(pf_set_strip_stream: interior launch || exchange -> strips; the exchange wait leaves the critical "
                         "path); 'inline' = behind the interior launch on the compute stream (round-2 form).  Bit-identical.")
    ap.add_argument("--workload", default="bm1_fd_512c", choices=["bm1_fd_512c", "bm1_fd_1024c", "bm1_fd_512s", "bm1_spectral_512s", "bm1_spectral_256c", "bm1_spectral_1024c",
                             "bm1_spectral_512c", "bm1_spectral_512c_weak", "bm6_spectral_512c", "bm6_fd_512c", "bm6_fd_512c_weak", "bm6_fd_256c", "bm6_fd_512c_elim", "bm1_fem_be", "bm2_fem_be", "bm3_fem_be",
                             "bm2_fd_512c", "bm3_fd_512c"])
    ap.add_argument("--variant", type=int, default=-1, help="fused-kernel variant (pfk_set_tuning key 0)")
    ap.add_argument("--kernel", default="fused", choices=["fused", "twopass"])
    ap.add_argument("--target-wgs", type=int, default=0, help="pfk_set_tuning key 1")
    ap.add_argument("--min-chunk", type=int, default=0, help="pfk_set_tuning key 2")
    ap.add_argument("--fused-slab", action="store_true",
                    help="with --transport ipc: one launch per step (pf_step_slab_fused)")
    ap.add_argument("--push-wgs", type=int, default=0, help="pfk_set_tuning key 7 (ipc transport)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slab", action="store_true",
                    help="N = 1 only: run the multi-GPU code path (RCCL process group of size 1, ghost planes, overlapped "
                         "exchange with itself) to measure its overhead against the plain single-GPU path")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "ipc"],
                    help="ghost-plane exchange of the FD slab path: RCCL send/recv (default) or peer-mapped ghost planes "
                         "written by a side-stream kernel (pfhubbenchmarks_amd.solver.IpcHaloTransport; one node)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 100 if a.workload == "bm1_fem_be" else (20 if a.workload.endswith("_fem_be") else 200)
    if a.warmup is None:
        a.warmup = 10 if a.workload == "bm1_fem_be" else (2 if a.workload.endswith("_fem_be") else 20)

    import torch
    from pfhubbenchmarks_amd import lib as L
    from pfhubbenchmarks_amd.solver import HipSlabEngine, PhaseFieldSolver, SlabSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # REHEARSAL ONLY (tests/test_gpu_parity.py): run the N > 1 launch contract on a 1-GPU box -- every rank on device 0,
    # gloo instead of RCCL (which refuses two ranks on one device).  The numbers of such a run mean nothing.
    rehearsal = os.environ.get("PFHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # one command, like the reference's `mpirun -np N python dolfin/bench1.py` (README.md:22): start the one-process-
            # per-GPU job ourselves -- as a CHILD process, before this process touches the GPU -- relay its output (rank 0
            # prints the JSON line) and exit with its return code
            sys.exit(spawn_ranks(a.gpus))
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    lib = L.load()                      # raises if the HIP extension is missing: no fallback
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    if a.variant >= 0:
        lib.pfk_set_tuning(0, a.variant)
    if a.target_wgs > 0:
        lib.pfk_set_tuning(1, a.target_wgs)
    if a.min_chunk > 0:
        lib.pfk_set_tuning(2, a.min_chunk)
    if a.push_wgs > 0:
        lib.pfk_set_tuning(7, a.push_wgs)

    if a.workload == "bm1_fem_be":
        o = bench_fem_be(a, world)
        run_cpu_legs(o)
        print(json.dumps(o if a.verbose else compact_line(o)), flush=True)
        return
    if a.workload in ("bm2_fem_be", "bm3_fem_be"):
        if world != 1:
            sys.exit("%s is a single-GPU workload" % a.workload)
        o = bench_fem_multi(a, a.workload[:3], steps=min(a.steps, 40), warmup=min(a.warmup, 4), cpu=not a.no_cpu_baseline)
        o.update(n_gpus=1, higher_is_better=True, scaling="weak", vs_baseline=None, data="synthetic", roofline=None)
        run_cpu_legs(o)
        print(json.dumps(o if a.verbose else compact_line(o)), flush=True)
        return

    dist = None
    if world > 1 or a.slab:
        import torch.distributed as dist
        if a.slab and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctx = {"world": world, "rank": rank, "local_rank": local_rank, "rehearsal": rehearsal, "dist": dist, "lib": lib}
    out = bench_grid(a, a.workload, ctx, a.steps, a.warmup, cpu=not a.no_cpu_baseline, copy_ceiling=True)
    if rank == 0 and world == 1 and a.workload == "bm1_fd_512c" and not a.no_also:
        out["also"] = side_measurements(a, ctx)
    # every GPU measurement is done: now the CPU legs (oracle timed on the host cores), headline first
    run_cpu_legs(out, *out.get("also", {}).values())
    if rank == 0:
        print(json.dumps(out if a.verbose else compact_line(out)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --standalone --nnodes=1
    --nproc-per-node N --local-addr 127.0.0.1 bench.py <same arguments>` as a child process and return its exit code.
    --standalone lets the launcher bind its own free rendezvous port (a port probed here and passed on could be taken by
    another process in between)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node", str(n),
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench.py] --gpus %d without a launcher: starting %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


# ---- the compact JSON line -------------------------------------------------------------------------------------------
# The driver keeps the last 8 KB of stdout.  Everything a reader needs to check a number stays (value, time per step,
# steps, steady, roofline {frac, achieved, traffic, bytes per cell update}, check, workload); the bookkeeping (every timed
# block, pf_status_string, notes, provenance strings) is --verbose only.
_ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_cell_update", "kernel_ms_per_step",
              "frac_hip_events")


def _r(x, sig=6):
    """floats at `sig` significant digits (the line is a report, not a checkpoint)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, list):
        return [_r(v, sig) for v in x]
    return x


def compact_entry(e, main=False):
    """one workload's dict -> its compact form (the main line keeps the contract keys, an `also` entry only what differs)"""
    keep = ["value", "unit", "ms_per_step", "us_per_step", "steps", "steady"]   # (an `also` entry)
    if main:
        keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "steady", "repeats", "preheat_ms", "fenics_on_host"]
    out = {k: e[k] for k in keep if k in e}
    cfg = e.get("config", {})
    ck = ("workload", "grid", "parallelism", "newton_iterations", "time_grid", "field_store", "transforms") if main else \
         ("workload", "newton_iterations", "transforms")
    out["config"] = {k: cfg[k] for k in ck if k in cfg}
    if main and cfg.get("status"):
        out["config"]["status"] = cfg["status"][:320]      # which kernels / transform path ran (the N > 1 FFT modes say so here)
    rf = e.get("roofline")
    if rf is not None or main:
        out["roofline"] = None if rf is None else {k: rf[k] for k in (_ROOF_KEEP if main else _ROOF_KEEP[1:2] + _ROOF_KEEP[4:7])
                                                   if k in rf}
        if rf is not None and main and "device_copy" in rf:
            out["roofline"]["device_copy_GBps"] = rf["device_copy"]["achieved"]
    if "check" in e:
        out["check"] = _r(e["check"], 12)
    if "cpu_baseline" in e:
        cb = e["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in (("value", "unit", "cores", "kind", "sample") if main else
                                                  ("value", "cores", "kind"))}
    if "ranks" in e and main:
        rk = e["ranks"]
        out["ranks"] = rk if rk["world_size"] > 1 else {"world_size": 1, "backend": rk["backend"],
                                                        "device_name": rk.get("device_name")}
    return out


def compact_line(out):
    line = _r(compact_entry(out, main=True), 7)
    if "check" in out:
        line["check"] = _r(out["check"], 12)
    if "also" in out:
        line["also"] = {k: _r(compact_entry(v), 6) for k, v in out["also"].items()}
        for k, v in out["also"].items():
            if "check" in v:
                line["also"][k]["check"] = _r(v["check"], 10)
    return line


def workload_table(workload, world):
    """-> dict(scheme, model, elim, bytes_per_cell, dim, gn, scaling, dt)"""
    w = {"scheme": "fd", "model": "bm1", "elim": False, "bytes_per_cell": BYTES_PER_CELL_UPDATE, "dim": 3,
         "scaling": "weak", "dt": 5e-4}
    if workload == "bm6_fd_512c_elim":
        # BM6 with phi eliminated algebraically (PF_FLAG_BM6_ELIMINATE_PHI): the step is the fused CH kernel alone
        w.update(model="bm6", elim=True, gn=(512, 512, 512 * world))
    elif workload in ("bm6_fd_512c", "bm6_fd_256c", "bm6_fd_512c_weak"):
        # CH kernel 16 B + phi read 8 B + Poisson transform pair idealised at 48 B (r2c 16, invert 16, c2r 16).
        # N > 1: BASELINE.json config 5 verbatim -- ONE 512^3 box split into z-slabs over the N ranks (strong scaling, the
        # reference's `mpirun -np N` on one problem, README.md:22): every axis stays within the hand-written passes' 128..1024
        # points.  _weak: 512 x 512 x 512 N (the slab engine's config.status says which transform path runs: beyond 1024
        # planes it is rocFFT + pack / unpack); _256c keeps the weak form (a rehearsal size)
        nn = 256 if workload.endswith("256c") else 512
        weak = not workload.endswith("512c")
        w.update(model="bm6", bytes_per_cell=72.0, gn=(nn, nn, nn * world if weak else nn), scaling="weak" if weak else "strong")
    elif workload == "bm6_spectral_512c":
        # BM6 with the spectral scheme: phi eliminated in Fourier space, same passes as bm1_spectral_512c (one GPU)
        w.update(scheme="spectral", model="bm6", bytes_per_cell=72.0, gn=(512, 512, 512), dt=1e-2)
        if world > 1:
            sys.exit("bm6_spectral_512c is single-GPU")
    elif workload in ("bm1_spectral_512c", "bm1_spectral_512c_weak"):
        weak = workload.endswith("_weak")   # default: one 512^3 box over the N ranks (strong), as bm6_fd_512c above
        w.update(scheme="spectral", bytes_per_cell=72.0, gn=(512, 512, 512 * world if weak else 512), dt=1e-2,
                 scaling="weak" if weak else "strong")
    elif workload == "bm1_spectral_1024c":
        w.update(scheme="spectral", bytes_per_cell=72.0, gn=(1024, 1024, 1024), dt=1e-2)
        if world > 1:
            sys.exit("bm1_spectral_1024c is single-GPU")
    elif workload in ("bm1_spectral_512s", "bm1_spectral_256c"):
        # BASELINE.json config 2: semi-implicit spectral; 72 B/cell-update = one-pass-per-transform idealisation
        # (f' 16 + r2c 16 + k-space 24 + c2r 16) -- 512^2 is launch-latency bound, not HBM bound
        w.update(scheme="spectral", bytes_per_cell=72.0, dt=1e-2)
        w.update(dim=2, gn=(512, 512, 1)) if workload.endswith("512s") else w.update(gn=(256, 256, 256))
        if world > 1:
            sys.exit("this 2-D / small spectral workload is single-GPU; use bm1_spectral_512c for N > 1")
    elif workload in ("bm2_fd_512c", "bm3_fd_512c"):
        # SURVEY 8f next-4 on the stencil design: explicit FD for the multi-field models (dolfin/bench2.py:76-113,
        # bench3.py:63-97), streaming LDS-tiled kernels (csrc/multi_fd.hip).  Algorithmic bytes per cell-update = every field
        # read once and written once: BM2 5 fields = 80 B, BM3 2 fields = 32 B (BM2's separate mu pass moves 136 B: the
        # roofline figure is still quoted against 80)
        # N > 1 (or --slab): a ring of 512 x 512 x 512 slabs (weak scaling), every field's ghost planes refreshed before each
        # step (MultiFieldSlabSolver; no interior / boundary overlap)
        if workload.startswith("bm2"):
            w.update(model="bm2", bytes_per_cell=80.0, gn=(512, 512, 512 * world), dt=2e-4)
        else:
            w.update(model="bm3", bytes_per_cell=32.0, gn=(512, 512, 512 * world), dt=2e-3)
    elif workload == "bm1_fd_512s":
        w.update(dim=2, gn=(512, 512, 1), dt=1e-3)
    elif workload == "bm1_fd_512c":
        w.update(gn=(512, 512, 512 * world))
    elif workload == "bm1_fd_1024c":
        w.update(gn=(1024, 1024, 1024), scaling="strong")
    else:
        sys.exit("unknown workload %r" % workload)
    return w


def eng_planes(nzg, world, rank):
    from pfhubbenchmarks_amd.solver import slab_partition
    return slab_partition(nzg, world, rank)[1]


def _median_index(vals):
    """index of the (lower) median element: an actual block, not an average of two"""
    order = sorted(range(len(vals)), key=lambda i: vals[i])
    return order[(len(vals) - 1) // 2]


def bench_grid(a, workload, ctx, steps, warmup, cpu=True, copy_ceiling=False):
    """One grid workload: pre-heat, warm-up, R timed K-step blocks; returns the JSON dict."""
    import math
    import torch
    from pfhubbenchmarks_amd import lib as L
    from pfhubbenchmarks_amd.solver import HipSlabEngine, PhaseFieldSolver, SlabSolver
    world, rank, local_rank, rehearsal, dist, lib = (ctx[k] for k in ("world", "rank", "local_rank", "rehearsal",
                                                                     "dist", "lib"))
    w = workload_table(workload, world)
    scheme, model, elim, bytes_per_cell = w["scheme"], w["model"], w["elim"], w["bytes_per_cell"]
    dim, gn, scaling, dt = w["dim"], w["gn"], w["scaling"], w["dt"]
    h = 1.0
    slab = dist is not None
    if slab:
        from pfhubbenchmarks_amd.solver import FFTSlabSolver, HipFFTSlabEngine
        if scheme == "spectral" or model == "bm6":
            eng = HipFFTSlabEngine(gn, h, world, rank, local_rank, scheme=scheme, model=model, eliminate_phi=elim)
            (eng.set_ic_bm6 if model == "bm6" else eng.set_ic_bm1)()
            solver = FFTSlabSolver(eng)
        elif model in ("bm2", "bm3"):
            from pfhubbenchmarks_amd.solver import HipMultiFieldSlabEngine, MultiFieldSlabSolver
            eng = HipMultiFieldSlabEngine(model, gn, h, world, rank, local_rank)
            eng.set_ic()
            solver = MultiFieldSlabSolver(eng)
        else:
            wide = a.halo == "wide" and not a.fused_slab and gn[2] // world >= 4     # the smallest slab: same answer on every rank
            eng = HipSlabEngine(gn, h, world, rank, local_rank, wide=wide)
            eng.set_ic_bm1(0.5, 0.05)
            if a.strips == "side" and a.transport == "rccl":
                eng.use_strip_stream()
            solver = SlabSolver(eng, transport=a.transport, fused=a.fused_slab)
        timer = eng
        local_cells = gn[0] * gn[1] * eng.nz

        def run(k):
            solver.step(dt, k)

        def sync():
            eng.sync()
            torch.cuda.synchronize()
            dist.barrier()
    else:
        n = gn[:dim]
        s = PhaseFieldSolver(dim=dim, n=n, h=h, kernel=a.kernel, device=local_rank, scheme=scheme, model=model,
                             eliminate_phi=elim)
        {"bm6": s.set_ic_bm6, "bm1": s.set_ic_bm1, "bm2": s.set_ic_bm2, "bm3": s.set_ic_bm3}[model]()
        solver = timer = s
        local_cells = gn[0] * gn[1] * gn[2]

        def run(k):
            s.step(dt, k)

        def sync():
            s.sync()
            torch.cuda.synchronize()

    def max_over_ranks(vals):
        if dist is None:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().tolist()

    F0, C0, _ = solver.diagnostics()
    sync()
    # ---- declared pre-heat by wall time (untimed): every rank runs the SAME number of steps (the slab path exchanges
    # ghost planes every step), decided from a probe block timed with the max over ranks
    t_pre = time.perf_counter()
    preheat_steps, nblk = 0, 5
    while a.preheat_s > 0:
        t0 = time.perf_counter()
        run(nblk)
        sync()
        t1 = time.perf_counter()
        preheat_steps += nblk
        spent, per = max_over_ranks([t1 - t_pre, (t1 - t0) / nblk])     # identical on every rank -> identical decisions
        if spent >= a.preheat_s:
            break
        nblk = max(1, min(200, int(math.ceil((a.preheat_s - spent) / max(per, 1e-7)))))
    preheat_ms = (time.perf_counter() - t_pre) * 1e3
    run(warmup)
    sync()
    # per-launch HIP events cost a few us each: fine beside a 0.4 ms kernel, not beside a 2-D step of a few us -> the
    # launch-bound 2-D workloads are timed by the wall clock alone (kernel_ms_per_step is then the wall time per step)
    # ... and the multi-GPU (slab) path is timed by the wall clock too: its two launches per step would need four event
    # records per step, each a barrier packet on the compute queue (10-15 us per step beside the cross-stream waits)
    per_launch_events = dim == 3 and dist is None
    blocks = []          # (wall seconds, kernel ms per launch, launches)
    repeats = a.repeats if a.repeats > 0 else None
    while True:
        timer.timing(per_launch_events)
        t0 = time.perf_counter()
        run(steps)
        sync()
        el = time.perf_counter() - t0
        k_ms, k_launches = timer.timing_read()
        timer.timing(False)
        blocks.append((el, k_ms, k_launches))
        if repeats is None:      # same decision on every rank: from the first block's max-over-ranks time
            el0 = max_over_ranks([el])[0]
            repeats = max(1, min(25, int(math.ceil(0.25 / max(el0, 1e-9)))))
        if len(blocks) >= repeats:
            break
    els = max_over_ranks([b[0] for b in blocks])
    mi = _median_index(els)
    el, (_, k_ms, k_launches) = els[mi], blocks[mi]
    if not per_launch_events or k_launches == 0:      # (the multi-field FD path records no per-launch events)
        k_ms, k_launches = el / steps * 1e3, steps
    F1, C1, _ = solver.diagnostics()
    if getattr(solver, "transport", None) is not None:
        solver.transport.check()       # peer-copy transport: a wait that gave up means wrong ghosts -- fail loudly

    total_cells = gn[0] * gn[1] * gn[2]
    value = total_cells * steps / el
    # who took part: world size of the process group and the device every rank ran on (checkable evidence of an N-rank job)
    dev_name = torch.cuda.get_device_name(local_rank)
    if dist is not None:
        mine = [{"rank": rank, "device": local_rank, "pid": os.getpid()}]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine[0])
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": gathered,
                      "device_name": dev_name}
    else:
        ranks_info = {"world_size": 1, "backend": None, "ranks": [{"rank": 0, "device": local_rank, "pid": os.getpid()}],
                      "device_name": dev_name}
    # dominant kernel: all step launches of this rank (1 per step on one GPU; interior + 2 boundary launches per
    # step in slab mode, summed)
    kernel_s_per_step = k_ms * 1e-3 * k_launches / max(steps, 1)
    achieved_events = bytes_per_cell * local_cells / kernel_s_per_step / 1e9 if kernel_s_per_step > 0 else 0.0
    # the headline roofline figure comes from the WALL clock of the timed block (the same clock as `value`); the HIP-event
    # average of the step launches sits beside it (it excludes the few us between launches, so it reads 1-2 % higher)
    achieved = bytes_per_cell * local_cells * steps / el / 1e9
    block_ms = [e / steps * 1e3 for e in els]
    out = {
        "metric": "cell-updates/sec on PFHub %s (%s)" % (
            {"bm1": "BM1 Cahn-Hilliard", "bm6": "BM6 Cahn-Hilliard + Poisson", "bm2": "BM2 Ostwald ripening (c + 4 order parameters)",
             "bm3": "BM3 dendritic growth (U, phi)"}[model],
            ("explicit FD, fused HIP stencil" + (" + FFT Poisson" if model == "bm6" else "")) if scheme == "fd"
            else "semi-implicit spectral; hand-written LDS-FFT passes on power-of-two grids 128..1024, rocFFT + HIP k-space kernels otherwise"),
        "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "grid": list(gn), "dt": dt, "h": h, "scheme": "fd-explicit" if scheme == "fd" else "spectral-semi-implicit",
                   "kernel": a.kernel, "variant": a.variant, "target_wgs": a.target_wgs, "ic": {"bm1": "PFHub BM1 (pfbase.py:187-189), z-extruded", "bm6": "PFHub BM6 (pfbase.py:217-219), z-extruded",
                                                                                             "bm2": "PFHub BM2 (pfbase.py:268-296), z-extruded",
                                                                                             "bm3": "PFHub BM3 (pfbase.py:298-320), z-extruded"}[model],
                   "parallelism": "slab%d%s%s%s%s" % (world, "-forced" if a.slab else "", "-REHEARSAL-one-device-gloo" if rehearsal else "",
                                                    ("-ipc-fused" if a.fused_slab else "-ipc")
                                                    if (dist is not None and a.transport == "ipc") else "",
                                                    ("-widehalo" if (slab and getattr(timer, "wide", False)) else "") +
                                                    ("-sidestrips" if (slab and getattr(timer, "strip_stream", None) is not None) else ""))},
        # timed-region bookkeeping: what ran before the clock started, and every timed block (the reported one is the median)
        "preheat_ms": preheat_ms, "preheat_steps": preheat_steps, "repeats": len(blocks),
        "block_ms_per_step": block_ms,
        "steady": (max(block_ms) - min(block_ms)) <= 0.03 * block_ms[mi],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": measured_traffic(workload, a.variant) if world == 1 else None,
                     "traffic_source": "profiles/r*/summary_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; "
                                       "bytes per launch)" % workload,
                     "timing": "wall clock of the timed block (ms_per_step); achieved_hip_events = the same bytes over the "
                               "HIP-event average of the step launches",
                     "achieved_hip_events": achieved_events, "frac_hip_events": achieved_events / HBM_PEAK_GBS,
                     "kernel_ms_per_step": kernel_s_per_step * 1e3, "launches_per_step": k_launches / max(steps, 1),
                     "bytes_per_cell_update": bytes_per_cell},
        "ranks": ranks_info,
        "check": {"F_before": F0, "F_after": F1, "C_rel_drift": abs(C1 - C0) / abs(C0)},
    }
    st = getattr(solver, "status", "") or getattr(timer, "status", "")
    if st:
        out["config"]["status"] = st      # pf_status_string: kernels in use; FFT modes: which transform path runs, chunking
    if scheme == "spectral" and not slab:
        # one pf_step call advances `steps` steps; the state is the resident spectrum, the real-space field is written by
        # the last two steps of the call (PFHIP_SPECTRAL_STORE_EVERY_STEP=1 writes it every step)
        out["config"]["field_store"] = ("every step" if os.environ.get("PFHIP_SPECTRAL_STORE_EVERY_STEP", "") == "1"
                                        else "last two steps of each pf_step call")
    if copy_ceiling and rank == 0 and world == 1 and dim == 3 and scheme == "fd":
        # the same 8 B read + 8 B write per cell as a plain device copy (pfk_stream_copy): what the memory system
        # delivers for this traffic pattern, measured in the same process and the same (pre-heated, busy) state:
        # launched straight behind a few more steps, no idle gap (SURVEY 8d "confirm with a device memcpy")
        import ctypes as C
        gap = local_cells + 8192            # dst starts 64 KB (mod 512 KB) after src ends: pfhip.h, pf_ext_buffer_offset
        blk = torch.ones(gap + local_cells, dtype=torch.float64, device="cuda")
        src, dst = blk[:local_cells], blk[gap:]
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(20)
        sync()
        for it in range(23):
            if it == 3:
                e0.record(st)
            L.check(lib.pfk_stream_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), local_cells,
                                        C.c_void_p(st.cuda_stream)))
        e1.record(st)
        torch.cuda.synchronize()
        copy_gbs = 16.0 * local_cells * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        assert bool((dst[:: max(1, local_cells // 1000)] == 1.0).all())
        del src, dst, blk
        out["roofline"]["device_copy"] = {"achieved": copy_gbs, "unit": "GB/s", "frac_of_peak": copy_gbs / HBM_PEAK_GBS,
                                          "kernel": "pfk_stream_copy (one 16-byte element per thread), %d doubles, 20 launches"
                                                    % local_cells}
        out["roofline"]["frac_of_device_copy"] = achieved / copy_gbs
    # CPU leg (the oracle on the host cores) is DEFERRED: the caller runs out["_cpu"] after every GPU measurement
    if model in ("bm6", "bm2", "bm3") or not cpu:
        pass          # no CPU leg for the BM6 box (the BM6 oracle is a test checker, minutes per step at this size)
    elif rank == 0 and world == 1 and scheme == "spectral":
        out["_cpu"] = lambda: out.__setitem__("cpu_baseline", cpu_baseline_spectral(gn[:dim], dt, 300 if dim == 2 else 8))
    elif rank == 0 and world == 1:
        if dim == 3:
            out["_cpu"] = lambda: out.__setitem__("cpu_baseline", cpu_baseline(gn[0], gn[1], 64, dt, 400))
        else:
            out["_cpu"] = lambda: out.__setitem__("cpu_baseline", cpu_baseline(gn[0], gn[1], 1, dt, 4000))
    if not slab:
        solver.close()
    torch.cuda.empty_cache()
    return out


def side_measurements(a, ctx):
    """Same run, same process, rank 0 of an N = 1 default run: the other configurations BASELINE.json's metric and
    north_star name -- the two 512^2 workloads (wall clock incl. launches), the 1024^3 stencil (north_star's roofline
    target; own pre-heat, timed blocks, per-launch events), the 512^3 spectral step, BM6 at 512^3 in both forms, BM2 / BM3
    on the stencil design, and the reference's own algorithm on the GPU (BM1 = config 1, BM2, BM3) with its CPU restatement
    timed beside it (deferred CPU legs: main() runs them after every GPU leg).
    Order of the returned dict = order in the JSON line: the BASELINE-config lines come LAST, so that a reader who keeps
    only the tail of the line keeps them."""
    from pfhubbenchmarks_amd.solver import PhaseFieldSolver
    res = {}
    for name, sch, nst, dts in (("bm1_spectral_512s", "spectral", 300, 1e-2), ("bm1_fd_512s", "fd", 4001, 1e-3)):
        with PhaseFieldSolver(dim=2, n=512, h=1.0, scheme=sch, device=ctx["local_rank"]) as s2:
            s2.set_ic_bm1(0.5, 0.05)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.1:      # pre-heat, as for the main workload
                s2.step(dts, 101)
                s2.sync()
            reps = []
            for _ in range(5):
                t0 = time.perf_counter()
                s2.step(dts, nst)
                s2.sync()
                reps.append(time.perf_counter() - t0)
            e2 = sorted(reps)[2]
            F2, C2, _ = s2.diagnostics()
        bpc = 72.0 if sch == "spectral" else BYTES_PER_CELL_UPDATE
        res[name] = {"value": 512 * 512 * nst / e2, "unit": "cell-updates/s", "us_per_step": e2 / nst * 1e6,
                     "steps": nst, "repeats": 5, "steady": (max(reps) - min(reps)) <= 0.03 * e2,
                     "config": {"workload": name, "grid": [512, 512, 1],
                                "note": "launch-latency bound 2-D step (a 2 MiB problem): wall clock incl. launches"},
                     "roofline": {"bound": "hbm", "achieved": bpc * 512 * 512 * nst / e2 / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": bpc * 512 * 512 * nst / e2 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                  "bytes_per_cell_update": bpc},
                     "check": {"F_after": F2, "C": C2}}
    notes = {"bm1_spectral_512c": "semi-implicit spectral scheme on the 512^3 box: four hand-written LDS-FFT passes per step; "
                                  "roofline at the 72 B/cell-update idealisation",
             "bm1_fd_1024c": "BASELINE.json config 4 on ONE GPU (16 GiB of state); north_star roofline target",
             "bm6_fd_512c": "BASELINE.json config 5 on ONE GPU, periodic box: FFT Poisson solve (hand-written passes) + coupled "
                            "fused FD step per step; roofline at 72 B/cell-update (CH 16 + phi 8 + Poisson 48)",
             "bm6_fd_512c_elim": "the same physics with phi eliminated (lap_h(k phi) = -(k^2/eps)(c - mean c)): the fused "
                                 "kernel alone, 16 B/cell-update",
             "bm2_fd_512c": "SURVEY 8f next-4 on the stencil design: BM2 (c + 4 order parameters) explicit FD, one-pass LDS-tiled "
                            "kernel; roofline at 80 B/cell-update (5 fields read once, written once)",
             "bm3_fd_512c": "BM3 (U, phi) explicit FD, streaming LDS-tiled kernel; roofline at 32 B/cell-update"}
    for name in ("bm1_spectral_512c", "bm1_fd_1024c", "bm6_fd_512c", "bm6_fd_512c_elim", "bm2_fd_512c", "bm3_fd_512c"):
        b = bench_grid(a, name, ctx, max(10, min(a.steps, 50 if name.startswith("bm1") else 40)), min(a.warmup, 10), cpu=False)
        b["config"]["note"] = notes[name]
        for k in ("metric", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "ranks"):
            b.pop(k, None)
        res[name] = b
    for model in ("bm2", "bm3"):
        res["%s_fem_be" % model] = bench_fem_multi(a, model, steps=6, warmup=2, cpu=not a.no_cpu_baseline)
    fb = bench_fem_be(a, 1, steps=8, warmup=2, ncpu_max=3)
    for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "roofline"):
        fb.pop(k, None)
    res["bm1_fem_be"] = fb
    order = ("bm2_fem_be", "bm3_fem_be", "bm1_fem_be", "bm2_fd_512c", "bm3_fd_512c", "bm6_fd_512c_elim", "bm1_fd_512s",
             "bm1_spectral_512s", "bm6_fd_512c", "bm1_fd_1024c", "bm1_spectral_512c")
    return {k: res[k] for k in order}


if __name__ == "__main__":
    main()
