/* pfhip.h -- C ABI of libpfhip: MI355X (gfx950) phase-field hot path for PFHub BM1 / BM6.
 *
 * This is the drop-in boundary for the time-stepping path of the reference's dolfin/bench1.py and
 * dolfin/bench6.py (paths below are relative to the reference tree).  The reference has no FFI of its own:
 * its "interface" is the handful of FEniCS calls the driver scripts make per step.  Each entry point here
 * names the reference call it replaces.
 *
 *   reference                                               | libpfhip
 *   --------------------------------------------------------+-----------------------------------------------
 *   mesh/space/params block   dolfin/bench1.py:21-69        | pf_create(const pf_config*, pf_handle**)
 *   w.interpolate(w_ic)       dolfin/bench1.py:134-135      | pf_set_ic_bm1 / pf_set_ic_bm6 / pf_set_field
 *     InitialConditionsBench1 dolfin/pfbase.py:177-193      |
 *     InitialConditionsBench6 dolfin/pfbase.py:322-339      |
 *   w0.assign(w); solver.solve() -> (niters, converged)     | pf_step(h, dt, nsteps, &info)   (info.ok ~ converged)
 *                             dolfin/bench1.py:158-162      |
 *   w.assign(w0) on failure   dolfin/bench1.py:171-173      | pf_rollback(h)
 *   total_solute / total_free_energy (df.assemble)          | pf_diagnostics(h, out[3])
 *                             dolfin/bench1.py:121-125,     |
 *                             dolfin/bench6.py:155-165      |
 *   w.split() / outfile.write dolfin/bench1.py:188-191      | pf_get_field(h, field, host, n)
 *   implicit PETSc ghost scatter inside solver.solve()      | pf_step_begin / pf_step_finish + pf_halo_layout:
 *     (dolfinx/pfbase/pde_problems.py:69,87 shows it)       |   the caller exchanges ghost planes in between
 *   implicit MPI_Allreduce inside df.assemble               | pf_diagnostics_local (caller all-reduces 3 doubles)
 *   distributed solves under mpirun (PETSc scatters)        | pf_dist_begin / pf_dist_advance: request protocol for the modes
 *                                                           |   that need an all-to-all (spectral, BM6 Poisson) -- the
 *                                                           |   library describes the exchange, the caller performs it
 *   the whole FEniCS pipeline, unchanged numerics           | cfg.scheme = PF_SCHEME_FEM_BE: P1 'crossed' mesh (bench1.py:21-23,
 *     (mesh, P1xP1, quadrature 3, BE, Newton 1e-6)          |   39-41), Strang-Fix quadrature (:14-16), BE Newton (:85-88);
 *                                                           |   pf_step == solver.solve(), info.iters == niters
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative pf_status; no C++
 * exception crosses the boundary; pf_last_error() gives a human-readable message.  A handle is not
 * thread-safe (one caller thread per handle, like one Python thread per MPI rank in the reference).
 * All field data is IEEE fp64, x fastest: index = (z*ny + y)*nx + x.
 */
#ifndef PFHIP_H
#define PFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFHIP_VERSION 100 /* 0.1.0 */

typedef enum pf_status {
  PF_OK = 0,
  PF_ERR_INVALID = -1,     /* bad argument / config (programmer error) */
  PF_ERR_UNSUPPORTED = -2, /* valid request this build cannot do */
  PF_ERR_HIP = -3,         /* HIP runtime / rocFFT / rocSOLVER error (message in pf_last_error) */
  PF_ERR_STATE = -4,       /* call sequence error (e.g. rollback with nothing to roll back) */
  PF_ERR_NOMEM = -5
} pf_status;

enum { PF_BC_PERIODIC = 0, PF_BC_MIRROR = 1 };           /* mirror = natural no-flux BC (bench1.py:69) */
enum {
  PF_SCHEME_FD_EXPLICIT = 0, /* explicit finite differences, fused stencil kernel (the throughput path) */
  PF_SCHEME_SPECTRAL_SI = 1, /* semi-implicit Fourier spectral (hand-written LDS FFTs on power-of-two axes 128..1024, else rocFFT);
                                BM6: periodic box on one GPU, phi eliminated in Fourier space (implicit term) */
  PF_SCHEME_FEM_BE = 2       /* BE-parity mode: the reference's own P1 'crossed'-mesh backward-Euler Newton solve
                                (bench1.py:21-110) -- 2-D, PF_BC_MIRROR, n[0] == n[1] = corner nodes per side;
                                fields are in the reference's node order: (n*n corners, then (n-1)*(n-1) centres) */
};
enum {
  PF_MODEL_BM1 = 1,
  PF_MODEL_BM6 = 6,
  PF_MODEL_BM2 = 2, /* Ostwald ripening: Cahn-Hilliard + 4 Allen-Cahn fields (dolfin/bench2.py); PF_SCHEME_FEM_BE or _FD_EXPLICIT */
  PF_MODEL_BM3 = 3  /* dendritic growth: heat diffusion + Allen-Cahn (dolfin/bench3.py);          PF_SCHEME_FEM_BE or _FD_EXPLICIT */
};
enum {
  PF_FIELD_C = 0,
  PF_FIELD_MU = 1,
  PF_FIELD_PHI = 2,  /* BM6: electrostatic potential; BM3: the order parameter phi */
  PF_FIELD_ETA1 = 3, /* BM2: eta1 .. eta4 = PF_FIELD_ETA1 .. PF_FIELD_ETA1 + 3 */
  PF_FIELD_U = 7     /* BM3: dimensionless temperature U */
};
enum { PF_KERNEL_AUTO = 0, PF_KERNEL_FUSED = 1, PF_KERNEL_TWOPASS = 2 }; /* FD step implementation */

/* Model + discretisation.  Defaults of the reference: dolfin/bench1.py:32-36, dolfin/bench6.py:38-39. */
typedef struct pf_config {
  int32_t struct_bytes; /* = sizeof(pf_config); guards against ABI drift */
  int32_t dim;          /* 2 or 3 */
  int32_t n[3];         /* PERIODIC: lattice points per axis (n[2] = 1 in 2-D).
                           MIRROR:   nodes of the physical no-flux domain per axis, i.e. intervals+1; the library
                                     runs on the even extension (2*(n-1) periodic points per axis). */
  int32_t bc;           /* PF_BC_* */
  int32_t scheme;       /* PF_SCHEME_* */
  int32_t model;        /* PF_MODEL_* */
  int32_t kernel;       /* PF_KERNEL_* (FD scheme only) */
  int32_t device;       /* HIP device ordinal */
  int32_t nranks;       /* slab decomposition along z (3-D only; 2-D problems run as replicas); 1 = whole domain.
                           PERIODIC: a ring of slabs.  MIRROR (FD scheme, BM1): a LINE of slabs over the n[2] physical
                           planes -- x and y stay even-extended, z is not: the first / last rank own the walls and
                           mirror their own planes into the outer ghost layers (>= 3 planes per rank).
                           MIRROR with the slab-FFT modes (PF_SCHEME_SPECTRAL_SI with BM1; PF_SCHEME_FD_EXPLICIT with BM6 = the
                           reference's Dirichlet-x / no-flux Poisson problem, dolfin/bench6.py:77-90): the box stays on its even
                           extension along z too, the slabs form a RING over the 2 (n[2] - 1) lattice planes and the periodic
                           slab FFT transforms the extension (BM6: of the odd-in-x right-hand side). */
  int32_t rank;
  int32_t force_slab;   /* 1: use the ghost-plane (slab) code path even with nranks == 1 -- the rank is then its own ring
                           neighbour; lets a single GPU exercise the exact multi-GPU path (tests) */
  double h;             /* grid spacing (same on every axis) */
  double rho_s, c_alpha, c_beta, kappa, M; /* bench1.py:32-36 */
  double k, eps_r;                         /* bench6.py:38-39 (BM6 only) */
  void* stream;         /* hipStream_t to launch on; NULL = library creates its own */
  double* ext_c[2];     /* optional caller-owned DEVICE buffers for the two c time levels, each
                           pf_field_elems_with_ghosts() doubles; NULL = library allocates (hipMalloc) */
  double* ext_a2a[2];   /* slab FFT modes (nranks > 1 with PF_SCHEME_SPECTRAL_SI or PF_MODEL_BM6): optional caller-owned
                           all-to-all buffers, pf_a2a_buffer_doubles() doubles each */
  double* ext_phi;      /* BM6 slab mode: optional caller-owned ghosted phi buffer (same size as an ext_c buffer) */
  int32_t flags;        /* PF_FLAG_* */
  int32_t max_newton;   /* PF_SCHEME_FEM_BE: Newton iteration cap per solve; 0 = 10, the reference's
                           nlparams['maximum_iterations'] (dolfin/bench1.py:88).  A solve that needs more reports
                           info.ok = 0 with the state untouched and the caller halves dt (bench1.py:164-177).
                           Following the COMMITTED run's time grid with exact linear solves needs up to 24 (rows
                           21, 37 of results/bench1_out.csv): the fixture driver passes 100. */
  double model_params[8]; /* PF_MODEL_BM2: {kappa_eta, w, alpha, L} (with c_alpha, c_beta, rho_s = rho, kappa = kappa_c, M
                             above; dolfin/bench2.py:33-41).  PF_MODEL_BM3: {W0, tau0, D, Delta} (bench3.py:31-37).
                             pf_config_model_defaults fills the reference's values. */
} pf_config;

/* pf_config.flags */
enum {
  PF_FLAG_WIDE_HALO = 2,        /* slab mode, FD scheme (BM1, or BM6 with phi eliminated), fused kernel: 4 ghost planes per side
                                   exchanged every SECOND step instead of 2 every step (communication-avoiding).  Steps
                                   alternate: A (needs fresh ghosts: pf_halo_layout.needs_exchange = 1) computes planes
                                   [-2, nz+2) -- begin = the interior [2, nz-2), finish = two 4-plane strips; B computes
                                   [0, nz) from them in ONE launch with no exchange (begin does it, finish only swaps).
                                   Halves the hand-offs per step and cuts the strips' redundant reads from 12 to 8 planes
                                   per step for 4 / nz more arithmetic.  Needs >= 4 (periodic) / 5 (mirror walls) planes per
                                   rank.  Results are bit-identical to the 2-ghost path. */
  PF_FLAG_FEM_ALWAYS_PIVOT = 4, /* PF_SCHEME_FEM_BE: row exchanges in EVERY dense factorisation of the Newton solve.  By default
                                   the dense reduction levels are factored without them (own cooperative LU kernel), and a Newton
                                   solve that then fails is repeated once with them.  Same converged states either way; set this
                                   when the step's ITERATION COUNT steers the run -- the reference's dt rule, bench1.py:180-183
                                   -- so that the time grid never depends on that optimisation (drivers: controller="reference") */
  PF_FLAG_BM6_ELIMINATE_PHI = 1 /* BM6, periodic box, FD scheme: phi solves lap_h(phi) = -(k/eps)(c - mean c) with the SAME
                                   discrete Laplacian the Cahn-Hilliard step applies to mu, so lap_h(k phi) =
                                   -(k^2/eps)(c - mean c) exactly and the time step needs no Poisson solve:
                                   c_new += -dt M k^2/eps (c - mean c).  phi is still solved for diagnostics / output.
                                   Differs from the explicit-phi path by rounding only. */
};

typedef struct pf_step_info {
  int32_t ok;        /* 1 if the state after the step(s) is finite and inside the guard band c in [-1, 2] */
  int32_t nsteps;    /* steps actually taken */
  double cmin, cmax; /* global extrema after the last step (local extrema in slab mode) */
  int32_t iters;     /* PF_SCHEME_FEM_BE: Newton iterations of the last step (the reference's `niters`) */
  int32_t reserved0;
} pf_step_info;

/* Ghost-plane layout of one rank's c buffer in slab mode (element offsets into the CURRENT c buffer). */
typedef struct pf_halo_layout {
  double* base;            /* device pointer of the current c buffer */
  int64_t plane_elems;     /* doubles per plane of the slab axis */
  int32_t ghost;           /* ghost planes per side (2; 4 with PF_FLAG_WIDE_HALO) */
  int32_t n_local;         /* owned planes */
  int64_t send_lo_off, send_hi_off; /* first / last `ghost` owned planes */
  int64_t recv_lo_off, recv_hi_off; /* ghost planes below / above */
  int32_t rank_lo, rank_hi;         /* neighbour ranks: a ring (periodic bc) or a line (mirror bc: -1 = wall, the
                                       library mirrors the owned planes into those ghost layers itself) */
  int32_t cur_index;                /* which of the two c buffers (cfg.ext_c[cur_index]) is current */
  int32_t needs_exchange;           /* 1: the next pf_step_finish reads the ghost planes of the current buffer -- exchange
                                       them between pf_step_begin and pf_step_finish; 0 (every second step with
                                       PF_FLAG_WIDE_HALO): no exchange for the next step */
} pf_halo_layout;

/* Distributed operations that need collectives the library does not perform itself (it never communicates).
 * Protocol: pf_dist_begin(h, op, dt); then loop { pf_dist_advance(h, &req); switch (req.kind) ... } until
 * PF_DIST_DONE.  The library runs its kernels up to the next exchange on the handle's stream and describes the
 * exchange; the caller performs it stream-ordered on the same stream (RCCL through torch.distributed) and calls
 * pf_dist_advance again. */
enum { PF_DIST_DONE = 0, PF_DIST_ALLTOALL = 1, PF_DIST_HALO = 2 };
enum {
  PF_DIST_OP_STEP = 1,   /* one time step (spectral slab step; BM6 slab step = Poisson solve + coupled FD step) */
  PF_DIST_OP_REFRESH = 2 /* make everything pf_diagnostics_local reads consistent with the current c:
                            ghost planes, phi (BM6), resident spectrum (spectral) */
};
typedef struct pf_dist_request {
  int32_t kind;             /* PF_DIST_* */
  int32_t n_halo;           /* PF_DIST_HALO: how many ghosted buffers to refresh (layout: pf_halo_layout_get) */
  double* src;              /* PF_DIST_ALLTOALL: equal split, doubles_per_peer doubles to / from every rank */
  double* dst;
  int64_t doubles_per_peer;
  double* halo_base[2];     /* PF_DIST_HALO: base pointers of the ghosted buffers */
} pf_dist_request;

typedef struct pf_handle pf_handle;

/* ---- library ---------------------------------------------------------------------------------------- */
int pf_version(void);
/* message of the last failure on this handle (NULL handle: last failure of a pf_create on this thread) */
const char* pf_last_error(const pf_handle* h);
/* one-line description of the compute path this handle runs (scheme, kernel family) -- and a WARNING when it is not the
 * fast one: an odd nx makes the FD scheme fall back from the fused 16 B/cell kernel to the two-pass kernels (40 B/cell,
 * ~6x slower).  The string lives as long as the handle. */
const char* pf_status_string(const pf_handle* h);
/* number of HIP devices visible, or <0; touches the HIP runtime */
int pf_device_count(void);
/* fill *cfg with the reference's BM1 constants for an n^dim periodic grid (pure host, no HIP call) */
int pf_config_default(pf_config* cfg, int dim, int n, double h);
/* switch an initialised config to `model` with the reference's constants for it (bench1.py:32-36, bench2.py:33-41,
 * bench3.py:31-37, bench6.py:38-39); pure host */
int pf_config_model_defaults(pf_config* cfg, int model);

/* ---- pure-host helpers (no HIP call; usable without a GPU) ------------------------------------------- */
/* planes [*first, *first + *count) of `n_planes` owned by `rank` of `nranks` (remainder to the low ranks) */
int pf_slab_partition(int n_planes, int nranks, int rank, int* first, int* count);
/* doubles a caller must provide per ext_c buffer for this config (owned + ghost planes) */
int64_t pf_field_elems_with_ghosts(const pf_config* cfg);
/* doubles of one rank's owned part of a field (what pf_set_field / pf_get_field move) */
int64_t pf_field_elems(const pf_config* cfg);

/* Recommended distance, in doubles, from ext_c[0] to ext_c[1] (which = 1) or to ext_phi (which = 2) when the caller
 * carves them out of ONE allocation: on MI355X the read and the write stream of the fused step collide in the HBM
 * channel map when the buffers sit 200-330 KB (mod 512 KB) apart -- 0.41 instead of 0.366 ms per 512^3 step
 * (profiles/r01/buffer_offset.log); separately allocated buffers land anywhere.  Library-owned buffers follow the
 * same rule.  >= pf_field_elems_with_ghosts(); <0: invalid config. */
int64_t pf_ext_buffer_offset(const pf_config* cfg, int which);

/* doubles per all-to-all buffer of the slab FFT modes for this config (<0: box not divisible by nranks) */
int64_t pf_a2a_buffer_doubles(const pf_config* cfg);

/* ---- life cycle ------------------------------------------------------------------------------------- */
int pf_create(const pf_config* cfg, pf_handle** out);
int pf_destroy(pf_handle* h);

/* ---- state ------------------------------------------------------------------------------------------ */
int pf_set_ic_bm1(pf_handle* h, double c0, double eps);  /* pfbase.py:187-189; z-extruded in 3-D (b13d.py:55) */
int pf_set_ic_bm6(pf_handle* h, double c0, double c1);   /* pfbase.py:332-334 */
/* InitialConditionsBench2 (pfbase.py:268-296; bench2.py:58-62: c0 0.5, eps 0.05, eps_eta 0.1, psi 1.5): c, mu = 0, eta1..4 */
int pf_set_ic_bm2(pf_handle* h, double c0, double eps, double eps_eta, double psi);
/* InitialConditionsBench3 (pfbase.py:298-320; bench3.py:52-57: r 8, w 1, vin 1, vout -1): U = Delta, phi = radial seed */
int pf_set_ic_bm3(pf_handle* h, double r, double w, double vin, double vout);
int pf_set_field(pf_handle* h, int field, const double* host, size_t n); /* caller-owned host buffer, copied */
int pf_get_field(pf_handle* h, int field, double* host, size_t n);

/* ---- time stepping ---------------------------------------------------------------------------------- */
/* nranks == 1: advance nsteps steps of size dt.  Asynchronous on the handle's stream unless info != NULL
 * (then the guard reduction is read back).  The state before the LAST step stays available to pf_rollback.
 * Intermediate states of a multi-step call are not observable and need not be materialised in HBM: the 2-D FD path
 * fuses up to 4 steps per launch, the spectral scheme writes the real-space field in the last two steps only. */
int pf_step(pf_handle* h, double dt, int nsteps, pf_step_info* info);
int pf_rollback(pf_handle* h);
/* PF_FLAG_BM6_ELIMINATE_PHI: the lattice mean of c (a conserved quantity).  nranks == 1: computed by the library when
 * needed; slab mode: the caller passes the GLOBAL mean (total_solute / volume) once after setting the state. */
int pf_set_mean_c(pf_handle* h, double mean_c);
int pf_sync(pf_handle* h);

/* slab mode (nranks > 1), one step:  begin (interior planes; needs no ghosts) -> caller exchanges the ghost
 * planes described by pf_halo_layout on its own stream and makes the handle's stream wait for it ->
 * finish (boundary planes + buffer swap). */
int pf_halo_layout_get(pf_handle* h, pf_halo_layout* out);
/* BM2 / BM3 explicit FD in slab mode (PF_SCHEME_FD_EXPLICIT, nranks > 1 or force_slab, dim 3; PF_BC_MIRROR: the ring runs over
 * the lattice planes of the even extension, pf_set_field takes the whole physical box, pf_get_field returns the rank's lattice
 * planes in physical x, y nodes): the ghost planes
 * of field `field` (0 .. nf-1: BM2 c, eta1..4; BM3 U, phi -- dolfin/bench2.py:76-113, bench3.py:63-97 are the equations) in
 * the CURRENT time level, ghost = 2 (BM2: c reaches through mu) or 1 (BM3) per side, the slabs a ring.  Protocol per step:
 * refresh the ghost planes of EVERY field from the two ring neighbours, then pf_step(h, dt, 1, info) -- or, overlapped like
 * the single-field path: post the exchange, pf_step_begin (the planes that need no ghosts), wait for the exchange,
 * pf_step_finish (the g owned planes next to each ghost layer; swaps the time levels).  The layout moves to
 * the other time level with each step (cur_index).  With pf_config.ext_c the two time levels live in caller-owned buffers
 * of pf_field_elems_with_ghosts() doubles each (all fields, field-major), so a communication library can send / receive
 * the planes in place.  Results are bit-identical to the single-GPU box. */
int pf_field_halo_layout(pf_handle* h, int field, pf_halo_layout* out);
int pf_step_begin(pf_handle* h, double dt);
int pf_step_finish(pf_handle* h);
/* Launch the boundary strips of pf_step_finish on `stream` instead of the handle's stream (NULL = back to the handle's
 * stream).  The strips only depend on the ghost planes and on the owned boundary planes of the CURRENT buffer, so on a
 * stream of their own they run beside the interior launch as soon as the exchange has delivered -- the exchange wait
 * leaves the critical path.  The caller orders the streams: the strip stream must wait for everything that last read the
 * output buffer (an event recorded on the handle's stream before pf_step_begin) and for the exchange; the handle's stream
 * must wait for the strips before the next exchange is posted / the next step begins.  SlabSolver does exactly that. */
int pf_set_strip_stream(pf_handle* h, void* stream);

/* slab mode, one step in ONE launch (peer-copy transport): the interior chunks are dispatched first; the workgroups
 * of the two boundary strips are dispatched last and poll flag_lo / flag_hi (device words the neighbours publish `seq`
 * in once their planes have landed: pfk_push_planes) before they read the ghost planes -- no second launch, no
 * cross-stream event on the critical path.  Null flag = no neighbour on that side (ignored at a mirror-bc wall
 * anyway).  The poll is bounded (~1 s); *timeout is set to 1 if it gives up.  Needs > 2 * ghost planes per rank. */
int pf_step_slab_fused(pf_handle* h, double dt, const int64_t* flag_lo, const int64_t* flag_hi, int64_t seq,
                       int32_t* timeout);

/* slab FFT modes and ghost refresh: see pf_dist_request above */
int pf_dist_begin(pf_handle* h, int op, double dt);
int pf_dist_advance(pf_handle* h, pf_dist_request* req);

/* ---- diagnostics (bench1.py:121-125, bench6.py:155-165) ---------------------------------------------- */
/* out = {total_free_energy, total_solute, f_elec part}; synchronises.  nranks > 1: use the _local variant
 * and sum the 3 doubles over ranks. */
int pf_diagnostics(pf_handle* h, double out[3]);
int pf_diagnostics_local(pf_handle* h, double out[3]);

/* Counters of the last pf_step, for tests and logs (solver.solve() reports only (niters, converged), bench1.py:162).
 * key PF_STAT_FEM_ATTEMPTS: Newton solves the step took (1; 2 = an optimistic solve without row exchanges failed and was
 * repeated with them); PF_STAT_FEM_NPVT_LEVELS: reduction levels its last attempt factored without row exchanges. */
enum { PF_STAT_FEM_ATTEMPTS = 0, PF_STAT_FEM_NPVT_LEVELS = 1 };
int pf_get_stat(pf_handle* h, int key, int64_t* value);

/* ---- measurement hooks ------------------------------------------------------------------------------- */
/* average device time (ms) of the dominant step kernel over the launches since the last call, measured with
 * HIP events on the handle's stream; enable with pf_timing_enable(h, 1).  Launches that pf_set_strip_stream moved to
 * another stream (the boundary strips of pf_step_finish) are NOT in this figure -- it then covers the interior launches
 * only; time such a configuration with the wall clock (bench.py times every slab-mode run that way). */
int pf_timing_enable(pf_handle* h, int on);
int pf_timing_read(pf_handle* h, double* avg_ms, int64_t* launches);
/* the individual launches since pf_timing_enable(h, 1) (at most 65536 are kept): device time of each (dur_ms) and
 * its start relative to the enable call (start_ms, may be NULL).  *n = samples available; min(*n, cap) are written.
 * Does not reset anything.  tools/ramp_probe.py and bench.py's steady-state check read the launch timeline here. */
int pf_timing_samples(pf_handle* h, double* dur_ms, double* start_ms, int64_t cap, int64_t* n);

/* ---- kernel-level entry points (stateless; caller-owned device pointers) ------------------------------ */
typedef struct pfk_ch_params {
  double c_alpha, c_beta, two_rho; /* f'(c) = two_rho (c-ca)(cb-c)((cb-c)-(c-ca)) */
  double kappa_over_h2;            /* kappa / h^2 */
  double dtM_over_h2;              /* dt * M / h^2 */
  double k_phi;                    /* BM6 coupling k (0 for BM1) */
  double gq, cbar;                 /* PF_FLAG_BM6_ELIMINATE_PHI: c_new += gq (c - cbar), gq = -dt M k^2/eps (0 = off) */
} pfk_ch_params;

/* One explicit FD Cahn-Hilliard step on planes [zlo, zhi) of a slab:
 *   mu = f'(c) - kappa lap_h c (+ k phi);  c_out = c_in + dt M lap_h mu      (7-point lap_h, 5-point when nz == 1)
 * c_in / c_out: (nz + 2*ghost) planes of ny*nx doubles, owned planes start at plane `ghost`.
 * zwrap != 0: the slab axis is periodic inside this buffer (ghost planes unused); else planes -2..nz+1 are read
 * from the ghost planes.  phi may be NULL (BM1).  impl: PF_KERNEL_*. */
int pfk_ch_fd_step(const double* c_in, double* c_out, const double* phi, int nx, int ny, int nz, int ghost,
                   int zwrap, int zlo, int zhi, const pfk_ch_params* p, int impl, void* stream);

/* tuning hooks for benchmarks: key 0 = fused-kernel variant (table in csrc/ch_fd_kernels.hip);
 * key 1 = target number of workgroups for the z-chunk split; key 2 = minimum planes per z-chunk;
 * key 3 = largest number of 2-D time steps fused into one launch (1, 2 or 4); key 4 = 10 K + rows per wave of the 2-D
 * K-step kernel; keys 5 / 6 = workgroups per CU / kernel form of pfk_stream_copy; key 7 = workgroups per pfk_push_planes
 * message; key 8 = diagnostics kernel form (10 x rows per wave + planes in flight; 0 = the round-1 kernel); key 9 = its
 * target workgroup count; key 10 = 0 / 1: plain / non-temporal (default) stores of the output planes of the BM2 / BM3 streaming kernels;
 * key 11 = TEST HOOK: every new device allocation of the library is filled with this byte (0 .. 255; -1 = off) -- results must not
 * depend on it */
int pfk_set_tuning(int key, int value);

/* Device memory with the placement policy the library uses for its own arrays (csrc/device_alloc.hip; no counterpart in the
 * reference, which leaves allocation to PETSc / CuArrays).  On MI355X the bandwidth of a kernel depends on where its arrays
 * lie physically, and hipMalloc does not give the same physical layout twice; arrays of 32 MiB and more are therefore
 * physically contiguous allocations (hipExtMallocWithFlags, hipDeviceMallocContiguous), which gives every handle of every
 * process the same step time.  A caller that hands its own buffers to pf_create (pf_config.ext_c / ext_phi) gets the same
 * behaviour by allocating them here.  PFHIP_ALLOC=plain|scatter[:KiB] select the other measured policies (A/B only).
 * pf_device_free takes pointers of pf_device_malloc only; it synchronises the device like hipFree. */
int pf_device_malloc(void** dev_ptr, size_t bytes);
int pf_device_free(void* dev_ptr);

/* Device copy dst[i] = src[i], n doubles, 16-byte accesses: the measured HBM ceiling for a 1-read + 1-write stream
 * (8 B + 8 B per element -- the algorithmic traffic of the fused CH step).  bench.py times it beside the stencil so
 * roofline.frac (against the 8 TB/s datasheet peak, SURVEY 8d: "confirm with a device memcpy/triad on the box") can
 * also be read against what the memory system delivers.  n must be even, pointers 16-byte aligned. */
int pfk_stream_copy(const double* src, double* dst, int64_t n, void* stream);

/* Peer-copy halo transport (an alternative to RCCL send/recv inside one node; pfhubbenchmarks_amd/solver.py:
 * IpcHaloTransport).  RCCL's send/recv kernel is starved beside the HBM-saturating stencil and finishes with it
 * (+27 us per step); here the sender writes the neighbour's ghost planes directly:
 *   pfk_push_planes: dst[0..n) = src[0..n) (dst may be a peer-mapped IPC pointer), then *flag = seq with system-scope
 *                    release, by the last workgroup to finish; `ticket` = zero-initialised uint32 owned by the sender.
 *   pfk_wait_flag:   stream-ordered wait until *flag >= seq (one polling lane; gives up after ~2 s and sets *timeout).
 * Sequence numbers are monotonic per neighbour pair, so no host-side ordering of record / wait calls is needed. */
int pfk_push_planes(const double* src, double* dst, int64_t n, int64_t* flag, int64_t seq, uint32_t* ticket,
                    void* stream);
int pfk_wait_flag(const int64_t* flag, int64_t seq, int32_t* timeout, void* stream);
/* stream-ordered *flag = seq with system-scope release (flag may be a peer-mapped IPC pointer): the consumer's
 * acknowledgement "I have read the ghost planes of exchange seq", which a sender waits for before it overwrites them */
int pfk_signal_flag(int64_t* flag, int64_t seq, void* stream);

/* Flag words of the peer-copy transport.  A flag is polled by a RUNNING kernel on its home GPU while a PEER GPU writes it
 * (pfk_push_planes), so it must not live in ordinary (coarse-grained) device memory, whose lines the home GPU's L2 may
 * keep across the peer's write: pfk_flags_alloc returns FINE-GRAINED device memory (hipExtMallocWithFlags,
 * hipDeviceMallocFinegrained), zero-filled; `timeout_host` (may be NULL) receives one int32 in mapped, pinned HOST memory
 * for the wait kernels' give-up mark, readable by the host at any time without a device sync.
 * pfk_ipc_export / pfk_ipc_import move such an allocation (or any hipMalloc'ed base pointer) between the processes of a
 * node: 64-byte handle = hipIpcMemHandle_t.  pfk_ipc_close unmaps an imported pointer. */
int pfk_flags_alloc(int n_words, int64_t** flags_dev, int32_t** timeout_host);
int pfk_flags_free(int64_t* flags_dev, int32_t* timeout_host);
int pfk_ipc_export(const void* dev_base, unsigned char handle[64]);
int pfk_ipc_import(const unsigned char handle[64], void** dev_ptr);
int pfk_ipc_close(void* dev_ptr);

/* Measures the cost of one grid-wide barrier (agent-scope release + atomic count-in + acquire) of a cooperative launch
 * with nblocks x nthreads: the price a persistent multi-phase kernel pays instead of a kernel boundary. */
int pfk_grid_barrier_probe(int nblocks, int nthreads, int iters, double* us_per_barrier);

/* The same question for a kernel whose workgroups all sit on ONE XCD (one L2: a hand-off needs no L2 write-back, only
 * loads that bypass the per-CU L1): `nblocks` workgroups are launched chip-wide, those that landed on XCD 0 (HW_REG_XCC_ID)
 * stay and run `iters` rounds of { write a 128-byte record, arrive, wait, read the next participant's record (handoff != 0),
 * arrive, wait }.  us_per_round = two barriers + the hand-off; participants = workgroups that took part; stale = records
 * read with an old value (must be 0). */
int pfk_xcd_barrier_probe(int nblocks, int nthreads, int iters, int handoff, double* us_per_round, int* participants,
                          int* stale);

/* Shader clock actually held, measured inside a kernel: delta s_memtime / delta s_memrealtime x 100 MHz over ~spin_us
 * microseconds, by `nblocks` workgroups.  out_dev: 2 * nblocks DEVICE doubles {MHz, start of the probe in ms of the
 * free-running 100 MHz counter}.  busy = 0: one sleeping lane per workgroup; busy = 1: all SIMDs run an fp64 fma chain
 * (no memory traffic) -- an "active but not HBM-bound" load for tools/ramp_probe.py.  Asynchronous on `stream`. */
int pfk_clock_probe(double* out_dev, int nblocks, int spin_us, int busy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PFHIP_H */
