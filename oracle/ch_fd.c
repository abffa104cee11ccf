/* ORACLE (test infrastructure, not product code) -- plain-C CPU restatement of the explicit finite-difference
 * Cahn-Hilliard step and its diagnostics, i.e. of the algorithm the HIP kernels in
 * pfhubbenchmarks_amd/csrc/ implement.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 *
 * Physics restated from the reference (paths relative to the reference tree):
 *   dolfin/pfbase.py:361-383   d/dt c = div(M grad mu),  mu = f'(c) - kappa lap c   (cahn_hilliard_weak_form)
 *   dolfin/bench1.py:63-65     f_chem = rho_s (c - c_alpha)^2 (c_beta - c)^2, dfdc = d f_chem / dc
 *   dolfin/bench6.py:66-68     f_elec = k c phi / 2, dfdc += k phi
 *   dolfin/bench1.py:121-125   total_solute = int c dx; total_free_energy = int f_chem + kappa/2 |grad c|^2 dx
 *   dolfin/pfbase.py:187-189   BM1 initial condition (bench1.py:48-49 amplitudes); b13d.py:55 z-extrusion
 *
 * The reference discretises with P1 finite elements + backward Euler inside FEniCS/PETSc (see oracle/fem_be.py
 * for that restatement, which is pinned against the reference's committed result files).  This file restates
 * the uniform-grid explicit scheme named by BASELINE.json's north_star; the operation order below is THE
 * definition the HIP kernels are bit-compared against (compile with -ffp-contract=off; fma() is explicit).
 *
 *   Lxy  = ((c[x-1] + c[x+1]) + (c[y-1] + c[y+1])) - 4 c        (last step: fma(-4, c, sum))
 *   Lz   = (c[z-1] + c[z+1]) - 2 c                              (fma(-2, c, sum); exactly 0 when nz == 1)
 *   a = c - ca; b = cb - c; fp = two_rho * ((a*b) * (b - a))
 *   mu   = fma(-kappa/h^2, Lxy + Lz, fp)   [+ fma(k, phi, .) for BM6]
 *   cnew = fma(dt M / h^2, Mxy + Mz, c)    with Mxy, Mz the same stencils applied to mu
 *   [BM6, periodic box, phi eliminated: lap_h(k phi) = -(k^2/eps)(c - mean c) exactly, because phi solves the discrete
 *    Poisson problem with the same lap_h (bench6.py:72); then  cnew = fma(gq, c - cbar, cnew),  gq = -dt M k^2 / eps]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* number of OpenMP threads the step uses (0 = leave the runtime default); returns the count in effect */
int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

typedef struct orc_ch_params {
  double c_alpha, c_beta, two_rho;
  double kappa_over_h2;
  double dtM_over_h2;
  double k_phi;
  double gq, cbar; /* BM6 in a periodic box with phi eliminated: cnew += gq (c - cbar), gq = -dt M k^2 / eps (0 = off) */
} orc_ch_params;

static inline int64_t wrap(int64_t i, int64_t n) {
  i %= n;
  return i < 0 ? i + n : i;
}

/* plane pointer for logical plane z in [-ghost, nz+ghost) (or any z when zwrap) */
static inline const double* plane_of(const double* base, int64_t z, int nz, int ghost, int zwrap, int64_t pe) {
  if (zwrap) z = wrap(z, nz);
  return base + (z + ghost) * pe;
}

static inline double mu_at(const double* pm, const double* p0, const double* pp, const double* phi0, int nx,
                           int ny, int x, int y, const orc_ch_params* q) {
  const int xm = (int)wrap(x - 1, nx), xp = (int)wrap(x + 1, nx);
  const int ym = (int)wrap(y - 1, ny), yp = (int)wrap(y + 1, ny);
  const double c = p0[(int64_t)y * nx + x];
  const double sx = p0[(int64_t)y * nx + xm] + p0[(int64_t)y * nx + xp];
  const double sy = p0[(int64_t)ym * nx + x] + p0[(int64_t)yp * nx + x];
  const double lxy = fma(-4.0, c, sx + sy);
  const double sz = pm[(int64_t)y * nx + x] + pp[(int64_t)y * nx + x];
  const double lz = fma(-2.0, c, sz);
  const double a = c - q->c_alpha, b = q->c_beta - c;
  const double fp = q->two_rho * ((a * b) * (b - a));
  double mu = fma(-q->kappa_over_h2, lxy + lz, fp);
  if (phi0) mu = fma(q->k_phi, phi0[(int64_t)y * nx + x], mu);
  return mu;
}

/* Same contract as pfk_ch_fd_step (include/pfhip.h).  Returns 0, or -1 on bad arguments. */
int orc_ch_fd_step(const double* c_in, double* c_out, const double* phi, int nx, int ny, int nz, int ghost,
                   int zwrap, int zlo, int zhi, const orc_ch_params* q) {
  if (nx < 1 || ny < 1 || nz < 1 || ghost < 0 || zlo < 0 || zhi > nz || zlo > zhi) return -1;
  if (!zwrap && ghost < 2) return -1;
  const int64_t pe = (int64_t)nx * ny;
  /* mu on planes zlo-1 .. zhi (rolling window of 3 planes) */
  double* mu = (double*)malloc(sizeof(double) * pe * 3);
  if (!mu) return -1;
  double* mrow[3] = {mu, mu + pe, mu + 2 * pe};
#define MU_PLANE(dst, z)                                                                        \
  do {                                                                                          \
    const double* pm_ = plane_of(c_in, (int64_t)(z)-1, nz, ghost, zwrap, pe);                   \
    const double* p0_ = plane_of(c_in, (int64_t)(z), nz, ghost, zwrap, pe);                     \
    const double* pp_ = plane_of(c_in, (int64_t)(z) + 1, nz, ghost, zwrap, pe);                 \
    const double* ph_ = phi ? plane_of(phi, (int64_t)(z), nz, ghost, zwrap, pe) : (const double*)0; \
    _Pragma("omp parallel for schedule(static)") for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x)(dst)[(int64_t)y * nx + x] = \
        mu_at(pm_, p0_, pp_, ph_, nx, ny, x, y, q);                                             \
  } while (0)
  if (zlo < zhi) {
    MU_PLANE(mrow[0], zlo - 1);
    MU_PLANE(mrow[1], zlo);
  }
  for (int z = zlo; z < zhi; ++z) {
    MU_PLANE(mrow[2], z + 1);
    const double* mm = mrow[0];
    const double* m0 = mrow[1];
    const double* mp = mrow[2];
    const double* c0 = plane_of(c_in, z, nz, ghost, 0, pe);
    double* o = c_out + ((int64_t)z + ghost) * pe;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ny; ++y) {
      const int ym = (int)wrap(y - 1, ny), yp = (int)wrap(y + 1, ny);
      for (int x = 0; x < nx; ++x) {
        const int xm = (int)wrap(x - 1, nx), xp = (int)wrap(x + 1, nx);
        const double m = m0[(int64_t)y * nx + x];
        const double sx = m0[(int64_t)y * nx + xm] + m0[(int64_t)y * nx + xp];
        const double sy = m0[(int64_t)ym * nx + x] + m0[(int64_t)yp * nx + x];
        const double mxy = fma(-4.0, m, sx + sy);
        const double sz = mm[(int64_t)y * nx + x] + mp[(int64_t)y * nx + x];
        const double mz = fma(-2.0, m, sz);
        double cn = fma(q->dtM_over_h2, mxy + mz, c0[(int64_t)y * nx + x]);
        if (q->gq != 0.0) cn = fma(q->gq, c0[(int64_t)y * nx + x] - q->cbar, cn);
        o[(int64_t)y * nx + x] = cn;
      }
    }
    double* t = mrow[0];
    mrow[0] = mrow[1];
    mrow[1] = mrow[2];
    mrow[2] = t;
  }
#undef MU_PLANE
  free(mu);
  return 0;
}

/* mu = f'(c) - kappa lap_h c (+ k phi) on owned planes [0, nz) -> mu_out (nz planes, no ghosts) */
int orc_ch_mu(const double* c_in, double* mu_out, const double* phi, int nx, int ny, int nz, int ghost, int zwrap,
              const orc_ch_params* q) {
  const int64_t pe = (int64_t)nx * ny;
  for (int z = 0; z < nz; ++z) {
    const double* pm = plane_of(c_in, (int64_t)z - 1, nz, ghost, zwrap, pe);
    const double* p0 = plane_of(c_in, (int64_t)z, nz, ghost, zwrap, pe);
    const double* pp = plane_of(c_in, (int64_t)z + 1, nz, ghost, zwrap, pe);
    const double* ph = phi ? plane_of(phi, (int64_t)z, nz, ghost, zwrap, pe) : 0;
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) mu_out[z * pe + (int64_t)y * nx + x] = mu_at(pm, p0, pp, ph, nx, ny, x, y, q);
  }
  return 0;
}

/* Raw sums over owned planes [0, nz) (Neumaier-compensated):
 *   out[0] = sum c
 *   out[1] = sum rho (c-ca)^2 (cb-c)^2
 *   out[2] = sum (c[x+1]-c)^2 + (c[y+1]-c)^2 + (c[z+1]-c)^2      (forward differences, periodic / ghost in z)
 *   out[3] = sum c * phi                                           (0 if phi == NULL)
 *   out[4] = min c, out[5] = max c
 * The caller scales: C = h^d out[0]; F = h^d (out[1] + kappa/(2 h^2) out[2] + k/2 out[3]). */
static inline void nsum(double* s, double* comp, double v) {
  const double t = *s + v;
  if (fabs(*s) >= fabs(v))
    *comp += (*s - t) + v;
  else
    *comp += (v - t) + *s;
  *s = t;
}

int orc_ch_diag(const double* c, const double* phi, int nx, int ny, int nz, int ghost, int zwrap, double rho,
                double c_alpha, double c_beta, double out[6]) {
  const int64_t pe = (int64_t)nx * ny;
  double s[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0};
  double mn = INFINITY, mx = -INFINITY;
  for (int z = 0; z < nz; ++z) {
    const double* p0 = plane_of(c, z, nz, ghost, zwrap, pe);
    const double* pp = plane_of(c, (int64_t)z + 1, nz, ghost, zwrap, pe);
    const double* ph = phi ? plane_of(phi, z, nz, ghost, zwrap, pe) : 0;
    for (int y = 0; y < ny; ++y) {
      const int yp = (int)wrap(y + 1, ny);
      for (int x = 0; x < nx; ++x) {
        const int xp = (int)wrap(x + 1, nx);
        const double v = p0[(int64_t)y * nx + x];
        const double a = v - c_alpha, b = c_beta - v, ab = a * b;
        const double dx = p0[(int64_t)y * nx + xp] - v;
        const double dy = p0[(int64_t)yp * nx + x] - v;
        const double dz = pp[(int64_t)y * nx + x] - v;
        nsum(&s[0], &k[0], v);
        nsum(&s[1], &k[1], rho * (ab * ab));
        nsum(&s[2], &k[2], (dx * dx + dy * dy) + dz * dz);
        if (ph) nsum(&s[3], &k[3], v * ph[(int64_t)y * nx + x]);
        if (v < mn) mn = v;
        if (v > mx) mx = v;
      }
    }
  }
  for (int i = 0; i < 4; ++i) out[i] = s[i] + k[i];
  out[4] = mn;
  out[5] = mx;
  return 0;
}

/* BM1 / BM6 initial condition on the lattice x_i = (x0 + i) h, y_j = (y0 + j) h, extruded in z.
 * w0 = 0.105 (BM1, pfbase.py:187) or 0.2 (BM6, pfbase.py:332). */
int orc_ic(double* c, int nx, int ny, int nz, double h, double c0, double amp, double w0) {
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const double X = x * h, Y = y * h;
      const double t2 = cos(0.13 * X) * cos(0.087 * Y);
      const double v = c0 + amp * (cos(w0 * X) * cos(0.11 * Y) + t2 * t2 +
                                   cos(0.025 * X - 0.15 * Y) * cos(0.07 * X - 0.02 * Y));
      for (int z = 0; z < nz; ++z) c[((int64_t)z * ny + y) * nx + x] = v;
    }
  return 0;
}
