/* ORACLE self-check under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU
 * build only).  Includes ch_fd.c itself and drives every entry point over the edge shapes the GPU tests use (1-cell
 * periodic wraps, ghost mode with plane sub-ranges, phi coupling, phi eliminated) on EXACTLY-sized heap buffers, so any
 * out-of-range index is an ASan report.  Prints a checksum; the Makefile builds the same driver without sanitizers and
 * tests/test_oracle_fd.py compares the two outputs.  Test infrastructure, like the rest of oracle/. */
#include "ch_fd.c"

#include <stdio.h>

static double* filled(int64_t n, unsigned seed) {
  double* p = (double*)malloc(sizeof(double) * (size_t)n);
  if (!p) abort();
  unsigned s = seed * 2654435761u + 12345u;
  for (int64_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    p[i] = 0.5 + 0.05 * ((double)(s >> 8) / 16777216.0 - 0.5);
  }
  return p;
}

static double checksum(const double* p, int64_t n) {
  double a = 0.0;
  for (int64_t i = 0; i < n; ++i) a = fma(a, 0.999, p[i]);
  return a;
}

int main(void) {
  static const int shapes[][3] = {{1, 1, 1}, {2, 1, 1}, {1, 2, 1}, {1, 1, 2}, {2, 2, 2}, {3, 5, 7}, {4, 3, 2},
                                  {130, 17, 5}, {64, 16, 4}, {6, 2, 3}, {33, 9, 1}};
  const orc_ch_params bm1 = {0.3, 0.7, 10.0, 2.0, 5e-3, 0.0, 0.0, 0.0};
  const orc_ch_params bm6 = {0.3, 0.7, 10.0, 2.0, 5e-3, 0.09, 0.0, 0.0};
  const orc_ch_params elim = {0.3, 0.7, 10.0, 2.0, 5e-3, 0.0, -4.5e-7, 0.5};
  double total = 0.0;
  int cases = 0;
  for (unsigned k = 0; k < sizeof(shapes) / sizeof(shapes[0]); ++k) {
    const int nx = shapes[k][0], ny = shapes[k][1], nz = shapes[k][2];
    const int64_t pe = (int64_t)nx * ny;
    /* periodic inside the buffer (ghost = 0, zwrap = 1) */
    {
      double* c = filled(pe * nz, k + 1);
      double* phi = filled(pe * nz, k + 101);
      double* out = (double*)calloc((size_t)(pe * nz), sizeof(double));
      double* mu = (double*)calloc((size_t)(pe * nz), sizeof(double));
      double d[6];
      if (orc_ch_fd_step(c, out, NULL, nx, ny, nz, 0, 1, 0, nz, &bm1)) return 2;
      total += checksum(out, pe * nz);
      if (orc_ch_fd_step(c, out, phi, nx, ny, nz, 0, 1, 0, nz, &bm6)) return 2;
      total += checksum(out, pe * nz);
      if (orc_ch_fd_step(c, out, NULL, nx, ny, nz, 0, 1, 0, nz, &elim)) return 2;
      total += checksum(out, pe * nz);
      if (orc_ch_mu(c, mu, phi, nx, ny, nz, 0, 1, &bm6)) return 2;
      total += checksum(mu, pe * nz);
      if (orc_ch_diag(c, phi, nx, ny, nz, 0, 1, 5.0, 0.3, 0.7, d)) return 2;
      total += d[0] + d[1] + d[2] + d[3] + d[4] + d[5];
      if (orc_ic(c, nx, ny, nz, 1.0, 0.5, 0.05, 0.105)) return 2;
      total += checksum(c, pe * nz);
      free(c); free(phi); free(out); free(mu);
      cases += 6;
    }
    /* ghost mode (2 ghost planes per side, zwrap = 0): whole range, then interior / boundary sub-ranges */
    {
      const int g = 2;
      double* c = filled(pe * (nz + 2 * g), k + 201);
      double* phi = filled(pe * (nz + 2 * g), k + 301);
      double* out = (double*)calloc((size_t)(pe * (nz + 2 * g)), sizeof(double));
      double d[6];
      if (orc_ch_fd_step(c, out, phi, nx, ny, nz, g, 0, 0, nz, &bm6)) return 2;
      total += checksum(out, pe * (nz + 2 * g));
      const int lo = nz > 4 ? 2 : 0, hi = nz > 4 ? nz - 2 : nz;
      if (orc_ch_fd_step(c, out, NULL, nx, ny, nz, g, 0, lo, hi, &bm1)) return 2;
      if (orc_ch_fd_step(c, out, NULL, nx, ny, nz, g, 0, 0, lo, &bm1)) return 2;
      if (orc_ch_fd_step(c, out, NULL, nx, ny, nz, g, 0, hi, nz, &bm1)) return 2;
      total += checksum(out, pe * (nz + 2 * g));
      if (orc_ch_diag(c, NULL, nx, ny, nz, g, 0, 5.0, 0.3, 0.7, d)) return 2;
      total += d[0] + d[1] + d[2];
      free(c); free(phi); free(out);
      cases += 5;
    }
  }
  /* argument checking: refused, not executed */
  {
    double x[8] = {0};
    if (orc_ch_fd_step(x, x, NULL, 0, 1, 1, 0, 1, 0, 1, &bm1) != -1) return 3;
    if (orc_ch_fd_step(x, x, NULL, 1, 1, 1, 1, 0, 0, 1, &bm1) != -1) return 3; /* ghost < 2 without zwrap */
    if (orc_ch_fd_step(x, x, NULL, 1, 1, 1, 0, 1, 0, 2, &bm1) != -1) return 3; /* zhi > nz */
  }
  printf("SANITIZE_OK cases=%d checksum=%.17g\n", cases, total);
  return 0;
}
