// Explicit finite-difference schemes for the multi-field benchmarks (round 2):
//
//   BM2  dolfin/bench2.py:76-111   c_t = M lap(mu), mu = df/dc - kappa_c lap c;  eta_i,t = -L (df/deta_i - kappa_eta lap eta_i)
//        f = f_alpha (1 - h) + f_beta h + w g   (hinterp / double_well, bench2.py:76-96)
//   BM3  dolfin/bench3.py:66-97    tau phi_t = W^2 lap phi + dfdp,  dfdp = (phi - lam U (1 - phi^2)) (1 - phi^2);
//        U_t = D lap U + phi_t / 2                                   (anisotropy switched off there: a = 1)
//
// Same conventions as the BM1 / BM6 grid path: fp64, x fastest, periodic lattice (the reference's no-flux boxes run on
// their even extension), 5-point (2-D) / 7-point (3-D) lap_h, forward Euler, forward-difference discrete energy.  Fields
// are stored as one structure-of-arrays block u[f][cell] per time level (ping-pong = rollback state).  These are plain
// one-thread-per-cell kernels (neighbours through the caches): the benchmark problems are 200^2 / 960^2 cells, far from
// HBM-bound; the LDS-tiled streaming design of ch_fd_kernels.hip is what a 512^3 BM2 would take next.
// The operation order below is restated by oracle/multi_fd.py (numpy, no fma: this file is compiled with
// -ffp-contract=off) and compared bit for bit.
#include <cmath>
#include <string>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {

namespace {

struct MfdParams {
  int model, nf, nx, ny, nz;
  double inv_h2;
  // BM2: ca, cb, rho2, kappa_c, M, kappa_eta, w, alpha, L     BM3: lam, 1/tau, W^2, D
  double q[9];
};

__device__ __forceinline__ int wrapm(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// lap_h u (times h^2) at cell (x, y, z): ((u[x-1] + u[x+1]) + (u[y-1] + u[y+1])) - 4 u  [+ ((u[z-1] + u[z+1]) - 2 u)]
__device__ __forceinline__ double lap_raw(const double* __restrict__ u, int x, int y, int z, int nx, int ny, int nz) {
  const int64_t row = (int64_t)nx, plane = (int64_t)nx * ny;
  const int64_t zc = z * plane, yc = y * row;
  const double c = u[zc + yc + x];
  const double sx = u[zc + yc + wrapm(x - 1, nx)] + u[zc + yc + wrapm(x + 1, nx)];
  const double sy = u[zc + wrapm(y - 1, ny) * row + x] + u[zc + wrapm(y + 1, ny) * row + x];
  double l = (sx + sy) - 4.0 * c;
  if (nz > 1) l = l + ((u[wrapm(z - 1, nz) * plane + yc + x] + u[wrapm(z + 1, nz) * plane + yc + x]) - 2.0 * c);
  return l;
}

__device__ __forceinline__ double hs(double u) { return ((u * u) * u) * ((6.0 * (u * u) - 15.0 * u) + 10.0); }
__device__ __forceinline__ double hsp(double u) { return (30.0 * (u * u)) * ((1.0 - u) * (1.0 - u)); }

// BM2 pass 1: mu = f_c - kappa_c inv_h2 lap_raw(c)
__global__ __launch_bounds__(256) void bm2_mu_kernel(const MfdParams p, const double* __restrict__ u,
                                                     double* __restrict__ mu) {
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3];
  const double c = u[i];
  double h = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) h = h + hs(u[(k + 1) * cells + i]);
  const double fc = (2.0 * r2) * (c - ca) * (1.0 - h) + (2.0 * r2) * (c - cb) * h;
  mu[i] = fc - (kc * p.inv_h2) * lap_raw(u, x, y, z, p.nx, p.ny, p.nz);
}

// BM2 pass 2: c+ = c + (dt M inv_h2) lap_raw(mu);  eta+ = eta - (dt L) (f_eta - kappa_eta inv_h2 lap_raw(eta))
__global__ __launch_bounds__(256) void bm2_update_kernel(const MfdParams p, const double* __restrict__ u,
                                                         const double* __restrict__ mu, double* __restrict__ un,
                                                         double dt) {
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], Mob = p.q[4], ke = p.q[5], w = p.q[6], al = p.q[7], L = p.q[8];
  const double c = u[i];
  un[i] = c + (dt * Mob * p.inv_h2) * lap_raw(mu, x, y, z, p.nx, p.ny, p.nz);
  double e[4], e2 = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] = u[(k + 1) * cells + i];
    e2 = e2 + e[k] * e[k];
  }
  const double dfab = r2 * ((c - cb) * (c - cb)) - r2 * ((c - ca) * (c - ca));  // f_beta - f_alpha
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double ek = e[k];
    const double well = (2.0 * ek * ((1.0 - ek) * (1.0 - ek)) - 2.0 * (ek * ek) * (1.0 - ek)) + (2.0 * al) * ek * (e2 - ek * ek);
    const double fe = dfab * hsp(ek) + w * well;
    const double lap = lap_raw(u + (k + 1) * cells, x, y, z, p.nx, p.ny, p.nz);
    un[(k + 1) * cells + i] = ek - (dt * L) * (fe - (ke * p.inv_h2) * lap);
  }
}

// BM3: phi_t = (1/tau) (W^2 inv_h2 lap_raw(phi) + dfdp);  phi+ = phi + dt phi_t;  U+ = U + dt (D inv_h2 lap_raw(U) + phi_t / 2)
__global__ __launch_bounds__(256) void bm3_update_kernel(const MfdParams p, const double* __restrict__ u,
                                                         double* __restrict__ un, double dt) {
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double lam = p.q[0], it = p.q[1], W2 = p.q[2], D = p.q[3];
  const double U = u[i], ph = u[cells + i];
  const double P = 1.0 - ph * ph;
  const double dfdp = (ph - (lam * U) * P) * P;
  const double pt = it * ((W2 * p.inv_h2) * lap_raw(u + cells, x, y, z, p.nx, p.ny, p.nz) + dfdp);
  un[cells + i] = ph + dt * pt;
  un[i] = U + dt * ((D * p.inv_h2) * lap_raw(u, x, y, z, p.nx, p.ny, p.nz) + 0.5 * pt);
}

// diagnostics, raw sums per block: {sum second (BM2: c, BM3: (phi+1)/2), sum f_chem, sum of weighted squared forward
// differences (sum_f gradc_f |fwd diff u_f|^2), min over all fields, max over all fields}
__device__ __forceinline__ double fwd2(const double* __restrict__ u, int x, int y, int z, int nx, int ny, int nz) {
  const int64_t row = (int64_t)nx, plane = (int64_t)nx * ny;
  const int64_t zc = z * plane, yc = y * row;
  const double c = u[zc + yc + x];
  const double dx = u[zc + yc + wrapm(x + 1, nx)] - c, dy = u[zc + wrapm(y + 1, ny) * row + x] - c;
  double g = dx * dx + dy * dy;
  if (nz > 1) {
    const double dz = u[wrapm(z + 1, nz) * plane + yc + x] - c;
    g = g + dz * dz;
  }
  return g;
}

__global__ __launch_bounds__(256) void mfd_diag_kernel(const MfdParams p, const double* __restrict__ u,
                                                       double* __restrict__ partials) {
  __shared__ double sh[5][4];
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz;
  double v[5] = {0.0, 0.0, 0.0, INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
    if (p.model == 2) {
      const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3], ke = p.q[5], w = p.q[6], al = p.q[7];
      const double c = u[i];
      double e[4], h = 0.0, g = 0.0;
      for (int k = 0; k < 4; ++k) e[k] = u[(k + 1) * cells + i];
      for (int k = 0; k < 4; ++k) {
        h = h + hs(e[k]);
        g = g + (e[k] * e[k]) * ((1.0 - e[k]) * (1.0 - e[k]));
        for (int j = k + 1; j < 4; ++j) g = g + al * ((e[k] * e[k]) * (e[j] * e[j]));
      }
      const double fa = r2 * ((c - ca) * (c - ca)), fb = r2 * ((c - cb) * (c - cb));
      v[0] += c;
      v[1] += (fa * (1.0 - h) + fb * h) + w * g;
      double gr = kc * fwd2(u, x, y, z, p.nx, p.ny, p.nz);
      for (int k = 0; k < 4; ++k) gr = gr + ke * fwd2(u + (k + 1) * cells, x, y, z, p.nx, p.ny, p.nz);
      v[2] += gr;
      v[3] = fmin(v[3], c);
      v[4] = fmax(v[4], c);
      for (int k = 0; k < 4; ++k) {
        v[3] = fmin(v[3], e[k]);
        v[4] = fmax(v[4], e[k]);
      }
    } else {
      const double lam = p.q[0], W2 = p.q[2];
      const double U = u[i], ph = u[cells + i], p2 = ph * ph;
      v[0] += 0.5 * (ph + 1.0);
      v[1] += (-0.5 * p2 + 0.25 * (p2 * p2)) + (lam * U) * ph * ((1.0 - (2.0 / 3.0) * p2) + 0.2 * (p2 * p2));
      v[2] += W2 * fwd2(u + cells, x, y, z, p.nx, p.ny, p.nz);
      v[3] = fmin(v[3], fmin(U, ph));
      v[4] = fmax(v[4], fmax(U, ph));
    }
  }
  for (int k = 0; k < 5; ++k) {
    double a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double b = __shfl_down(a, o, 64);
      a = k < 3 ? a + b : (k == 3 ? fmin(a, b) : fmax(a, b));
    }
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    double a;
    if (k < 3)
      a = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
    else if (k == 3)
      a = fmin(fmin(sh[k][0], sh[k][1]), fmin(sh[k][2], sh[k][3]));
    else
      a = fmax(fmax(sh[k][0], sh[k][1]), fmax(sh[k][2], sh[k][3]));
    partials[(int64_t)blockIdx.x * 5 + k] = a;
  }
}

__global__ void mfd_diag_final_kernel(const double* __restrict__ partials, int nb, double* __restrict__ out5) {
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    double a = k < 3 ? 0.0 : (k == 3 ? INFINITY : -INFINITY);
    for (int b = 0; b < nb; ++b) {
      const double v = partials[(int64_t)b * 5 + k];
      a = k < 3 ? a + v : (k == 3 ? fmin(a, v) : fmax(a, v));
    }
    out5[k] = a;
  }
}

// initial conditions on the lattice (z-extruded); mnx / mny > 0: even extension of a no-flux domain with that many nodes
__global__ __launch_bounds__(256) void mfd_ic_kernel(const MfdParams p, double* __restrict__ u, double h, int mnx, int mny,
                                                     double a0, double a1, double a2, double a3, double a4) {
  const int64_t plane = (int64_t)p.nx * p.ny, cells = plane * p.nz;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane) return;
  const int x = (int)(i % p.nx), y = (int)(i / p.nx);
  const int xe = (mnx > 0 && x >= mnx) ? 2 * (mnx - 1) - x : x;
  const int ye = (mny > 0 && y >= mny) ? 2 * (mny - 1) - y : y;
  const double X = xe * h, Y = ye * h;
  double val[5];
  int nf;
  if (p.model == 2) {  // pfbase.py:268-296; a0..a3 = c0, eps, eps_eta, psi
    nf = 5;
    const double t2 = cos(0.13 * X) * cos(0.087 * Y);
    val[0] = a0 + a1 * (cos(0.105 * X) * cos(0.11 * Y) + t2 * t2 + cos(0.025 * X - 0.15 * Y) * cos(0.07 * X - 0.02 * Y));
    for (int k = 0; k < 4; ++k) {
      const double ii = k + 1.0, i0 = (double)k;
      const double a = cos((0.01 * ii) * X - 4.0) * cos((0.007 + 0.01 * ii) * Y);
      const double b = cos((0.11 + 0.01 * ii) * X) * cos((0.11 + 0.01 * ii) * Y);
      const double cc = cos((0.046 + 0.001 * i0) * X - (0.0405 + 0.001 * i0) * Y) * cos((0.031 + 0.001 * i0) * X - (0.004 + 0.001 * i0) * Y);
      const double sum = (a + b) + a3 * (cc * cc);
      val[1 + k] = a2 * (sum * sum);
    }
  } else {  // pfbase.py:298-320; a0..a4 = Delta, r, w, vin, vout
    nf = 2;
    const double r = sqrt(X * X + Y * Y);
    val[0] = a0;
    if (r < a1 - 0.5 * a2)
      val[1] = a3;
    else if (r > a1 + 0.5 * a2)
      val[1] = a4;
    else
      val[1] = a4 + 0.5 * (a3 - a4) * (1.0 + cos(3.14159265358979323846 * (r - a1 + 0.5 * a2) / a2));
  }
  for (int f = 0; f < nf; ++f)
    for (int z = 0; z < p.nz; ++z) u[f * cells + z * plane + i] = val[f];
}

}  // namespace

struct MultiFD {
  MfdParams p;
  int64_t cells = 0;
  double h = 1.0;
  double* u[2] = {nullptr, nullptr};
  double* mu = nullptr;
  double *partials = nullptr, *out5 = nullptr, *out5_host = nullptr;
  int cur = 0;
  bool have_prev = false;
  hipStream_t stream = nullptr;
  std::string err;
};

#define MF_HIP(expr)                                                     \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess) {                                              \
      mf->err = std::string(#expr) + ": " + hipGetErrorString(e_);       \
      return -3;                                                         \
    }                                                                    \
  } while (0)

const char* multifd_error(const MultiFD* mf) { return mf->err.c_str(); }
int multifd_nfields(const MultiFD* mf) { return mf->p.nf; }

// model 2: mp = {c_alpha, c_beta, rho, kappa_c, M, kappa_eta, w, alpha, L}; model 3: mp = {W0, tau0, D, Delta}
int multifd_create(MultiFD** out, int model, int nx, int ny, int nz, double h, const double* mp, hipStream_t stream,
                   std::string* err) {
  MultiFD* mf = new MultiFD();
  *out = mf;
  MfdParams& p = mf->p;
  p.model = model;
  p.nf = model == 2 ? 5 : 2;
  p.nx = nx;
  p.ny = ny;
  p.nz = nz;
  p.inv_h2 = 1.0 / (h * h);
  for (double& q : p.q) q = 0.0;
  if (model == 2) {
    for (int i = 0; i < 9; ++i) p.q[i] = mp[i];
    p.q[2] = mp[2] * mp[2];  // rho^2
  } else {
    p.q[0] = mp[2] * mp[1] / (0.6267 * mp[0] * mp[0]);  // lam = D tau0 / (0.6267 W0^2)
    p.q[1] = 1.0 / mp[1];
    p.q[2] = mp[0] * mp[0];
    p.q[3] = mp[2];
  }
  mf->h = h;
  mf->cells = (int64_t)nx * ny * nz;
  mf->stream = stream;
  auto body = [&]() -> int {
    const size_t bytes = sizeof(double) * (size_t)mf->cells * p.nf;
    MF_HIP(hipMalloc(&mf->u[0], bytes));
    MF_HIP(hipMalloc(&mf->u[1], bytes));
    MF_HIP(hipMemsetAsync(mf->u[0], 0, bytes, stream));
    MF_HIP(hipMemsetAsync(mf->u[1], 0, bytes, stream));
    if (model == 2) MF_HIP(hipMalloc(&mf->mu, sizeof(double) * (size_t)mf->cells));
    MF_HIP(hipMalloc(&mf->partials, sizeof(double) * 5 * 1024));
    MF_HIP(hipMalloc(&mf->out5, sizeof(double) * 8));
    MF_HIP(hipHostMalloc(&mf->out5_host, sizeof(double) * 8, hipHostMallocDefault));
    return 0;
  };
  int rc = body();
  if (rc && err) *err = mf->err;
  return rc;
}

void multifd_destroy(MultiFD* mf) {
  if (!mf) return;
  for (void* q : {(void*)mf->u[0], (void*)mf->u[1], (void*)mf->mu, (void*)mf->partials, (void*)mf->out5})
    if (q) (void)hipFree(q);
  if (mf->out5_host) (void)hipHostFree(mf->out5_host);
  delete mf;
}

// a: BM2 {c0, eps, eps_eta, psi, -}; BM3 {Delta, r, w, vin, vout}
int multifd_set_ic(MultiFD* mf, int mnx, int mny, const double* a) {
  const MfdParams& p = mf->p;
  const int64_t plane = (int64_t)p.nx * p.ny;
  hipLaunchKernelGGL(mfd_ic_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, mf->stream, p, mf->u[mf->cur],
                     mf->h, mnx, mny, a[0], a[1], a[2], a[3], a[4]);
  MF_HIP(hipGetLastError());
  mf->have_prev = false;
  return 0;
}

double* multifd_field_ptr(MultiFD* mf, int f) { return mf->u[mf->cur] + (int64_t)f * mf->cells; }
void multifd_touch(MultiFD* mf) { mf->have_prev = false; }

int multifd_step(MultiFD* mf, double dt, int nsteps) {
  const MfdParams& p = mf->p;
  const unsigned nb = (unsigned)((mf->cells + 255) / 256);
  for (int s = 0; s < nsteps; ++s) {
    const double* u = mf->u[mf->cur];
    double* un = mf->u[1 - mf->cur];
    if (p.model == 2) {
      hipLaunchKernelGGL(bm2_mu_kernel, dim3(nb), dim3(256), 0, mf->stream, p, u, mf->mu);
      hipLaunchKernelGGL(bm2_update_kernel, dim3(nb), dim3(256), 0, mf->stream, p, u, (const double*)mf->mu, un, dt);
    } else {
      hipLaunchKernelGGL(bm3_update_kernel, dim3(nb), dim3(256), 0, mf->stream, p, u, un, dt);
    }
    mf->cur ^= 1;
    mf->have_prev = true;
  }
  MF_HIP(hipGetLastError());
  return 0;
}

int multifd_rollback(MultiFD* mf) {
  if (!mf->have_prev) return -4;
  mf->cur ^= 1;
  mf->have_prev = false;
  return 0;
}

// raw[5] = {sum second, sum f_chem, weighted sum of squared forward differences, min, max}; synchronises
int multifd_diag_raw(MultiFD* mf, double raw[5]) {
  const MfdParams& p = mf->p;
  int nb = (int)((mf->cells + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(mfd_diag_kernel, dim3(nb), dim3(256), 0, mf->stream, p, (const double*)mf->u[mf->cur], mf->partials);
  hipLaunchKernelGGL(mfd_diag_final_kernel, dim3(1), dim3(64), 0, mf->stream, (const double*)mf->partials, nb, mf->out5);
  MF_HIP(hipMemcpyAsync(mf->out5_host, mf->out5, 5 * sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  MF_HIP(hipStreamSynchronize(mf->stream));
  for (int k = 0; k < 5; ++k) raw[k] = mf->out5_host[k];
  return 0;
}

}  // namespace pfhip
