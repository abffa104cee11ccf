// Poisson solve of PFHub BM6:  lap(phi) = -k c / eps      (dolfin/pfbase.py:410-421 poisson_weak_form, called at
// dolfin/bench6.py:72 with f = -k*c/epsilon, M = 1).  Fast direct solver (hand-written passes, or rocFFT through its native
// API, fftplan.hip, for sizes they do not cover): the 5/7-point Laplacian is
// diagonal in the discrete Fourier basis of the periodic lattice the solver runs on.
//
// Two boundary modes:
//  PERIODIC   (performance configs, no counterpart in the reference): fully periodic box; the k = 0 mode of the
//             right-hand side is dropped (uniform neutralising background), phi has zero mean.
//  DIRICHLET_X (the reference's BM6, bench6.py:77-90): phi = 0 on x = 0, phi = sin(y/7) on x = Lx, no-flux on the other
//             faces.  c lives on the even (mirror) extension of the N+1-node domain (2N-periodic lattice).  With the
//             boundary values moved to the right-hand side, the homogeneous Dirichlet-x / Neumann-y operator is
//             diagonalised by a sine transform in x and a cosine transform in y = the FFT of the ODD-in-x, EVEN-in-y
//             extension on the same lattice.  The result is handed to the Cahn-Hilliard kernel as the EVEN-in-x
//             extension of phi (mu must stay mirror symmetric), with the Dirichlet values written on x = 0, Lx.
#include "pfhip_internal.h"

namespace pfhip {

namespace {

constexpr double TWO_PI_P = 6.283185307179586476925286766559;

struct PoArgs {
  int nx, ny, nz, nxh;  // lattice extents, nxh = nx/2+1
  int npx;              // DIRICHLET_X: nodes per side of the physical domain in x (lattice nx = 2 (npx-1)); else 0
  int npy;              // same for y (needed for the boundary function's argument)
  double h, k_over_eps, inv_h2, inv_n;
};

__device__ __forceinline__ double bdry_right(double y) { return sin(y / 7.0); }  // bench6.py:84 phi_right

// odd-in-x right-hand side:  s(i) * ( -(k/eps) c  -  [node == N-1] sin(y/7)/h^2 ),  s = 0 on the Dirichlet planes
__global__ __launch_bounds__(256) void rhs_dirichlet_kernel(const double* __restrict__ c, double* __restrict__ r,
                                                            const PoArgs a) {
  const int64_t n = (int64_t)a.nx * a.ny * a.nz;
  const int N = a.npx - 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % a.nx);
    const int y = (int)((i / a.nx) % a.ny);
    const int xr = x <= N ? x : 2 * N - x;  // physical node
    const int yr = y < a.npy ? y : 2 * (a.npy - 1) - y;
    double v = 0.0;
    if (xr != 0 && xr != N) {
      v = -a.k_over_eps * c[i];
      if (xr == N - 1) v -= bdry_right(yr * a.h) * a.inv_h2;
      if (x > N) v = -v;
    }
    r[i] = v;
  }
}

// phi_hat = rhs_hat / lambda, lambda = sum_d (2 cos(2 pi m_d / n_d) - 2) / h^2 ; zero mode -> 0 ; folded 1/N
// rhs_is_c: the transform input was c itself (PERIODIC mode): multiply by -(k/eps)
__global__ __launch_bounds__(256) void invert_laplacian_kernel(double2* __restrict__ ph, int64_t nh, const PoArgs a,
                                                               int rhs_is_c) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (int64_t)gridDim.x * 256) {
    const int mx = (int)(i % a.nxh);
    const int64_t rr = i / a.nxh;
    const int my = (int)(rr % a.ny), mz = (int)(rr / a.ny);
    double lam = (2.0 * cos(TWO_PI_P * mx / a.nx) - 2.0) + (2.0 * cos(TWO_PI_P * my / a.ny) - 2.0);
    if (a.nz > 1) lam += 2.0 * cos(TWO_PI_P * mz / a.nz) - 2.0;
    lam *= a.inv_h2;
    double s = (mx == 0 && my == 0 && mz == 0) ? 0.0 : a.inv_n / lam;
    if (rhs_is_c) s *= -a.k_over_eps;
    const double2 v = ph[i];
    ph[i] = make_double2(v.x * s, v.y * s);
  }
}

// odd-in-x solution -> even-in-x extension with the Dirichlet values on x = 0 and x = Lx
__global__ __launch_bounds__(256) void fixup_dirichlet_kernel(double* __restrict__ phi, const PoArgs a) {
  const int64_t n = (int64_t)a.nx * a.ny * a.nz;
  const int N = a.npx - 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % a.nx);
    const int y = (int)((i / a.nx) % a.ny);
    const int yr = y < a.npy ? y : 2 * (a.npy - 1) - y;
    double v = phi[i];
    if (x == 0)
      v = 0.0;
    else if (x == N)
      v = bdry_right(yr * a.h);
    else if (x > N)
      v = -v;
    phi[i] = v;
  }
}

int grid_for_p(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

struct Poisson {
  PoArgs a;
  int64_t n, nh;
  FftPlan *fwd = nullptr, *inv = nullptr;  // library transforms (null on the hand-written path)
  double* rhs = nullptr;
  double2* ph = nullptr;
  Fused2D* fast = nullptr;  // hand-written LDS-FFT passes (spectral2d_fused.hip) instead of library transforms
  bool dirichlet_fast = false;  // reference BCs by the sine / cosine passes on the physical nodes (fused_poisson_dirichlet)
  double* S = nullptr;          // their work array
  std::string err;
};

#define PO_HIP(expr)                                                       \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      po->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
      return -3;                                                           \
    }                                                                      \
  } while (0)
#define PO_FFT(expr)                                                       \
  do {                                                                     \
    if ((expr) != 0) return -3; /* po->err was filled by the fftplan_* call */ \
  } while (0)

const char* poisson_error(const Poisson* po) { return po->err.c_str(); }
// which transforms run (for pf_status_string)
const char* poisson_path(const Poisson* po) { return po->fast ? fused2d_describe(po->fast) : "rocFFT (native API) + pointwise kernels"; }

// npx, npy > 0 selects DIRICHLET_X on the even extension of an npx x npy(-node) domain; 0 = PERIODIC
int poisson_create(Poisson** out, int dim, int nx, int ny, int nz, int npx, int npy, double h, double k, double eps,
                   hipStream_t stream, std::string* err) {
  Poisson* po = new Poisson();
  *out = po;
  PoArgs& a = po->a;
  a.nx = nx;
  a.ny = ny;
  a.nz = dim == 3 ? nz : 1;
  a.nxh = nx / 2 + 1;
  a.npx = npx;
  a.npy = npy > 0 ? npy : ny;
  a.h = h;
  a.k_over_eps = k / eps;
  a.inv_h2 = 1.0 / (h * h);
  po->n = (int64_t)nx * ny * a.nz;
  po->nh = (int64_t)a.nxh * ny * a.nz;
  a.inv_n = 1.0 / (double)po->n;
  auto body = [&]() -> int {
    const bool fast_dir = npx > 0 && fused_dirichlet_supported(dim, nx, ny, a.nz);  // reference BCs: sine / cosine passes
    const bool fast = fast_dir || (npx == 0 && fused2d_supported(dim, nx, ny, a.nz));  // periodic boxes: hand-written passes
    po->dirichlet_fast = fast_dir;
    if (!fast) {  // the hand-written passes need no library plans
      const int nn[3] = {nx, ny, a.nz};
      PO_FFT(fftplan_real(&po->fwd, dim, nn, 1, true, stream, &po->err));
      PO_FFT(fftplan_real(&po->inv, dim, nn, 1, false, stream, &po->err));
    }
    if (fast_dir) {
      PO_HIP(pf_malloc(&po->S, sizeof(double) * fused_dirichlet_work_doubles(npx, a.npy, dim == 3 ? nz / 2 + 1 : 1)));
      PO_HIP(hipMemsetAsync(po->S, 0, sizeof(double) * fused_dirichlet_work_doubles(npx, a.npy, dim == 3 ? nz / 2 + 1 : 1), stream));
    } else {  // the hand-written passes use a padded row pitch (fused_spectrum_pitch); the rocFFT path the natural one
      const SpecLayout lay = fused_spectrum_layout(dim, nx, ny, a.nz);
      const int64_t nh_alloc = fast ? (int64_t)lay.pitch * lay.rows : po->nh;
      PO_HIP(pf_malloc(&po->ph, sizeof(double2) * nh_alloc));
      if (nh_alloc != po->nh) PO_HIP(hipMemsetAsync(po->ph, 0, sizeof(double2) * nh_alloc, stream));
    }
    if (npx > 0 && !fast_dir) PO_HIP(pf_malloc(&po->rhs, sizeof(double) * po->n));
    if (fast && fused2d_create(&po->fast, nx, ny, a.nz, h, stream, fast_dir ? 1 : 0) != 0) {
      po->err = "fused2d_create failed";
      return -3;
    }
    return 0;
  };
  int rc = body();
  if (rc && err) *err = po->err;
  return rc;
}

// The lattice kernels of the reference's boundary conditions on a SLAB of `nzl` lattice planes (slab mode: the periodic slab
// FFT transforms the odd-in-x / even-in-y,z extension; the boundary function depends on y only, so a slab needs no z index):
// r <- odd-in-x right-hand side of c (even extension), Dirichlet data of x = Lx moved to node N - 1; and, after the solve,
// phi <- even-in-x extension with the Dirichlet values written on x = 0, Lx.  dolfin/bench6.py:77-90, pfbase.py:410-421.
int poisson_dirichlet_rhs_planes(const double* c, double* r, int nx, int ny, int nzl, int npx, int npy, double h,
                                 double k_over_eps, hipStream_t stream) {
  PoArgs a{};
  a.nx = nx;
  a.ny = ny;
  a.nz = nzl;
  a.nxh = nx / 2 + 1;
  a.npx = npx;
  a.npy = npy;
  a.h = h;
  a.k_over_eps = k_over_eps;
  a.inv_h2 = 1.0 / (h * h);
  hipLaunchKernelGGL(rhs_dirichlet_kernel, dim3(grid_for_p((int64_t)nx * ny * nzl)), dim3(256), 0, stream, c, r, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int poisson_dirichlet_fixup_planes(double* phi, int nx, int ny, int nzl, int npx, int npy, double h, hipStream_t stream) {
  PoArgs a{};
  a.nx = nx;
  a.ny = ny;
  a.nz = nzl;
  a.npx = npx;
  a.npy = npy;
  a.h = h;
  hipLaunchKernelGGL(fixup_dirichlet_kernel, dim3(grid_for_p((int64_t)nx * ny * nzl)), dim3(256), 0, stream, phi, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

void poisson_destroy(Poisson* po) {
  if (!po) return;
  fftplan_destroy(po->fwd);
  fftplan_destroy(po->inv);
  if (po->rhs) (void)pf_free(po->rhs);
  if (po->S) (void)pf_free(po->S);
  if (po->ph) (void)pf_free(po->ph);
  if (po->fast) fused2d_destroy(po->fast);
  delete po;
}

// phi <- solution for the given c (both on the lattice, no ghost planes)
int poisson_solve(Poisson* po, const double* c, double* phi, hipStream_t stream) {
  const PoArgs& a = po->a;
  if (po->dirichlet_fast) {
    if (fused_poisson_dirichlet(po->fast, c, phi, po->S, a.npx, a.npy, a.nz > 1 ? a.nz / 2 + 1 : 1, a.h, a.k_over_eps) != 0) {
      po->err = "fused_poisson_dirichlet launch failed";
      return -3;
    }
  } else if (a.npx > 0) {
    hipLaunchKernelGGL(rhs_dirichlet_kernel, dim3(grid_for_p(po->n)), dim3(256), 0, stream, c, po->rhs, a);
    PO_FFT(fftplan_exec(po->fwd, po->rhs, po->ph, &po->err));
    hipLaunchKernelGGL(invert_laplacian_kernel, dim3(grid_for_p(po->nh)), dim3(256), 0, stream, po->ph, po->nh, a, 0);
    PO_FFT(fftplan_exec(po->inv, po->ph, phi, &po->err));
    hipLaunchKernelGGL(fixup_dirichlet_kernel, dim3(grid_for_p(po->n)), dim3(256), 0, stream, phi, a);
  } else if (po->fast) {
    if (fused3d_poisson(po->fast, c, phi, po->ph, a.k_over_eps, a.inv_h2) != 0) {
      po->err = "fused3d_poisson launch failed";
      return -3;
    }
  } else {
    PO_FFT(fftplan_exec(po->fwd, const_cast<double*>(c), po->ph, &po->err));
    hipLaunchKernelGGL(invert_laplacian_kernel, dim3(grid_for_p(po->nh)), dim3(256), 0, stream, po->ph, po->nh, a, 1);
    PO_FFT(fftplan_exec(po->inv, po->ph, phi, &po->err));
  }
  PO_HIP(hipGetLastError());
  return 0;
}

}  // namespace pfhip
