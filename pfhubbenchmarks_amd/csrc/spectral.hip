// Semi-implicit Fourier-spectral Cahn-Hilliard step (BASELINE.json config 2: 512^2, fp64): HIP k-space kernels around
// library transforms -- rocFFT through its native API (fftplan.hip) -- for the sizes the hand-written passes of
// spectral2d_fused.hip do not cover.
//
//   c_t = M lap( f'(c) - kappa lap c )        dolfin/pfbase.py:361-383, f' from dolfin/bench1.py:63-65
//   (c^+_k - c_k)/dt = -M k^2 N_k - M kappa k^4 c^+_k,   N = f'(c^n)   (stiff term implicit, nonlinearity explicit)
//   =>  c^+_k = (c_k - dt M k^2 N_k) / (1 + dt M kappa k^4)
//
// Per step: [HIP] g = f'(c)  ->  rocFFT r2c(g)  ->  [HIP] k-space update (c_k stays resident; also emits c_k/N for
// the inverse)  ->  rocFFT c2r.  The inverse transform's input is a scratch copy because multi-dimensional c2r
// may overwrite its input.  The same pointwise f' as the FD kernel: a = c-ca; b = cb-c; 2 rho ((a b)(b-a)).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {

namespace {

constexpr double TWO_PI = 6.283185307179586476925286766559;

__global__ __launch_bounds__(256) void dfdc_kernel(const double* __restrict__ c, double* __restrict__ g, int64_t n,
                                                   double ca, double cb, double two_rho) {
  // n is even whenever nx is; handle pairs with 16-byte accesses, tail scalar
  const int64_t npair = n >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (int64_t)gridDim.x * 256) {
    const double2 v = reinterpret_cast<const double2*>(c)[i];
    double2 o;
    double a = v.x - ca, b = cb - v.x;
    o.x = two_rho * ((a * b) * (b - a));
    a = v.y - ca;
    b = cb - v.y;
    o.y = two_rho * ((a * b) * (b - a));
    reinterpret_cast<double2*>(g)[i] = o;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double w = c[n - 1];
    const double a = w - ca, b = cb - w;
    g[n - 1] = two_rho * ((a * b) * (b - a));
  }
}

struct KsArgs {
  int nxh, ny, nz;   // half-spectrum extents (x fastest)
  int pitch;         // complex elements per k_x row in memory (nxh, or 264 on the hand-written 512^3 path)
  int nyp = 0;       // rows reserved per z-plane (ny + pad rows on the hand-written 3-D path; 0 = ny)
  int nx;            // full x extent
  double kx0, ky0, kz0;  // 2 pi / (n h) per axis
  double dtM, dtMkappa, inv_n;
  double gam = 0.0;  // BM6: dt M k_c^2 / eps, the screened-Poisson term treated implicitly (0 for BM1)
};

__device__ __forceinline__ double ksq(const KsArgs& a, int64_t idx) {
  const int mx = (int)(idx % a.nxh);
  const int64_t r = idx / a.nxh;
  int my = (int)(r % a.ny);
  int mz = (int)(r / a.ny);
  if (2 * my > a.ny) my -= a.ny;
  if (2 * mz > a.nz) mz -= a.nz;
  const double kx = a.kx0 * mx, ky = a.ky0 * my, kz = a.kz0 * mz;
  return (kx * kx + ky * ky) + kz * kz;
}

// chat <- (chat - dtM k^2 ghat) / (1 + dtM kappa k^4);  scratch <- chat / N
__global__ __launch_bounds__(256) void kspace_update_kernel(double2* __restrict__ chat, const double2* __restrict__ ghat,
                                                            double2* __restrict__ scratch, int64_t nh, const KsArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (int64_t)gridDim.x * 256) {
    const double k2 = ksq(a, i);
    const double num = a.dtM * k2;
    const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
    const double2 ch = chat[i], gh = ghat[i];
    double2 o;
    o.x = fma(-num, gh.x, ch.x) * den;
    o.y = fma(-num, gh.y, ch.y) * den;
    chat[i] = o;
    scratch[i] = make_double2(o.x * a.inv_n, o.y * a.inv_n);
  }
}

// sum_k w_k k^2 |chat_k|^2 and sum_{k != 0} w_k |chat_k|^2 / k^2 over the half spectrum (w = 1 on the self-conjugate
// x-columns mx = 0 and mx = nx/2, 2 elsewhere) -> per-block partials -> final (fixed order, deterministic).  The second
// sum is the electrostatic energy of BM6 in the spectral scheme: sum_x c phi = (k/eps) / N * it.
__global__ __launch_bounds__(256) void kspace_grad_energy_kernel(const double2* __restrict__ chat, int64_t nh,
                                                                 const KsArgs a, double* __restrict__ partials) {
  __shared__ double sh[8];
  double acc = 0.0, acc2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (int64_t)gridDim.x * 256) {
    const int mx = (int)(i % a.nxh);
    const double w = (mx == 0 || 2 * mx == a.nx) ? 1.0 : 2.0;
    const int64_t row = i / a.nxh;                         // i runs over the logical half spectrum: row = z ny + y
    const int64_t z = a.nyp ? row / a.ny : 0;              // (pad rows per plane: SpecLayout)
    const double2 ch = chat[(row + z * (a.nyp ? a.nyp - a.ny : 0)) * a.pitch + mx];
    const double k2 = ksq(a, i), m2 = ch.x * ch.x + ch.y * ch.y;
    acc += w * k2 * m2;
    if (k2 > 0.0) acc2 += w * m2 / k2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_down(acc, o, 64);
    acc2 += __shfl_down(acc2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = acc;
    sh[4 + (threadIdx.x >> 6)] = acc2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    partials[2 * blockIdx.x + 1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

// out[0], out[1] <- the two sums of the n per-block partial pairs
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, int n,
                                                           double* __restrict__ out) {
  __shared__ double sh[8];
  double acc = 0.0, acc2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    acc += partials[2 * i];
    acc2 += partials[2 * i + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_down(acc, o, 64);
    acc2 += __shfl_down(acc2, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = acc;
    sh[4 + (threadIdx.x >> 6)] = acc2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    out[1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

struct Spectral {
  int dim, nx, ny, nz;
  int64_t n, nh;
  FftPlan *fwd = nullptr, *inv = nullptr;  // library transforms (null on the hand-written path)
  unsigned char* block = nullptr;  // one allocation: chat | ghat | scratch
  double2 *chat = nullptr, *ghat = nullptr, *scratch = nullptr;
  double* g = nullptr;
  double* partials = nullptr;  // 2 x 2048 doubles
  double gq = 0.0;             // BM6: k_c^2 / eps (spectral_set_screening)
  bool chat_valid = false;
  Fused2D* fast = nullptr;  // 2-D power-of-two grids: hand-written LDS FFT path (2 launches per step)
  KsArgs ks;
  std::string err;
};

#define SP_HIP(expr)                                                       \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      sp->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
      return -3;                                                           \
    }                                                                      \
  } while (0)
#define SP_FFT(expr)                                                       \
  do {                                                                     \
    if ((expr) != 0) return -3; /* sp->err was filled by the fftplan_* call */ \
  } while (0)

const char* spectral_error(const Spectral* sp) { return sp->err.c_str(); }
// which transforms run (for pf_status_string)
const char* spectral_path(const Spectral* sp) { return sp->fast ? fused2d_describe(sp->fast) : "rocFFT (native API) + pointwise k-space kernels"; }

int spectral_create(Spectral** out, int dim, int nx, int ny, int nz, double h, hipStream_t stream, std::string* err) {
  Spectral* sp = new Spectral();
  *out = sp;
  sp->dim = dim;
  sp->nx = nx;
  sp->ny = ny;
  sp->nz = dim == 3 ? nz : 1;
  sp->n = (int64_t)nx * ny * sp->nz;
  const int nxh = nx / 2 + 1;
  sp->nh = (int64_t)nxh * ny * sp->nz;
  const char* e3 = getenv("PFHIP_SPECTRAL_2D");
  const bool want_fast = fused2d_supported(dim, nx, ny, sp->nz) && !(dim == 2 && e3 && std::string(e3) == "rocfft");
  SpecLayout lay{nxh, ny, (int64_t)ny * sp->nz};
  if (want_fast) lay = fused_spectrum_layout(dim, nx, ny, sp->nz);
  sp->ks.pitch = lay.pitch;
  sp->ks.nyp = lay.nyp;
  const int64_t nh_alloc = (int64_t)lay.pitch * lay.rows;
  sp->ks.nxh = nxh;
  sp->ks.ny = ny;
  sp->ks.nz = sp->nz;
  sp->ks.nx = nx;
  sp->ks.kx0 = TWO_PI / (nx * h);
  sp->ks.ky0 = TWO_PI / (ny * h);
  sp->ks.kz0 = sp->nz > 1 ? TWO_PI / (sp->nz * h) : 0.0;
  sp->ks.inv_n = 1.0 / (double)sp->n;
  auto fail = [&](int rc) {
    if (err) *err = sp->err;
    return rc;
  };
  auto body = [&]() -> int {
    if (!want_fast) {  // the hand-written passes need neither the library plans (and their work buffers) nor sp->g
      const int nn[3] = {nx, ny, sp->nz};
      SP_FFT(fftplan_real(&sp->fwd, dim, nn, 1, true, stream, &sp->err));
      SP_FFT(fftplan_real(&sp->inv, dim, nn, 1, false, stream, &sp->err));
    }
    // The three half-spectrum arrays (resident spectrum, ghat, the inverse transforms' input) come from ONE allocation
    // (separate hipMallocs land wherever the allocator puts them).
    const size_t bytes = sizeof(double2) * nh_alloc, slot = (bytes + 255) / 256 * 256;
    {
      unsigned char* blk = nullptr;
      SP_HIP(pf_malloc(&blk, 3 * slot));
      sp->block = blk;
      sp->chat = reinterpret_cast<double2*>(blk);
      sp->ghat = reinterpret_cast<double2*>(blk + slot);
      sp->scratch = reinterpret_cast<double2*>(blk + 2 * slot);
    }
    if (nh_alloc != sp->nh) {  // padded rows: the pad columns are never written by the passes; keep them defined
      SP_HIP(hipMemsetAsync(sp->chat, 0, sizeof(double2) * nh_alloc, stream));
      SP_HIP(hipMemsetAsync(sp->ghat, 0, sizeof(double2) * nh_alloc, stream));
      SP_HIP(hipMemsetAsync(sp->scratch, 0, sizeof(double2) * nh_alloc, stream));
    }
    if (!want_fast) SP_HIP(pf_malloc(&sp->g, sizeof(double) * sp->n));
    SP_HIP(pf_malloc(&sp->partials, sizeof(double) * 4096));
    if (want_fast) {  // PFHIP_SPECTRAL_2D / _3D = rocfft force the library path (A/B comparison)
      if (fused2d_create(&sp->fast, nx, ny, sp->nz, h, stream) != 0) {
        sp->err = "fused2d_create failed";
        return -3;
      }
    }
    return 0;
  };
  return fail(body());
}

void spectral_destroy(Spectral* sp) {
  if (!sp) return;
  fftplan_destroy(sp->fwd);
  fftplan_destroy(sp->inv);
  if (sp->block) (void)pf_free(sp->block);
  if (sp->g) (void)pf_free(sp->g);
  if (sp->partials) (void)pf_free(sp->partials);
  if (sp->fast) fused2d_destroy(sp->fast);
  delete sp;
}

void spectral_set_screening(Spectral* sp, double gq) { sp->gq = gq; }

void spectral_invalidate(Spectral* sp) {
  sp->chat_valid = false;
  if (sp->fast) fused2d_invalidate(sp->fast);
}

static int ensure_chat(Spectral* sp, const double* c) {
  if (sp->chat_valid) return 0;
  if (sp->fast) {
    if (fused2d_spectrum(sp->fast, c, sp->chat, sp->ghat) != 0) {
      sp->err = "fused2d_spectrum launch failed";
      return -3;
    }
    sp->chat_valid = true;
    return 0;
  }
  SP_FFT(fftplan_exec(sp->fwd, const_cast<double*>(c), sp->chat, &sp->err));
  sp->chat_valid = true;
  return 0;
}

// one semi-implicit step: reads c_in (real space), writes c_out (real space); c_k stays resident
// store_field = false (an intermediate step of a multi-step pf_step call): the hand-written passes do not write the
// real-space field -- the state lives in the resident spectrum and in G, c_out stays untouched; the library path ignores it
int spectral_step(Spectral* sp, const double* c_in, double* c_out, double dt, double M, double kappa, double ca,
                  double cb, double two_rho, hipStream_t stream, bool store_field) {
  int rc = ensure_chat(sp, c_in);
  if (rc) return rc;
  if (sp->fast) {
    if (fused2d_step(sp->fast, c_in, store_field ? c_out : nullptr, sp->chat, sp->ghat, sp->scratch, dt, M, kappa, ca, cb, two_rho,
                     dt * M * sp->gq) != 0) {
      sp->err = "fused2d_step launch failed";
      return -3;
    }
    return 0;
  }
  hipLaunchKernelGGL(dfdc_kernel, dim3(grid_for(sp->n / 2)), dim3(256), 0, stream, c_in, sp->g, sp->n, ca, cb, two_rho);
  SP_FFT(fftplan_exec(sp->fwd, sp->g, sp->ghat, &sp->err));
  KsArgs ks = sp->ks;
  ks.dtM = dt * M;
  ks.dtMkappa = dt * M * kappa;
  ks.gam = dt * M * sp->gq;
  hipLaunchKernelGGL(kspace_update_kernel, dim3(grid_for(sp->nh)), dim3(256), 0, stream, sp->chat,
                     (const double2*)sp->ghat, sp->scratch, sp->nh, ks);
  SP_FFT(fftplan_exec(sp->inv, sp->scratch, c_out, &sp->err));
  SP_HIP(hipGetLastError());
  return 0;
}

// sum_k w_k k^2 |c_k|^2  (= N sum over the lattice of |grad c|^2 by Parseval) -> out_dev[0];
// sum_{k != 0} w_k |c_k|^2 / k^2 -> out_dev[1]
int spectral_grad_energy(Spectral* sp, const double* c, double* out_dev, hipStream_t stream) {
  int rc = ensure_chat(sp, c);
  if (rc) return rc;
  const int nb = grid_for(sp->nh);
  hipLaunchKernelGGL(kspace_grad_energy_kernel, dim3(nb), dim3(256), 0, stream, (const double2*)sp->chat, sp->nh,
                     sp->ks, sp->partials);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const double*)sp->partials, nb, out_dev);
  SP_HIP(hipGetLastError());
  return 0;
}

double spectral_inv_n(const Spectral* sp) { return sp->ks.inv_n; }

}  // namespace pfhip
