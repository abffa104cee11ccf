// Slab-decomposed 3-D FFT building blocks for the multi-GPU spectral / Poisson paths (SURVEY.md section 8e:
// "slab FFT with one all-to-all transpose each way per transform").
//
// A rank owns nzl = nz/P z-planes of a periodic nx x ny x nz box.  Forward transform of a real slab:
//   (1) batched 2-D r2c over (y, x) of the local planes            -> tmp [zl][y][kx]
//   (2) pack by destination rank (y split into P chunks of nyl)    -> A   [q][zl][yq][kx]
//   (3) ALL-TO-ALL (done by the caller over RCCL)                  -> B   [p][zl][yq][kx]  ==  T [z][yq][kx]
//   (4) batched strided 1-D c2c along z                            -> spectrum in the "transposed" layout T
// and the mirror image back.  k-space kernels work on T: global indices kz = z, ky = rank*nyl + yq, kx.
// The library never communicates: pf_dist_advance() runs up to the next exchange and tells the caller which
// buffers to all-to-all (include/pfhip.h: pf_dist_request).
#include "pfhip_internal.h"

namespace pfhip {

namespace {

constexpr double TWO_PI_S = 6.283185307179586476925286766559;

struct SfGeom {
  int nx, ny, nz, nxh, P, rank, nzl, nyl;
  int pitch;  // complex elements per k_x row of every spectrum array: nxh (rocFFT path) or the padded pitch of the
              // hand-written passes (fused_spectrum_pitch); the pad columns are zero and skipped by the k-space kernels
  double h;
};

__global__ __launch_bounds__(256) void sf_pack_kernel(const double2* __restrict__ tmp, double2* __restrict__ A,
                                                      const SfGeom g) {
  const int64_t n = (int64_t)g.nzl * g.ny * g.nxh;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int kx = (int)(i % g.nxh);
    const int y = (int)((i / g.nxh) % g.ny);
    const int zl = (int)(i / ((int64_t)g.nxh * g.ny));
    const int q = y / g.nyl, yq = y % g.nyl;
    A[(((int64_t)q * g.nzl + zl) * g.nyl + yq) * g.nxh + kx] = tmp[i];
  }
}

__global__ __launch_bounds__(256) void sf_unpack_kernel(const double2* __restrict__ A, double2* __restrict__ tmp,
                                                        const SfGeom g) {
  const int64_t n = (int64_t)g.nzl * g.ny * g.nxh;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int kx = (int)(i % g.nxh);
    const int y = (int)((i / g.nxh) % g.ny);
    const int zl = (int)(i / ((int64_t)g.nxh * g.ny));
    const int q = y / g.nyl, yq = y % g.nyl;
    tmp[i] = A[(((int64_t)q * g.nzl + zl) * g.nyl + yq) * g.nxh + kx];
  }
}

__device__ __forceinline__ void t_index(const SfGeom& g, int64_t i, int& kx, int& ky, int& kz) {
  kx = (int)(i % g.pitch);
  const int yq = (int)((i / g.pitch) % g.nyl);
  kz = (int)(i / ((int64_t)g.pitch * g.nyl));
  ky = g.rank * g.nyl + yq;
}

// Poisson on T: B <- -(k/eps) B / lambda_h / N   (5/7-point eigenvalues; zero mode dropped)
__global__ __launch_bounds__(256) void sf_poisson_kernel(double2* __restrict__ B, const SfGeom g, double k_over_eps,
                                                         double inv_n) {
  const int64_t n = (int64_t)g.nz * g.nyl * g.pitch;
  const double inv_h2 = 1.0 / (g.h * g.h);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int kx, ky, kz;
    t_index(g, i, kx, ky, kz);
    if (kx >= g.nxh) continue;
    double lam = (2.0 * cos(TWO_PI_S * kx / g.nx) - 2.0) + (2.0 * cos(TWO_PI_S * ky / g.ny) - 2.0);
    lam += 2.0 * cos(TWO_PI_S * kz / g.nz) - 2.0;
    lam *= inv_h2;
    const double s = (kx == 0 && ky == 0 && kz == 0) ? 0.0 : -k_over_eps * inv_n / lam;
    const double2 v = B[i];
    B[i] = make_double2(v.x * s, v.y * s);
  }
}

__device__ __forceinline__ double t_ksq(const SfGeom& g, int64_t i) {
  int kx, ky, kz;
  t_index(g, i, kx, ky, kz);
  if (2 * ky > g.ny) ky -= g.ny;
  if (2 * kz > g.nz) kz -= g.nz;
  const double a = TWO_PI_S / (g.nx * g.h) * kx, b = TWO_PI_S / (g.ny * g.h) * ky, c = TWO_PI_S / (g.nz * g.h) * kz;
  return (a * a + b * b) + c * c;
}

// spectral CH on T: chat <- (chat - dtM k^2 B) / (1 + dtM kappa k^4);  B <- chat / N   (B holds ghat on entry)
__global__ __launch_bounds__(256) void sf_spectral_update_kernel(double2* __restrict__ chat, double2* __restrict__ B,
                                                                 const SfGeom g, double dtM, double dtMkappa,
                                                                 double inv_n) {
  const int64_t n = (int64_t)g.nz * g.nyl * g.pitch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if ((int)(i % g.pitch) >= g.nxh) continue;
    const double k2 = t_ksq(g, i);
    const double num = dtM * k2;
    const double den = 1.0 / fma(dtMkappa, k2 * k2, 1.0);
    const double2 ch = chat[i], gh = B[i];
    double2 o;
    o.x = fma(-num, gh.x, ch.x) * den;
    o.y = fma(-num, gh.y, ch.y) * den;
    chat[i] = o;
    B[i] = make_double2(o.x * inv_n, o.y * inv_n);
  }
}

// local part of sum_k w_k k^2 |chat_k|^2  -> partials (one per block)
__global__ __launch_bounds__(256) void sf_grad_energy_kernel(const double2* __restrict__ chat, const SfGeom g,
                                                             double* __restrict__ partials) {
  __shared__ double sh[4];
  const int64_t n = (int64_t)g.nz * g.nyl * g.pitch;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int kx = (int)(i % g.pitch);
    if (kx >= g.nxh) continue;
    const double w = (kx == 0 || 2 * kx == g.nx) ? 1.0 : 2.0;
    const double2 c = chat[i];
    acc += w * t_ksq(g, i) * (c.x * c.x + c.y * c.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sf_sum_kernel(const double* __restrict__ partials, int n, double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sf_dfdc_kernel(const double* __restrict__ c, double* __restrict__ g, int64_t n,
                                                      double ca, double cb, double two_rho) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double w = c[i];
    const double a = w - ca, b = cb - w;
    g[i] = two_rho * ((a * b) * (b - a));
  }
}

int sf_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

struct SlabFFT {
  SfGeom g;
  int64_t blk;  // complex elements per peer block
  FftPlan *p2f = nullptr, *p2i = nullptr, *pzf = nullptr, *pzi = nullptr;  // library transforms (rocFFT, native API)
  Fused2D* fast = nullptr;  // power-of-two boxes: the hand-written LDS-FFT passes instead of rocFFT + pack / unpack
  double2 *tmp = nullptr, *A = nullptr, *B = nullptr, *chat = nullptr;
  bool own_ab = false;
  double *greal = nullptr, *partials = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
};

#define SF_HIP(expr)                                                       \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      sf->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
      return -3;                                                           \
    }                                                                      \
  } while (0)
#define SF_FFT(expr)                                                       \
  do {                                                                     \
    if ((expr) != 0) return -3; /* sf->err was filled by the fftplan_* call */ \
  } while (0)

const char* slabfft_error(const SlabFFT* sf) { return sf->err.c_str(); }
// which transforms this rank runs between the all-to-alls (for pf_status_string)
const char* slabfft_path(const SlabFFT* sf) {
  return sf->fast ? fused2d_describe(sf->fast)
                  : "rocFFT (native API): batched 2-D r2c / c2r on the local planes, strided 1-D c2c along z, pack / unpack kernels "
                    "(the hand-written passes take power-of-two boxes of 128..1024 points per axis)";
}
int64_t slabfft_doubles_per_peer(const SlabFFT* sf) { return 2 * sf->blk; }
double* slabfft_buf(const SlabFFT* sf, int which) { return reinterpret_cast<double*>(which == 0 ? sf->A : sf->B); }

// doubles each of the two all-to-all buffers must hold (pure host arithmetic; <0 if the box does not divide)
int64_t slabfft_buffer_doubles(int nx, int ny, int nz, int P) {
  if (P < 1 || ny % P || nz % P) return -1;
  const int pitch = fusedslab_supported(nx, ny, nz, P) ? fused_spectrum_pitch(3, nx, ny, nz) : nx / 2 + 1;
  return 2 * (int64_t)(nz / P) * ny * pitch;
}

int slabfft_create(SlabFFT** out, int nx, int ny, int nz, int P, int rank, double h, bool with_spectral, double* extA,
                   double* extB, hipStream_t stream, std::string* err) {
  SlabFFT* sf = new SlabFFT();
  *out = sf;
  sf->stream = stream;
  SfGeom& g = sf->g;
  g.nx = nx;
  g.ny = ny;
  g.nz = nz;
  g.nxh = nx / 2 + 1;
  g.P = P;
  g.rank = rank;
  g.h = h;
  auto body = [&]() -> int {
    if (ny % P || nz % P) {
      sf->err = "slab FFT needs ny and nz divisible by the number of ranks";
      return -2;
    }
    g.nzl = nz / P;
    g.nyl = ny / P;
    const bool fast = fusedslab_supported(nx, ny, nz, P);
    g.pitch = fast ? fused_spectrum_pitch(3, nx, ny, nz) : g.nxh;
    sf->blk = (int64_t)g.nzl * g.nyl * g.pitch;
    if (fast) {
      if (fused2d_create(&sf->fast, nx, ny, nz, h, stream) != 0) {
        sf->err = "fused2d_create failed";
        return -3;
      }
    } else {
      const int n2[2] = {nx, ny};   // batched 2-D r2c / c2r over the local planes
      SF_FFT(fftplan_real(&sf->p2f, 2, n2, g.nzl, true, stream, &sf->err));
      SF_FFT(fftplan_real(&sf->p2i, 2, n2, g.nzl, false, stream, &sf->err));
      const int stride = g.nyl * g.nxh;   // 1-D c2c along z on the transposed layout T[z][yq][kx], in place
      SF_FFT(fftplan_c2c_strided(&sf->pzf, nz, stride, 1, stride, stream, true, &sf->err));
      SF_FFT(fftplan_c2c_strided(&sf->pzi, nz, stride, 1, stride, stream, false, &sf->err));
    }
    const size_t loc = sizeof(double2) * (size_t)g.nzl * ny * g.pitch;
    SF_HIP(pf_malloc(&sf->tmp, loc));
    if (extA && extB) {
      sf->A = reinterpret_cast<double2*>(extA);
      sf->B = reinterpret_cast<double2*>(extB);
    } else {
      sf->own_ab = true;
      SF_HIP(pf_malloc(&sf->A, loc));
      SF_HIP(pf_malloc(&sf->B, loc));
    }
    SF_HIP(pf_malloc(&sf->partials, sizeof(double) * 2049));
    if (with_spectral) {
      SF_HIP(pf_malloc(&sf->chat, loc));
      if (!fast) SF_HIP(pf_malloc(&sf->greal, sizeof(double) * (size_t)g.nzl * ny * nx));
    }
    if (fast) {  // the passes never write the pad columns: keep them zero in every array they travel through
      SF_HIP(hipMemsetAsync(sf->tmp, 0, loc, stream));
      SF_HIP(hipMemsetAsync(sf->A, 0, loc, stream));
      SF_HIP(hipMemsetAsync(sf->B, 0, loc, stream));
      if (sf->chat) SF_HIP(hipMemsetAsync(sf->chat, 0, loc, stream));
    }
    return 0;
  };
  int rc = body();
  if (rc && err) *err = sf->err;
  return rc;
}

void slabfft_destroy(SlabFFT* sf) {
  if (!sf) return;
  fftplan_destroy(sf->p2f);
  fftplan_destroy(sf->p2i);
  fftplan_destroy(sf->pzf);
  fftplan_destroy(sf->pzi);
  if (sf->fast) fused2d_destroy(sf->fast);
  if (sf->tmp) (void)pf_free(sf->tmp);
  if (sf->own_ab) {
    if (sf->A) (void)pf_free(sf->A);
    if (sf->B) (void)pf_free(sf->B);
  }
  if (sf->chat) (void)pf_free(sf->chat);
  if (sf->greal) (void)pf_free(sf->greal);
  if (sf->partials) (void)pf_free(sf->partials);
  delete sf;
}

// real slab (nzl contiguous planes) -> A, ready for the all-to-all A -> B
int slabfft_forward_local(SlabFFT* sf, const double* real_in) {
  if (sf->fast) {
    if (fusedslab_forward_xy(sf->fast, real_in, sf->tmp, sf->A, sf->g.nzl, sf->g.P, 0, 0.0, 0.0, 0.0) != 0) {
      sf->err = "fusedslab_forward_xy launch failed";
      return -3;
    }
    return 0;
  }
  const int64_t n = (int64_t)sf->g.nzl * sf->g.ny * sf->g.nxh;
  SF_FFT(fftplan_exec(sf->p2f, const_cast<double*>(real_in), sf->tmp, &sf->err));
  hipLaunchKernelGGL(sf_pack_kernel, dim3(sf_grid(n)), dim3(256), 0, sf->stream, (const double2*)sf->tmp, sf->A, sf->g);
  SF_HIP(hipGetLastError());
  return 0;
}
// same for f'(c) of the slab (spectral scheme)
int slabfft_forward_local_dfdc(SlabFFT* sf, const double* c, double ca, double cb, double two_rho) {
  if (sf->fast) {  // f'(c) is evaluated inside the row kernel
    if (fusedslab_forward_xy(sf->fast, c, sf->tmp, sf->A, sf->g.nzl, sf->g.P, 1, ca, cb, two_rho) != 0) {
      sf->err = "fusedslab_forward_xy launch failed";
      return -3;
    }
    return 0;
  }
  const int64_t n = (int64_t)sf->g.nzl * sf->g.ny * sf->g.nx;
  hipLaunchKernelGGL(sf_dfdc_kernel, dim3(sf_grid(n)), dim3(256), 0, sf->stream, c, sf->greal, n, ca, cb, two_rho);
  return slabfft_forward_local(sf, sf->greal);
}
int slabfft_z(SlabFFT* sf, int inverse) {
  if (sf->fast) {
    if (fusedslab_z(sf->fast, sf->B, nullptr, inverse ? 1 : 0, sf->g.nyl, sf->g.rank * sf->g.nyl, 0.0, 0.0) != 0) {
      sf->err = "fusedslab_z launch failed";
      return -3;
    }
    return 0;
  }
  SF_FFT(fftplan_exec(inverse ? sf->pzi : sf->pzf, sf->B, nullptr, &sf->err));
  return 0;
}
// A (after the all-to-all B -> A) -> real slab
int slabfft_inverse_local(SlabFFT* sf, double* real_out) {
  if (sf->fast) {
    if (fusedslab_inverse_yx(sf->fast, sf->A, sf->tmp, real_out, sf->g.nzl, sf->g.P) != 0) {
      sf->err = "fusedslab_inverse_yx launch failed";
      return -3;
    }
    return 0;
  }
  const int64_t n = (int64_t)sf->g.nzl * sf->g.ny * sf->g.nxh;
  hipLaunchKernelGGL(sf_unpack_kernel, dim3(sf_grid(n)), dim3(256), 0, sf->stream, (const double2*)sf->A, sf->tmp, sf->g);
  SF_FFT(fftplan_exec(sf->p2i, sf->tmp, real_out, &sf->err));
  SF_HIP(hipGetLastError());
  return 0;
}
int slabfft_poisson_on_T(SlabFFT* sf, double k_over_eps) {
  const int64_t n = (int64_t)sf->g.nz * sf->g.nyl * sf->g.pitch;
  const double inv_n = 1.0 / ((double)sf->g.nx * sf->g.ny * sf->g.nz);
  hipLaunchKernelGGL(sf_poisson_kernel, dim3(sf_grid(n)), dim3(256), 0, sf->stream, sf->B, sf->g, k_over_eps, inv_n);
  SF_HIP(hipGetLastError());
  return 0;
}
int slabfft_store_chat(SlabFFT* sf) {  // B (spectrum of c on T) -> resident chat
  SF_HIP(hipMemcpyAsync(sf->chat, sf->B, sizeof(double2) * (size_t)sf->g.nz * sf->g.nyl * sf->g.pitch,
                        hipMemcpyDeviceToDevice, sf->stream));
  return 0;
}
int slabfft_spectral_update_on_T(SlabFFT* sf, double dtM, double dtMkappa) {
  const int64_t n = (int64_t)sf->g.nz * sf->g.nyl * sf->g.pitch;
  const double inv_n = 1.0 / ((double)sf->g.nx * sf->g.ny * sf->g.nz);
  hipLaunchKernelGGL(sf_spectral_update_kernel, dim3(sf_grid(n)), dim3(256), 0, sf->stream, sf->chat, sf->B, sf->g, dtM,
                     dtMkappa, inv_n);
  SF_HIP(hipGetLastError());
  return 0;
}
// forward z transform of B (ghat on T) -> k-space update of the resident chat -> inverse z transform of chat / N in B
int slabfft_z_update(SlabFFT* sf, double dtM, double dtMkappa) {
  if (sf->fast) {  // one pass: the z columns stay in LDS between the two transforms
    if (fusedslab_z(sf->fast, sf->B, sf->chat, 2, sf->g.nyl, sf->g.rank * sf->g.nyl, dtM, dtMkappa) != 0) {
      sf->err = "fusedslab_z launch failed";
      return -3;
    }
    return 0;
  }
  int rc = slabfft_z(sf, 0);
  if (rc == 0) rc = slabfft_spectral_update_on_T(sf, dtM, dtMkappa);
  if (rc == 0) rc = slabfft_z(sf, 1);
  return rc;
}
// this rank's share of sum_k w_k k^2 |chat_k|^2 -> out_dev[0]
int slabfft_grad_energy_local(SlabFFT* sf, double* out_dev) {
  const int64_t n = (int64_t)sf->g.nz * sf->g.nyl * sf->g.pitch;
  const int nb = sf_grid(n);
  hipLaunchKernelGGL(sf_grad_energy_kernel, dim3(nb), dim3(256), 0, sf->stream, (const double2*)sf->chat, sf->g,
                     sf->partials);
  hipLaunchKernelGGL(sf_sum_kernel, dim3(1), dim3(256), 0, sf->stream, (const double*)sf->partials, nb, out_dev);
  SF_HIP(hipGetLastError());
  return 0;
}

}  // namespace pfhip
