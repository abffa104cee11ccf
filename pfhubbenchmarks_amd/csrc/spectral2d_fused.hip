// Fused 2-D spectral step for power-of-two grids (BASELINE.json config 2: 512^2): hand-written LDS FFTs instead of six
// library launches per step.
//
// The rocFFT path (spectral.hip) spends 6 kernels per step on a 2 MiB problem and is launch-latency bound (37-42 us per
// step at 512^2).  Here a step is TWO kernels, each transform living entirely in LDS:
//   column kernel  (per 8 adjacent k_x columns): forward FFT along y of the row-transformed f'(c)  ->  k-space update of
//                  the resident spectrum c_k  ->  inverse FFT along y of c_k / N
//   row kernel     (per pair of rows): inverse real FFT along x  ->  c (stored)  ->  f'(c)  ->  forward real FFT along
//                  x, ready for the next step's column kernel
// Real rows are transformed two at a time as one complex FFT (row a in the real part, row b in the imaginary part) and
// separated by Hermitian symmetry.  FFT = in-place radix-2 decimation in time on bit-reversed input, one wave per
// transform, twiddles from an LDS table.  Same layout / scaling as rocFFT's D2Z output ([ny][nx/2+1], unnormalised
// forward), so diagnostics and the resident spectrum are shared with spectral.hip.
//   c^+_k = (c_k - dt M k^2 N_k) / (1 + dt M kappa k^4),  N = f'(c)       dolfin/pfbase.py:361-383, bench1.py:63-65
#include <cmath>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {

namespace {

constexpr double TWO_PI_F = 6.283185307179586476925286766559;
// columns per workgroup of the column kernel: the 2 MiB problem is L2 resident, so parallelism (129 workgroups at
// 512^2) beats 128-byte coalescing (CW = 8: 33 workgroups, 26 us instead of 14 us)
constexpr int CW = 2;
constexpr int CT = 256;  // threads per column

struct F2Args {
  int nx, ny, nxh, lgx, lgy;
  double ca, cb, two_rho;
  double dtM, dtMkappa, kx0, ky0, inv_n;
};

__device__ __forceinline__ int brev(int i, int lg) { return (int)(__brev((unsigned)i) >> (32 - lg)); }
// LDS index skew: one 16-byte pad slot every 32 elements, so the power-of-two strides of bit-reversed and butterfly
// accesses do not pile onto one bank
__device__ __forceinline__ int px(int j) { return j + (j >> 5); }

// in-place radix-2 DIT FFT of x[0..N) (LDS, bit-reversed input -> natural output) by ONE wave; tw[k] = e^{-2 pi i k/N}.
// SIGN = -1: forward (e^{-i...}); +1: inverse (unnormalised).  Every stage ends with a workgroup barrier (the waves of
// a workgroup run independent transforms in lockstep).
template <int SIGN, int G>  // G = threads cooperating on one transform (t = index inside the group)
__device__ __forceinline__ void fft_inplace(double2* x, const double2* tw, int N, int lgN, int t) {
  // Radix-2 DIT stages merged two at a time (stages s and s+1 on the 4 elements j, j+h, j+2h, j+3h, h = 2^(s-1)):
  // half the LDS round trips and barriers of a plain radix-2 sweep, same data order.  A last single stage if lgN is odd.
  constexpr int MAXQ4 = 256 / G > 0 ? 256 / G : 1;  // 4-element groups per thread: (N/4) / G, N <= 1024
  auto cmul = [](double2 w, double2 v) { return make_double2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x); };
  int s = 1;
  for (; s + 1 <= lgN; s += 2) {
    const int h = 1 << (s - 1);
    double2 e0[MAXQ4], e1[MAXQ4], e2[MAXQ4], e3[MAXQ4], w1[MAXQ4], w2[MAXQ4];
    int jj[MAXQ4];
#pragma unroll
    for (int i = 0; i < MAXQ4; ++i) {
      const int q = t + G * i;
      if (q < N / 4) {
        const int pos = q & (h - 1);
        jj[i] = ((q >> (s - 1)) << (s + 1)) + pos;
        w1[i] = tw[pos * (N >> s)];        // W_{2h}^pos
        w2[i] = tw[pos * (N >> (s + 1))];  // W_{4h}^pos
        e0[i] = x[px(jj[i])];
        e1[i] = x[px(jj[i] + h)];
        e2[i] = x[px(jj[i] + 2 * h)];
        e3[i] = x[px(jj[i] + 3 * h)];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXQ4; ++i) {
      const int q = t + G * i;
      if (q < N / 4) {
        double2 a1 = w1[i], a2 = w2[i];
        if (SIGN > 0) {
          a1.y = -a1.y;
          a2.y = -a2.y;
        }
        // stage s: (e0, e1) and (e2, e3) with W_{2h}^pos
        const double2 t1 = cmul(a1, e1[i]), t3 = cmul(a1, e3[i]);
        const double2 p0 = make_double2(e0[i].x + t1.x, e0[i].y + t1.y);
        const double2 p1 = make_double2(e0[i].x - t1.x, e0[i].y - t1.y);
        const double2 p2 = make_double2(e2[i].x + t3.x, e2[i].y + t3.y);
        const double2 p3 = make_double2(e2[i].x - t3.x, e2[i].y - t3.y);
        // stage s+1: (p0, p2) with W_{4h}^pos, (p1, p3) with W_{4h}^(pos+h) = -+i W_{4h}^pos
        const double2 u2 = cmul(a2, p2);
        const double2 b3 = cmul(a2, p3);
        const double2 u3 = SIGN > 0 ? make_double2(-b3.y, b3.x) : make_double2(b3.y, -b3.x);  // (+i or -i) * b3
        x[px(jj[i])] = make_double2(p0.x + u2.x, p0.y + u2.y);
        x[px(jj[i] + 2 * h)] = make_double2(p0.x - u2.x, p0.y - u2.y);
        x[px(jj[i] + h)] = make_double2(p1.x + u3.x, p1.y + u3.y);
        x[px(jj[i] + 3 * h)] = make_double2(p1.x - u3.x, p1.y - u3.y);
      }
    }
    __syncthreads();
  }
  if (s == lgN) {  // odd number of stages: one plain radix-2 stage
    constexpr int MAXB = 512 / G > 0 ? 512 / G : 1;
    const int half = 1 << (s - 1);
    const int tstep = N >> s;
    double2 w[MAXB], u[MAXB], v[MAXB];
    int jb[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
      const int b = t + G * i;
      if (b < N / 2) {
        const int pos = b & (half - 1);
        jb[i] = ((b >> (s - 1)) << s) + pos;
        w[i] = tw[pos * tstep];
        u[i] = x[px(jb[i])];
        v[i] = x[px(jb[i] + half)];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
      const int b = t + G * i;
      if (b < N / 2) {
        double2 a = w[i];
        if (SIGN > 0) a.y = -a.y;
        const double2 tt = cmul(a, v[i]);
        x[px(jb[i])] = make_double2(u[i].x + tt.x, u[i].y + tt.y);
        x[px(jb[i] + half)] = make_double2(u[i].x - tt.x, u[i].y - tt.y);
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ double fp2(double c, const F2Args& a) {
  const double p = c - a.ca, q = a.cb - c;
  return a.two_rho * ((p * q) * (q - p));
}

// Row kernel: one wave per pair of rows (y0 = 2 blockIdx.x, y0 + 1).
//   from_spectrum: inverse real FFT of H[y][0..nx/2] -> c_out rows;   else rows are read from c_in
//   use_fprime:    the forward transform is applied to f'(c) (a step) or to c itself (spectrum initialisation)
//   G[y][k] <- forward real FFT along x
constexpr int RT = 256;  // threads per row pair
__global__ __launch_bounds__(RT) void f2_row_kernel(const F2Args a, const double2* __restrict__ H,
                                                    const double* __restrict__ c_in, double* __restrict__ c_out,
                                                    double2* __restrict__ G, const double2* __restrict__ twx_g,
                                                    int from_spectrum, int use_fprime) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double2* X = reinterpret_cast<double2*>(smem_raw);
  double2* TW = X + px(a.nx);
  const int lane = threadIdx.x, N = a.nx, lg = a.lgx;
  const int y0 = 2 * blockIdx.x, y1 = y0 + 1;
  for (int k = lane; k < N / 2; k += RT) TW[k] = twx_g[k];
  constexpr int MAXP = 1024 / RT;  // N / RT elements per thread, N <= 1024
  double2 z[MAXP];
  if (from_spectrum) {
    for (int k = lane; k <= N / 2; k += RT) {
      const double2 p = H[(int64_t)y0 * a.nxh + k], q = H[(int64_t)y1 * a.nxh + k];
      X[px(brev(k, lg))] = make_double2(p.x - q.y, p.y + q.x);  // p + i q
      if (k > 0 && k < N / 2) X[px(brev(N - k, lg))] = make_double2(p.x + q.y, q.x - p.y);  // conj(p) + i conj(q)
    }
    __syncthreads();
    fft_inplace<+1, RT>(X, TW, N, lg, lane);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int x = lane + RT * i;
      if (x < N) {
        z[i] = X[px(x)];
        c_out[(int64_t)y0 * N + x] = z[i].x;
        c_out[(int64_t)y1 * N + x] = z[i].y;
      }
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int x = lane + RT * i;
      if (x < N) z[i] = make_double2(c_in[(int64_t)y0 * N + x], c_in[(int64_t)y1 * N + x]);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int x = lane + RT * i;
    if (x < N) X[px(brev(x, lg))] = use_fprime ? make_double2(fp2(z[i].x, a), fp2(z[i].y, a)) : z[i];
  }
  __syncthreads();
  fft_inplace<-1, RT>(X, TW, N, lg, lane);
  for (int k = lane; k <= N / 2; k += RT) {
    const double2 w = X[px(k)], m = X[px((N - k) & (N - 1))];
    G[(int64_t)y0 * a.nxh + k] = make_double2(0.5 * (w.x + m.x), 0.5 * (w.y - m.y));
    G[(int64_t)y1 * a.nxh + k] = make_double2(0.5 * (w.y + m.y), -0.5 * (w.x - m.x));
  }
}

// Column kernel: CW adjacent k_x columns per workgroup, one wave per column.
//   forward FFT along y of G;  init_only: store it as the resident spectrum chat and stop;
//   else chat <- (chat - dtM k^2 Ghat) / (1 + dtM kappa k^4),  H <- inverse FFT along y of chat / N
__global__ __launch_bounds__(CT * CW) void f2_col_kernel(const F2Args a, const double2* __restrict__ G,
                                                         double2* __restrict__ chat, double2* __restrict__ H,
                                                         const double2* __restrict__ twy_g, int init_only) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int NP = px(a.ny) + 1;                         // padded column stride: columns land on different banks
  double2* X = reinterpret_cast<double2*>(smem_raw);  // [CW][ny + 1]
  double2* TW = X + CW * NP;
  const int tid = threadIdx.x, lane = tid % CT, wave = tid / CT;  // (thread in column group, column)
  const int N = a.ny, lg = a.lgy;
  const int kxb = blockIdx.x * CW;
  for (int k = tid; k < N / 2; k += CT * CW) TW[k] = twy_g[k];
  for (int idx = tid; idx < N * CW; idx += CT * CW) {
    const int y = idx / CW, ci = idx % CW, kx = kxb + ci;
    X[ci * NP + px(brev(y, lg))] = kx < a.nxh ? G[(int64_t)y * a.nxh + kx] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  fft_inplace<-1, CT>(X + wave * NP, TW, N, lg, lane);
  constexpr int MAXQ = 1024 / CT;  // N * CW / (CT * CW) = N / CT, N <= 1024
  double2 o[MAXQ];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int idx = tid + CT * CW * i;
    if (idx < N * CW) {
      const int ky = idx / CW, ci = idx % CW, kx = kxb + ci;
      const double2 gh = X[ci * NP + px(ky)];
      o[i] = gh;
      if (kx < a.nxh) {
        if (init_only) {
          chat[(int64_t)ky * a.nxh + kx] = gh;
        } else {
          const int my = 2 * ky > N ? ky - N : ky;
          const double kxv = a.kx0 * kx, kyv = a.ky0 * my;
          const double k2 = (kxv * kxv + kyv * kyv) + 0.0;  // same grouping as spectral.hip's ksq with kz = 0
          const double num = a.dtM * k2;
          const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0);
          const double2 ch = chat[(int64_t)ky * a.nxh + kx];
          double2 r;
          r.x = fma(-num, gh.x, ch.x) * den;
          r.y = fma(-num, gh.y, ch.y) * den;
          chat[(int64_t)ky * a.nxh + kx] = r;
          o[i] = make_double2(r.x * a.inv_n, r.y * a.inv_n);
        }
      }
    }
  }
  if (init_only) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int idx = tid + CT * CW * i;
    if (idx < N * CW) X[(idx % CW) * NP + px(brev(idx / CW, lg))] = o[i];
  }
  __syncthreads();
  fft_inplace<+1, CT>(X + wave * NP, TW, N, lg, lane);
  for (int idx = tid; idx < N * CW; idx += CT * CW) {
    const int y = idx / CW, ci = idx % CW, kx = kxb + ci;
    if (kx < a.nxh) H[(int64_t)y * a.nxh + kx] = X[ci * NP + px(y)];
  }
}

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

}  // namespace

struct Fused2D {
  F2Args a;
  double2 *twx = nullptr, *twy = nullptr;
  size_t lds_row = 0, lds_col = 0;
  hipStream_t stream = nullptr;
  bool g_valid = false;  // G holds the row transform of f'(current c)
};

bool fused2d_supported(int dim, int nx, int ny) {
  const int lx = ilog2(nx), ly = ilog2(ny);
  return dim == 2 && lx >= 7 && lx <= 10 && ly >= 7 && ly <= 10;
}

int fused2d_create(Fused2D** out, int nx, int ny, double h, hipStream_t stream) {
  Fused2D* f = new Fused2D();
  *out = f;
  f->stream = stream;
  F2Args& a = f->a;
  a.nx = nx;
  a.ny = ny;
  a.nxh = nx / 2 + 1;
  a.lgx = ilog2(nx);
  a.lgy = ilog2(ny);
  a.kx0 = TWO_PI_F / (nx * h);
  a.ky0 = TWO_PI_F / (ny * h);
  a.inv_n = 1.0 / ((double)nx * ny);
  auto table = [&](int N, double2** dev) -> hipError_t {
    std::vector<double2> t(N / 2);
    for (int k = 0; k < N / 2; ++k) {
      const double ang = TWO_PI_F * k / N;
      t[k] = make_double2(std::cos(ang), -std::sin(ang));
    }
    hipError_t e = hipMalloc(dev, sizeof(double2) * t.size());
    if (e != hipSuccess) return e;
    return hipMemcpy(*dev, t.data(), sizeof(double2) * t.size(), hipMemcpyHostToDevice);
  };
  if (table(nx, &f->twx) != hipSuccess || table(ny, &f->twy) != hipSuccess) return -3;
  f->lds_row = sizeof(double2) * (nx + nx / 32 + nx / 2);
  f->lds_col = sizeof(double2) * ((size_t)CW * (ny + ny / 32 + 1) + ny / 2);
  if (f->lds_col > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(f2_col_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)f->lds_col) != hipSuccess)
    return -3;
  return 0;
}

void fused2d_destroy(Fused2D* f) {
  if (!f) return;
  if (f->twx) (void)hipFree(f->twx);
  if (f->twy) (void)hipFree(f->twy);
  delete f;
}

void fused2d_invalidate(Fused2D* f) { f->g_valid = false; }

// chat <- 2-D spectrum of c (same as a rocFFT D2Z); G is clobbered
int fused2d_spectrum(Fused2D* f, const double* c, double2* chat, double2* G) {
  const F2Args& a = f->a;
  hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_row, f->stream, a, (const double2*)nullptr, c,
                     (double*)nullptr, G, (const double2*)f->twx, 0, 0);
  hipLaunchKernelGGL(f2_col_kernel, dim3((a.nxh + CW - 1) / CW), dim3(CT * CW), f->lds_col, f->stream, a,
                     (const double2*)G, chat, (double2*)nullptr, (const double2*)f->twy, 1);
  f->g_valid = false;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// one semi-implicit step c_in -> c_out; chat (resident, valid for c_in) is advanced; G, H are work arrays
int fused2d_step(Fused2D* f, const double* c_in, double* c_out, double2* chat, double2* G, double2* H, double dt,
                 double M, double kappa, double ca, double cb, double two_rho) {
  F2Args a = f->a;
  a.ca = ca;
  a.cb = cb;
  a.two_rho = two_rho;
  a.dtM = dt * M;
  a.dtMkappa = dt * M * kappa;
  if (!f->g_valid)  // row transform of f'(c_in) (first step, or after the field was replaced)
    hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_row, f->stream, a, (const double2*)nullptr, c_in,
                       (double*)nullptr, G, (const double2*)f->twx, 0, 1);
  hipLaunchKernelGGL(f2_col_kernel, dim3((a.nxh + CW - 1) / CW), dim3(CT * CW), f->lds_col, f->stream, a,
                     (const double2*)G, chat, H, (const double2*)f->twy, 0);
  hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_row, f->stream, a, (const double2*)H,
                     (const double*)nullptr, c_out, G, (const double2*)f->twx, 1, 1);
  f->g_valid = true;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace pfhip
