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
//
// The same transforms serve 3-D boxes whose axes are powers of two in 128..1024 (four passes per step: z, y, x, y;
// 512-point axes by the one-wave radix-8 kernels, the others by the radix-2^2 kernels), the 3-D periodic Poisson solve of
// BM6, and the slab-decomposed transforms of the multi-GPU modes (fusedslab_*, used by slabfft.hip).
#include <cmath>
#include <cstdlib>
#include <string>
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
  int pitch;  // complex elements per k_x row of the half spectrum in memory: nxh (rocFFT's D2Z layout) in 2-D; rounded up
              // to a multiple of 8 (one 128-byte line per 8 columns) on the 512^3 path -- see fused_spectrum_pitch()
  double ca, cb, two_rho;
  double dtM, dtMkappa, kx0, ky0, inv_n;
  int nz = 1;        // 3-D path (512^3) only
  double kz0 = 0.0;
  double gam = 0.0;  // BM6: dt M k_c^2 / eps added to the implicit denominator for k != 0
  int yoff = 0;      // slab-decomposed z pass: global k_y index of the first local y-row
};

// Column passes of the slab-decomposed transforms (csrc/slabfft.hip) move their columns between the natural layout
// [batch][row][kx] and the all-to-all layout [row / split][batch][row % split][kx] on the fly: element r of the column of
// batch b sits at (r >> lg) * chunk + b * bstride + (r & (2^lg - 1)) * col_stride + kx.  on = 1: the store side of a
// MODE 0 pass (written to H instead of in place); on = 2: the load side of a MODE 1 pass (read from A, written to H).
struct ColSplit {
  int on = 0, lg = 0;
  int64_t chunk = 0, bstride = 0;
};
__device__ __forceinline__ int64_t split_addr(const ColSplit& sp, int b, int kx, int r, int64_t col_stride) {
  return (int64_t)(r >> sp.lg) * sp.chunk + (int64_t)b * sp.bstride + (int64_t)(r & ((1 << sp.lg) - 1)) * col_stride + kx;
}

__device__ __forceinline__ int brev(int i, int lg) { return (int)(__brev((unsigned)i) >> (32 - lg)); }
// LDS index skew: one 16-byte pad slot every 32 elements, so the power-of-two strides of bit-reversed and butterfly
// accesses do not pile onto one bank
__device__ __forceinline__ int px(int j) { return j + (j >> 5); }

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The column kernels give every XCD one
// contiguous band of column blocks, so that the workgroups sharing a 128-byte line of G / chat / H (8 columns) meet in
// the same L2 instead of pulling the line into up to four of them (17.9 -> 13.9 us per 512^2 step).  Bijection on
// [0, gridDim.x) for any grid size.
__device__ __forceinline__ int xcd_band_block() {
  const int nb = gridDim.x, per = (nb + 7) / 8, full = nb % 8;
  const int x = blockIdx.x % 8, r = blockIdx.x / 8;
  if (full == 0) return x * per + r;
  return (x < full ? x * per : full * per + (x - full) * (per - 1)) + r;  // `full` XCDs own per, the rest per - 1
}

// in-place radix-2 DIT FFT of x[0..N) (LDS, bit-reversed input -> natural output) by ONE wave; tw[k] = e^{-2 pi i k/N}.
// SIGN = -1: forward (e^{-i...}); +1: inverse (unnormalised).  Every stage ends with a workgroup barrier (the waves of
// a workgroup run independent transforms in lockstep).
template <int SIGN, int G, int NMAX = 1024>  // G = threads cooperating on one transform (t = index inside the group)
__device__ __forceinline__ void fft_inplace(double2* x, const double2* tw, int N, int lgN, int t) {
  // Radix-2 DIT stages merged two at a time (stages s and s+1 on the 4 elements j, j+h, j+2h, j+3h, h = 2^(s-1)):
  // half the LDS round trips and barriers of a plain radix-2 sweep, same data order.  A last single stage if lgN is odd.
  constexpr int MAXQ4 = NMAX / 4 / G > 0 ? NMAX / 4 / G : 1;  // 4-element groups per thread: (N/4) / G, N <= NMAX
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
    constexpr int MAXB = NMAX / 2 / G > 0 ? NMAX / 2 / G : 1;
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
      const double2 p = H[(int64_t)y0 * a.pitch + k], q = H[(int64_t)y1 * a.pitch + k];
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
        if (c_out) {  // null: an intermediate step of a multi-step call, the field is not materialised
          c_out[(int64_t)y0 * N + x] = z[i].x;
          c_out[(int64_t)y1 * N + x] = z[i].y;
        }
      }
    }
    if (from_spectrum == 2) return;  // inverse only (Poisson solve)
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
    G[(int64_t)y0 * a.pitch + k] = make_double2(0.5 * (w.x + m.x), 0.5 * (w.y - m.y));
    G[(int64_t)y1 * a.pitch + k] = make_double2(0.5 * (w.y + m.y), -0.5 * (w.x - m.x));
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
  const int kxb = xcd_band_block() * CW;
  for (int k = tid; k < N / 2; k += CT * CW) TW[k] = twy_g[k];
  for (int idx = tid; idx < N * CW; idx += CT * CW) {
    const int y = idx / CW, ci = idx % CW, kx = kxb + ci;
    X[ci * NP + px(brev(y, lg))] = kx < a.nxh ? G[(int64_t)y * a.pitch + kx] : make_double2(0.0, 0.0);
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
          chat[(int64_t)ky * a.pitch + kx] = gh;
        } else {
          const int my = 2 * ky > N ? ky - N : ky;
          const double kxv = a.kx0 * kx, kyv = a.ky0 * my;
          const double k2 = (kxv * kxv + kyv * kyv) + 0.0;  // same grouping as spectral.hip's ksq with kz = 0
          const double num = a.dtM * k2;
          const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
          const double2 ch = chat[(int64_t)ky * a.pitch + kx];
          double2 r;
          r.x = fma(-num, gh.x, ch.x) * den;
          r.y = fma(-num, gh.y, ch.y) * den;
          chat[(int64_t)ky * a.pitch + kx] = r;
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
    if (kx < a.nxh) H[(int64_t)y * a.pitch + kx] = X[ci * NP + px(y)];
  }
}

// =====================================================================================================================
// 512-point fast path: ONE wave per transform, 8 points per lane, three radix-8 passes in registers (512 = 8 * 8 * 8)
// with two wave-private LDS exchanges in between -- no bit reversal, no multi-wave barriers, 2 instead of 9 LDS round
// trips per transform.
//   n = l + 64 j          stage A: lane holds x[l + 64 j], j = 0..7 -> radix-8 over j -> y[l][q], times W_512^(l q)
//   l = l0 + 8 l1         stage B: lane (q, l0) gathers l1 = 0..7 -> radix-8 -> z[q][l0][s], times W_64^(l0 s)
//   k = q + 8 s + 64 t    stage C: lane (q, s) gathers l0 = 0..7 -> radix-8 -> X[q + 8 s + 64 t], t = 0..7
// Physical lane p = 8 q + s ends up holding X[T(p) + 64 t] with T(p) = q + 8 s (the two octal digits of p swapped).
// The input side accepts any lane -> l map m(p) (the twiddles W_512^(m q) come from a table row), so a second transform
// can consume the first one's output in place with m = T: forward and inverse chain without touching LDS in between.
// LDS layout of the exchanges: 8 blocks of 72 slots (64 + 8 pad); exchange 1 at [q*72 + l], exchange 2 at
// [q*72 + 9*l0 + s]: both directions of both exchanges are bank-conflict free per quarter-wave of 16-byte accesses.
constexpr int W8 = 576;  // double2 slots of LDS per wave (8 * 72)
constexpr double RSQRT2 = 0.70710678118654752440084436210485;

__device__ __forceinline__ double2 cmul2(double2 w, double2 v) {
  return make_double2(w.x * v.x - w.y * v.y, w.x * v.y + w.y * v.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (SIGN < 0, forward) or +i (inverse)
template <int SIGN>
__device__ __forceinline__ double2 rot90(double2 a) {
  return SIGN < 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
}

// a[q] <- sum_j a[j] e^{SIGN 2 pi i j q / 8}
template <int SIGN>
__device__ __forceinline__ void radix8(double2 (&a)[8]) {
  const double2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
  const double2 b1 = cadd(a[1], a[5]), d5 = csub(a[1], a[5]);
  const double2 b2 = cadd(a[2], a[6]), d6 = csub(a[2], a[6]);
  const double2 b3 = cadd(a[3], a[7]), d7 = csub(a[3], a[7]);
  // odd branch inputs times W_8^j: W_8 = (1 + SIGN i) / sqrt 2, W_8^2 = SIGN i, W_8^3 = (-1 + SIGN i) / sqrt 2
  const double2 r5 = rot90<SIGN>(d5), r7 = rot90<SIGN>(d7);
  const double2 b5 = make_double2((d5.x + r5.x) * RSQRT2, (d5.y + r5.y) * RSQRT2);
  const double2 b6 = rot90<SIGN>(d6);
  const double2 b7 = make_double2((r7.x - d7.x) * RSQRT2, (r7.y - d7.y) * RSQRT2);
  // two 4-point transforms
  {
    const double2 e0 = cadd(b0, b2), e1 = csub(b0, b2), e2 = cadd(b1, b3), e3 = rot90<SIGN>(csub(b1, b3));
    a[0] = cadd(e0, e2);
    a[4] = csub(e0, e2);
    a[2] = cadd(e1, e3);
    a[6] = csub(e1, e3);
  }
  {
    const double2 e0 = cadd(b4, b6), e1 = csub(b4, b6), e2 = cadd(b5, b7), e3 = rot90<SIGN>(csub(b5, b7));
    a[1] = cadd(e0, e2);
    a[5] = csub(e0, e2);
    a[3] = cadd(e1, e3);
    a[7] = csub(e1, e3);
  }
}

// v[j] = x[m + 64 j] on entry (m = this lane's input index, any bijection of the lanes), v[t] = X[T(lane) + 64 t] on
// exit.  twA[q-1] = e^{-2 pi i m q / 512}, twB[s-1] = e^{-2 pi i (lane & 7) s / 64} (conjugated here for SIGN > 0).
// L: this wave's 576-slot LDS region (private to the wave: the exchanges need wave-level ordering only).
// Ordering point between two phases of LDS traffic that stay inside ONE wave's region: the LDS unit executes a wave's
// instructions in issue order, so a read issued after a write of the same wave sees it -- no s_barrier is needed
// (measured: no slower and no faster than workgroup barriers here, 3.17 vs 3.18 ms per 512^3 step on the same box).  The fence + wave barrier only stop the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int SIGN>
__device__ __forceinline__ void fft512_wave(double2 (&v)[8], double2* L, int m, const double2 (&twA)[7],
                                            const double2 (&twB)[7], int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  radix8<SIGN>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    double2 w = twA[q - 1];
    if (SIGN > 0) w.y = -w.y;
    v[q] = cmul2(w, v[q]);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) L[q * 72 + m] = v[q];
  wave_lds_sync();
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) v[l1] = L[hi * 72 + lo + 8 * l1];
  radix8<SIGN>(v);
#pragma unroll
  for (int s = 1; s < 8; ++s) {
    double2 w = twB[s - 1];
    if (SIGN > 0) w.y = -w.y;
    v[s] = cmul2(w, v[s]);
  }
  wave_lds_sync();
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx) L[hi * 72 + 9 * lo + sidx] = v[sidx];
  wave_lds_sync();
#pragma unroll
  for (int l0 = 0; l0 < 8; ++l0) v[l0] = L[hi * 72 + 9 * l0 + lo];
  radix8<SIGN>(v);
}

// Same transform with the twiddles fetched from the (L1/L2-resident, 9 KB) tables right before each use instead of held
// in 56 VGPRs for the whole kernel: the column passes then fit 128 VGPRs with the resident spectrum in flight, i.e. two
// 8-wave workgroups per CU instead of one (f3_col512_kernel).  rowA = this lane's input index m, rowB = lane & 7.
template <int SIGN>
__device__ __forceinline__ void fft512_wave_tw(double2 (&v)[8], double2* L, int m, const double2* __restrict__ twA_g,
                                               const double2* __restrict__ twB_g, int lane) {
  const int hi = lane >> 3, lo = lane & 7;
  {
    double2 tw[7];
#pragma unroll
    for (int q = 1; q < 8; ++q) tw[q - 1] = twA_g[m * 8 + q];
    radix8<SIGN>(v);
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      double2 w = tw[q - 1];
      if (SIGN > 0) w.y = -w.y;
      v[q] = cmul2(w, v[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) L[q * 72 + m] = v[q];
  wave_lds_sync();
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) v[l1] = L[hi * 72 + lo + 8 * l1];
  {
    double2 tw[7];
#pragma unroll
    for (int sx = 1; sx < 8; ++sx) tw[sx - 1] = twB_g[lo * 8 + sx];
    radix8<SIGN>(v);
#pragma unroll
    for (int sx = 1; sx < 8; ++sx) {
      double2 w = tw[sx - 1];
      if (SIGN > 0) w.y = -w.y;
      v[sx] = cmul2(w, v[sx]);
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx) L[hi * 72 + 9 * lo + sidx] = v[sidx];
  wave_lds_sync();
#pragma unroll
  for (int l0 = 0; l0 < 8; ++l0) v[l0] = L[hi * 72 + 9 * l0 + lo];
  radix8<SIGN>(v);
}

__device__ __forceinline__ void load_tw(double2 (&tw)[7], const double2* __restrict__ table, int row) {
#pragma unroll
  for (int q = 1; q < 8; ++q) tw[q - 1] = table[row * 8 + q];
}

// Row kernel, nx == 512: one wave per pair of rows; same contract as f2_row_kernel.
constexpr int RW = 1;  // row pairs (waves) per workgroup
// LAZY: twiddles fetched right before each use (fft512_wave_tw) instead of held in 84 VGPRs -- the 512^3 pass streams
// from HBM and wants occupancy; the latency-bound 2-D step keeps them resident (loaded beside the data at kernel entry).
template <bool LAZY>
__global__ __launch_bounds__(64 * RW) void f2_row512_kernel(const F2Args a, const double2* __restrict__ H,
                                                            const double* __restrict__ c_in,
                                                            double* __restrict__ c_out, double2* __restrict__ G,
                                                            const double2* __restrict__ twA_g,
                                                            const double2* __restrict__ twB_g, int from_spectrum,
                                                            int use_fprime) {
  __shared__ __attribute__((aligned(16))) double2 Lall[RW * W8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double2* L = Lall + wave * W8;
  constexpr int N = 512;
  const int pair = blockIdx.x * RW + wave;
  const int y0 = 2 * pair, y1 = y0 + 1;  // ny is even and (ny / 2) % RW == 0 (checked by the launcher)
  const int T = (lane >> 3) + 8 * (lane & 7);
  double2 twN[7], twB[7], v[8];
  if (!LAZY) {
    load_tw(twN, twA_g, lane);
    load_tw(twB, twB_g, lane & 7);
  }
  int m = lane;
  if (from_spectrum) {
    double2 twT[7];
    if (!LAZY) load_tw(twT, twA_g, T);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = lane + 64 * j;
      const bool upper = k > N / 2;
      const int kk = upper ? N - k : k;
      const double2 p = H[(int64_t)y0 * a.pitch + kk], q = H[(int64_t)y1 * a.pitch + kk];
      // X[k] = p + i q for k <= N/2, conj(p) + i conj(q) beyond (Hermitian rows)
      v[j] = upper ? make_double2(p.x + q.y, q.x - p.y) : make_double2(p.x - q.y, p.y + q.x);
    }
    if (LAZY)
      fft512_wave_tw<+1>(v, L, lane, twA_g, twB_g, lane);
    else
      fft512_wave<+1>(v, L, lane, twN, twB, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (c_out) {  // null: an intermediate step of a multi-step call, the field is not materialised
        c_out[(int64_t)y0 * N + T + 64 * t] = v[t].x;
        c_out[(int64_t)y1 * N + T + 64 * t] = v[t].y;
      }
    }
    if (from_spectrum == 2) return;  // inverse only (Poisson solve): no forward transform of the result
    m = T;
    if (!LAZY) {
#pragma unroll
      for (int q = 0; q < 7; ++q) twN[q] = twT[q];
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = make_double2(c_in[(int64_t)y0 * N + lane + 64 * j], c_in[(int64_t)y1 * N + lane + 64 * j]);
  }
  if (use_fprime) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = make_double2(fp2(v[j].x, a), fp2(v[j].y, a));
  }
  if (LAZY)
    fft512_wave_tw<-1>(v, L, m, twA_g, twB_g, lane);
  else
    fft512_wave<-1>(v, L, m, twN, twB, lane);
  // Hermitian separation of the two real rows: needs X[k] and X[N - k] -> one more exchange (skewed: k + k / 8)
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = T + 64 * t;
    L[k + (k >> 3)] = v[t];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int k = lane + 64 * t;
    if (k <= N / 2) {
      const int km = (N - k) & (N - 1);
      const double2 w = L[k + (k >> 3)], mm = L[km + (km >> 3)];
      G[(int64_t)y0 * a.pitch + k] = make_double2(0.5 * (w.x + mm.x), 0.5 * (w.y - mm.y));
      G[(int64_t)y1 * a.pitch + k] = make_double2(0.5 * (w.y + mm.y), -0.5 * (w.x - mm.x));
    }
  }
}

constexpr int W8C = W8 + 32;  // per-wave LDS region of the staged column kernels: FFT exchanges (576) or a skewed
                              // natural-order column (575) + 4 * column index

// Column kernel, ny == 512 (2-D): one wave per k_x column, contract of f2_col_kernel.  No LDS staging: every wave reads /
// writes its own column straight from global memory (16-byte accesses, one cache line per lane) and relies on the XCD
// band to find the line in its L2 after a sibling fetched it.  A form that staged every access through LDS with the
// column index fastest (full 64-byte sectors) was measured once the band was in place and is slower, 13.9 vs 13.05 us
// per step (extra LDS round trips and workgroup barriers on the critical path of a latency-bound kernel); the 512^3
// passes below, which stream from HBM, do stage (f3_col512_kernel).
template <int CWN>
__global__ __launch_bounds__(64 * CWN) void f2_col512_direct_kernel(const F2Args a, const double2* __restrict__ G,
                                                                    double2* __restrict__ chat,
                                                                    double2* __restrict__ H,
                                                                    const double2* __restrict__ twA_g,
                                                                    const double2* __restrict__ twB_g, int init_only) {
  __shared__ __attribute__((aligned(16))) double2 Lall[CWN * W8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double2* L = Lall + wave * W8;
  constexpr int N = 512;
  const int kx = xcd_band_block() * CWN + wave;
  const bool on = kx < a.nxh;
  const int kxc = on ? kx : a.nxh - 1;  // idle waves of the last workgroup shadow a valid column (no stores)
  const int T = (lane >> 3) + 8 * (lane & 7);
  double2 twN[7], twT[7], twB[7], v[8], ch[8];
  load_tw(twN, twA_g, lane);
  load_tw(twT, twA_g, T);
  load_tw(twB, twB_g, lane & 7);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = G[(int64_t)(lane + 64 * j) * a.pitch + kxc];
  if (!init_only) {
#pragma unroll
    for (int t = 0; t < 8; ++t) ch[t] = chat[(int64_t)(T + 64 * t) * a.pitch + kxc];
  }
  fft512_wave<-1>(v, L, lane, twN, twB, lane);
  if (init_only) {
    if (on) {
#pragma unroll
      for (int t = 0; t < 8; ++t) chat[(int64_t)(T + 64 * t) * a.pitch + kx] = v[t];
    }
    return;
  }
  const double kxv = a.kx0 * kxc;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int ky = T + 64 * t;
    const int my = 2 * ky > N ? ky - N : ky;
    const double kyv = a.ky0 * my;
    const double k2 = (kxv * kxv + kyv * kyv) + 0.0;
    const double num = a.dtM * k2;
    const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
    double2 r;
    r.x = fma(-num, v[t].x, ch[t].x) * den;
    r.y = fma(-num, v[t].y, ch[t].y) * den;
    if (on) chat[(int64_t)ky * a.pitch + kx] = r;
    v[t] = make_double2(r.x * a.inv_n, r.y * a.inv_n);
  }
  __syncthreads();
  fft512_wave<+1>(v, L, T, twT, twB, lane);
  if (on) {
#pragma unroll
    for (int t = 0; t < 8; ++t) H[(int64_t)(T + 64 * t) * a.pitch + kx] = v[t];
  }
}

// =====================================================================================================================
// 3-D (512^3): the same one-wave radix-8 transforms applied along y and along z of the [z][y][kx] half spectrum.
// A workgroup owns 8 adjacent k_x columns (one full 128-byte line per row of the column) of one "batch" (a z-plane for
// the y pass, a y-row for the z pass), moves them cooperatively with the column index fastest and stages them through
// LDS; one wave transforms one column.  Passes per step (4 launches, 11.9 GB of traffic instead of rocFFT's ~18 GB):
//   Z: forward z-FFT of G -> k-space update of the resident spectrum -> inverse z-FFT -> H      (MODE 2)
//   Y: inverse y-FFT of H in place                                                              (MODE 1)
//   X: f2_row512_kernel over all ny*nz rows: inverse x-FFT -> c stored -> f'(c) -> forward x-FFT -> G
//   Y: forward y-FFT of G in place                                                              (MODE 0)
// MODE 3 = forward z-FFT stored as the resident spectrum (initialisation).
template <int MODE, int CW3, bool EARLY = false>  // EARLY (MODE 2): request the resident spectrum before the forward FFT
__global__ __launch_bounds__(64 * CW3, 4) void f3_col512_kernel(const F2Args a, double2* __restrict__ A,
                                                             double2* __restrict__ chat, double2* __restrict__ H,
                                                             int64_t col_stride, int64_t batch_stride, int nblk,
                                                             int nitems, const double2* __restrict__ twA_g,
                                                             const double2* __restrict__ twB_g, const ColSplit sp) {
  __shared__ __attribute__((aligned(16))) double2 Lall[CW3 * W8C];
  constexpr int N = 512, NT = 64 * CW3, PER = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double2* L = Lall + wave * W8C + 4 * wave;
  const int T = (lane >> 3) + 8 * (lane & 7);
  auto nat = [](int n) { return n + (n >> 3); };
  const int ci = tid % CW3;
  double2* Lc = Lall + ci * W8C + 4 * ci;
  double2 v[8], ch[PER];
  // One work item (batch b, block of CW3 k_x columns) per workgroup.  (A persistent, software-pipelined form -- next
  // item's loads in flight during the transforms -- was measured and is slower: 3.85 vs 3.25 ms per step; its extra 32
  // VGPRs cost a wave per SIMD, and short-lived workgroups already overlap through the dispatcher.)
  // CW3 = 4: a 128-byte line holds two items' columns; items i and i + 8 are taken by workgroups on the same XCD back
  // to back (round-robin dispatch), so give THEM the two halves of one line.
  {
    const int item = blockIdx.x;
    int lb = item;
    if (CW3 == 4) {
      const int grp = lb >> 4, r = lb & 15;
      if ((grp << 4) + 16 <= nitems) lb = (grp << 4) + ((r & 7) << 1) + (r >> 3);  // ragged last group: identity
    }
    const int b = lb / nblk, kx = (lb % nblk) * CW3 + ci;
    const bool on = kx < a.nxh;
    const int64_t base = (int64_t)b * batch_stride + kx;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int r = (tid + NT * i) / CW3;
      const int64_t src = (MODE == 1 && sp.on == 2) ? split_addr(sp, b, kx, r, col_stride) : base + (int64_t)r * col_stride;
      v[i] = on ? A[src] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) Lc[nat((tid + NT * i) / CW3)] = v[i];
    if (MODE == 2 && EARLY) {  // the resident spectrum is requested now and consumed after the forward transform (its
                               // HBM latency hides behind the FFT) -- 128 VGPRs with ~14 dwords of scratch
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        ch[i] = on ? chat[base + (int64_t)r * col_stride] : make_double2(0.0, 0.0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
    if (MODE == 1)
      fft512_wave_tw<+1>(v, L, lane, twA_g, twB_g, lane);
    else
      fft512_wave_tw<-1>(v, L, lane, twA_g, twB_g, lane);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
    if (MODE == 2 && !EARLY) {  // fetched only now: 108 VGPRs, no scratch, but the load latency is exposed
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        ch[i] = on ? chat[base + (int64_t)r * col_stride] : make_double2(0.0, 0.0);
      }
    }
    __syncthreads();
    if (MODE == 4) {
      // Poisson solve, z pass: multiply by 1 / eigenvalue of the 7-point Laplacian (tables of 2 cos(2 pi m / n) - 2 for the
      // x, y and z axis back to back, carried in `chat`), zero mode -> 0; then inverse z in place
      const double* sym = reinterpret_cast<const double*>(chat);
      const double cxy = sym[kx < a.nxh ? kx : 0] + sym[a.nx + b];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int kz = (tid + NT * i) / CW3;
        const double lam = (cxy + sym[a.nx + a.ny + kz]) * a.dtM;    // dtM = 1 / h^2 here
        const double sc = (kx == 0 && b == 0 && kz == 0) ? 0.0 : (a.inv_n / lam) * a.dtMkappa;  // dtMkappa = -k / eps
        const double2 gh = Lc[nat(kz)];
        Lc[nat(kz)] = make_double2(gh.x * sc, gh.y * sc);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
      fft512_wave_tw<+1>(v, L, lane, twA_g, twB_g, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        if (on) A[base + (int64_t)r * col_stride] = Lc[nat(r)];
      }
    } else if (MODE == 0 || MODE == 1 || MODE == 3) {
      double2* dst = MODE == 3 ? chat : (sp.on ? H : A);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        const int64_t di = (MODE == 0 && sp.on == 1) ? split_addr(sp, b, kx, r, col_stride) : base + (int64_t)r * col_stride;
        if (on) dst[di] = Lc[nat(r)];
      }
    } else {
      // MODE 2: this is the z pass -- b is the (local) y index, the row along the column is k_z
      const int gy = b + a.yoff;
      const int my = 2 * gy > a.ny ? gy - a.ny : gy;
      const double kxv = a.kx0 * kx, kyv = a.ky0 * my;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int kz = (tid + NT * i) / CW3;
        const int mz = 2 * kz > N ? kz - N : kz;
        const double kzv = a.kz0 * mz;
        const double k2 = (kxv * kxv + kyv * kyv) + kzv * kzv;  // same grouping as spectral.hip's ksq
        const double num = a.dtM * k2;
        const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
        const double2 gh = Lc[nat(kz)];
        double2 r;
        r.x = fma(-num, gh.x, ch[i].x) * den;
        r.y = fma(-num, gh.y, ch[i].y) * den;
        if (on) chat[base + (int64_t)kz * col_stride] = r;
        Lc[nat(kz)] = make_double2(r.x * a.inv_n, r.y * a.inv_n);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
      fft512_wave_tw<+1>(v, L, lane, twA_g, twB_g, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        if (on) H[base + (int64_t)r * col_stride] = Lc[nat(r)];
      }
    }
  }
}

// Column passes for the other power-of-two axis lengths (128, 256, 1024): the multi-stage radix-2^2 LDS transform of the
// 2-D path (fft_inplace), G threads per column (one radix-4 group each: 64 up to 256 points, 256 for 1024 -- sized so
// that the kernels stay under 64 VGPRs and fill the CU: the first form, one wave per 1024-capable column, needed 162-178
// VGPRs), CWG adjacent k_x columns per workgroup moved with the column index fastest.  Same MODEs as f3_col512_kernel.
template <int MODE, int CWG, int G, int NMAX>  // G threads per column, axis length N <= NMAX
__global__ __launch_bounds__(G * CWG, 8) void f3_col_kernel(const F2Args a, double2* __restrict__ A,
                                                          double2* __restrict__ chat, double2* __restrict__ H,
                                                          int64_t col_stride, int64_t batch_stride, int nblk, int nitems,
                                                          int N, int lg, const double2* __restrict__ tw_g,
                                                          const ColSplit sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int NP = px(N) + 1;
  double2* X = reinterpret_cast<double2*>(smem_raw);  // [CWG][NP]
  double2* TW = X + CWG * NP;
  constexpr int NT = G * CWG, MAXP = NMAX / G;  // N / G elements per thread
  const int tid = threadIdx.x, lane = tid % G, wave = tid / G, ci = tid % CWG;
  int lb = blockIdx.x;
  if (CWG == 4) {  // two items per 128-byte line: hand the halves to workgroups that land on the same XCD (i, i + 8)
    const int grp = lb >> 4, r = lb & 15;
    if ((grp << 4) + 16 <= nitems) lb = (grp << 4) + ((r & 7) << 1) + (r >> 3);
  }
  const int b = lb / nblk, kx = (lb % nblk) * CWG + ci;
  const bool on = kx < a.nxh;
  const int64_t base = (int64_t)b * batch_stride + kx;
  for (int k = tid; k < N / 2; k += NT) TW[k] = tw_g[k];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int r = (tid + NT * i) / CWG;
    const int64_t src = (MODE == 1 && sp.on == 2) ? split_addr(sp, b, kx, r, col_stride) : base + (int64_t)r * col_stride;
    if (r < N) X[ci * NP + px(brev(r, lg))] = on ? A[src] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  if (MODE == 1)
    fft_inplace<+1, G, NMAX>(X + wave * NP, TW, N, lg, lane);
  else
    fft_inplace<-1, G, NMAX>(X + wave * NP, TW, N, lg, lane);
  if (MODE == 0 || MODE == 1 || MODE == 3) {
    double2* dst = MODE == 3 ? chat : (sp.on ? H : A);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int r = (tid + NT * i) / CWG;
      const int64_t di = (MODE == 0 && sp.on == 1) ? split_addr(sp, b, kx, r, col_stride) : base + (int64_t)r * col_stride;
      if (r < N && on) dst[di] = X[ci * NP + px(r)];
    }
    return;
  }
  double2 o[MAXP];
  if (MODE == 2) {  // z pass: b is the (local) y index, the row along the column is k_z
    const int gy = b + a.yoff;
    const int my = 2 * gy > a.ny ? gy - a.ny : gy;
    const double kxv = a.kx0 * kx, kyv = a.ky0 * my;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int kz = (tid + NT * i) / CWG;
      if (kz < N) {
        const int mz = 2 * kz > N ? kz - N : kz;
        const double kzv = a.kz0 * mz;
        const double k2 = (kxv * kxv + kyv * kyv) + kzv * kzv;  // same grouping as spectral.hip's ksq
        const double num = a.dtM * k2;
        const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
        const double2 gh = X[ci * NP + px(kz)];
        const double2 ch = on ? chat[base + (int64_t)kz * col_stride] : make_double2(0.0, 0.0);
        double2 r;
        r.x = fma(-num, gh.x, ch.x) * den;
        r.y = fma(-num, gh.y, ch.y) * den;
        if (on) chat[base + (int64_t)kz * col_stride] = r;
        o[i] = make_double2(r.x * a.inv_n, r.y * a.inv_n);
      }
    }
  } else {  // MODE 4: Poisson solve, divide by the eigenvalue of the 7-point Laplacian (see f3_col512_kernel)
    const double* sym = reinterpret_cast<const double*>(chat);
    const double cxy = sym[on ? kx : 0] + sym[a.nx + b];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int kz = (tid + NT * i) / CWG;
      if (kz < N) {
        const double lam = (cxy + sym[a.nx + a.ny + kz]) * a.dtM;
        const double sc = (kx == 0 && b == 0 && kz == 0) ? 0.0 : (a.inv_n / lam) * a.dtMkappa;
        const double2 gh = X[ci * NP + px(kz)];
        o[i] = make_double2(gh.x * sc, gh.y * sc);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int r = (tid + NT * i) / CWG;
    if (r < N) X[ci * NP + px(brev(r, lg))] = o[i];
  }
  __syncthreads();
  fft_inplace<+1, G, NMAX>(X + wave * NP, TW, N, lg, lane);
  double2* dst = MODE == 2 ? H : A;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int r = (tid + NT * i) / CWG;
    if (r < N && on) dst[base + (int64_t)r * col_stride] = X[ci * NP + px(r)];
  }
}

int g_generic512 = 0;  // PFHIP_FFT3D_GENERIC512 = 1: 512-point columns by f3_col_kernel too (A/B against the radix-8 wave kernel)
int g_cwg = 0;  // columns per workgroup of f3_col_kernel on 128- / 256-point axes: 0 = 8; PFHIP_FFT3D_CWG = 4 | 8
int g_zearly = 0;  // z pass: request the resident spectrum before the forward FFT (PFHIP_FFT3D_ZEARLY = 0 | 1)
int g_cw3 = 0;  // k_x columns per workgroup of the 3-D column passes: 0 = per pass (z: 4, y: 8), PFHIP_FFT3D_CW = 4 | 8 forces one
int g_cw512 = 1;  // columns per workgroup of the 2-D column kernel (PFHIP_FFT512_CW = 1 | 2 | 4): 13.05 / 14.3 / 17.5 us

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

}  // namespace

struct Fused2D {
  F2Args a;
  double2 *twx = nullptr, *twy = nullptr;
  double2 *tw8a = nullptr, *tw8b = nullptr;  // radix-8 tables of the 512-point fast path
  bool row512 = false, col512 = false;
  double* sym = nullptr;  // 512^3 Poisson: 3 x 512 doubles, 2 cos(2 pi m / n) - 2 per axis
  bool cube512 = false;  // 3-D box: x rows by f2_row512_kernel / f2_row_kernel, y and z columns by f3_col512_kernel (512-point
                         // axes) / f3_col_kernel (128, 256, 1024)
  double2* twz = nullptr;
  int lgz = 0;
  size_t lds_row = 0, lds_col = 0;
  hipStream_t stream = nullptr;
  bool g_valid = false;  // G holds the row transform of f'(current c)
};

bool fused2d_supported(int dim, int nx, int ny, int nz) {
  if (dim == 3) {
    const char* e = getenv("PFHIP_SPECTRAL_3D");  // "rocfft" forces the library path (A/B comparison)
    if (e && std::string(e) == "rocfft") return false;
    const int lx = ilog2(nx), ly = ilog2(ny), lz = ilog2(nz);
    if (!(lx >= 7 && lx <= 10 && ly >= 7 && ly <= 10 && lz >= 7 && lz <= 10)) return false;
    // measured per step against the rocFFT path (profiles/r02/spectral3d_ab_sizes.log, bench_spectral_sizes.log):
    // 128^3 0.080 ms vs 0.088; 256^3 0.375 vs 0.567; 512^3 2.64 vs 3.5; 1024^3 24.9 vs 49.3
    return true;
  }
  const int lx = ilog2(nx), ly = ilog2(ny);
  return dim == 2 && lx >= 7 && lx <= 10 && ly >= 7 && ly <= 10;
}

// Row pitch (complex elements) of every half-spectrum array this file touches.  2-D: nx/2 + 1, rocFFT's D2Z layout (the
// 2 MiB problem is L2-resident).  512^3: 264 instead of 257.  With 257 a column workgroup's 8 adjacent columns (128 bytes
// per row) start 16 bytes further into a 128-byte line on every row, so 7 of 8 rows straddle two lines: rocprofv3
// measured FETCH_SIZE x2 = 2.02 GB for the 1.08 GB a y pass needs (1.875x = (7*2 + 1)/8 exactly) and 4.09 GB for the z
// pass's 2.16 GB, with the passes already at 5.4-5.9 TB/s of HBM traffic (profiles/r02/spectral_512c_before_pitch.md).
// A pitch that is a multiple of 8 makes every (row, column-block) exactly one line.
int fused_spectrum_pitch(int dim, int nx, int ny, int nz) {
  const int nxh = nx / 2 + 1;
  if (dim == 3 && fused2d_supported(dim, nx, ny, nz)) {
    const char* e = getenv("PFHIP_FFT3D_PITCH");  // "natural": keep nx/2 + 1 (A/B comparison)
    if (!(e && std::string(e) == "natural")) return (nxh + 7) / 8 * 8;
  }
  return nxh;
}

int fused2d_create(Fused2D** out, int nx, int ny, int nz, double h, hipStream_t stream) {
  Fused2D* f = new Fused2D();
  *out = f;
  f->stream = stream;
  F2Args& a = f->a;
  f->cube512 = nz > 1;
  a.nz = nz;
  a.kz0 = nz > 1 ? TWO_PI_F / (nz * h) : 0.0;
  a.nx = nx;
  a.ny = ny;
  a.nxh = nx / 2 + 1;
  a.pitch = fused_spectrum_pitch(nz > 1 ? 3 : 2, nx, ny, nz);
  a.lgx = ilog2(nx);
  a.lgy = ilog2(ny);
  a.kx0 = TWO_PI_F / (nx * h);
  a.ky0 = TWO_PI_F / (ny * h);
  a.inv_n = 1.0 / ((double)nx * ny * nz);
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
  if (nz > 1) {
    f->lgz = ilog2(nz);
    if (table(nz, &f->twz) != hipSuccess) return -3;
  }
  const char* e = getenv("PFHIP_FFT512");  // "radix2": keep the multi-wave radix-2^2 kernels (A/B comparison)
  const bool allow8 = !(e && std::string(e) == "radix2");
  f->row512 = (allow8 || f->cube512) && nx == 512 && (ny / 2) % RW == 0;
  if (const char* rg = getenv("PFHIP_FFT3D_ROW"))  // "generic": 512-point rows of a 3-D box by f2_row_kernel (A/B)
    if (f->cube512 && std::string(rg) == "generic") f->row512 = false;
  f->col512 = (allow8 || f->cube512) && (ny == 512 || nz == 512);  // (3-D: "some column pass needs the radix-8 tables")
  if (const char* c3 = getenv("PFHIP_FFT3D_CW")) g_cw3 = std::atoi(c3) == 4 ? 4 : 8;
  if (const char* ze = getenv("PFHIP_FFT3D_ZEARLY")) g_zearly = std::atoi(ze) != 0;
  if (const char* gg = getenv("PFHIP_FFT3D_GENERIC512")) g_generic512 = std::atoi(gg) != 0;
  if (const char* cg = getenv("PFHIP_FFT3D_CWG")) {
    const int c = std::atoi(cg);
    g_cwg = (c == 4 || c == 8) ? c : 0;
  }

  if (const char* cw = getenv("PFHIP_FFT512_CW")) {
    const int c = std::atoi(cw);
    if (c == 1 || c == 2 || c == 4) g_cw512 = c;
  }
  if (f->row512 || f->col512) {
    std::vector<double2> ta(512), tb(64);
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < 8; ++q) {
        const double ang = TWO_PI_F * (double)(l * q) / 512.0;
        ta[l * 8 + q] = make_double2(std::cos(ang), -std::sin(ang));
      }
    for (int l0 = 0; l0 < 8; ++l0)
      for (int sidx = 0; sidx < 8; ++sidx) {
        const double ang = TWO_PI_F * (double)(l0 * sidx) / 64.0;
        tb[l0 * 8 + sidx] = make_double2(std::cos(ang), -std::sin(ang));
      }
    if (hipMalloc(&f->tw8a, sizeof(double2) * 512) != hipSuccess ||
        hipMalloc(&f->tw8b, sizeof(double2) * 64) != hipSuccess ||
        hipMemcpy(f->tw8a, ta.data(), sizeof(double2) * 512, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->tw8b, tb.data(), sizeof(double2) * 64, hipMemcpyHostToDevice) != hipSuccess)
      return -3;
  }
  f->lds_row = sizeof(double2) * (nx + nx / 32 + nx / 2);
  f->lds_col = sizeof(double2) * ((size_t)CW * (ny + ny / 32 + 1) + ny / 2);
  if (f->lds_col > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(f2_col_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)f->lds_col) != hipSuccess)
    return -3;
  if (f->cube512) {  // 1024-point columns: 4 columns of 1057 slots + 512 twiddles = 76 KB of LDS per workgroup
    auto big = [](const void* k) {
      return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    };
    if (!big(reinterpret_cast<const void*>(f3_col_kernel<0, 8, 128, 512>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<1, 8, 128, 512>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<2, 8, 128, 512>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<3, 8, 128, 512>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<4, 8, 128, 512>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<0, 4, 256, 1024>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<1, 4, 256, 1024>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<2, 4, 256, 1024>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<3, 4, 256, 1024>)) ||
        !big(reinterpret_cast<const void*>(f3_col_kernel<4, 4, 256, 1024>)))
      return -3;
  }
  return 0;
}

void fused2d_destroy(Fused2D* f) {
  if (!f) return;
  if (f->twz) (void)hipFree(f->twz);
  if (f->twx) (void)hipFree(f->twx);
  if (f->twy) (void)hipFree(f->twy);
  if (f->tw8a) (void)hipFree(f->tw8a);
  if (f->tw8b) (void)hipFree(f->tw8b);
  if (f->sym) (void)hipFree(f->sym);
  delete f;
}

void fused2d_invalidate(Fused2D* f) { f->g_valid = false; }

namespace {
void launch_row(const Fused2D* f, const F2Args& a, const double2* H, const double* c_in, double* c_out, double2* G,
                int from_spectrum, int use_fprime) {
  if (f->row512)
    hipLaunchKernelGGL(f2_row512_kernel<false>, dim3(a.ny / 2 / RW), dim3(64 * RW), 0, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, from_spectrum, use_fprime);
  else
    hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_row, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->twx, from_spectrum, use_fprime);
}
void launch_col(const Fused2D* f, const F2Args& a, const double2* G, double2* chat, double2* H, int init_only) {
  if (f->col512 && g_cw512 == 4)
    hipLaunchKernelGGL(f2_col512_direct_kernel<4>, dim3((a.nxh + 3) / 4), dim3(256), 0, f->stream, a, G, chat, H,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, init_only);
  else if (f->col512 && g_cw512 == 2)
    hipLaunchKernelGGL(f2_col512_direct_kernel<2>, dim3((a.nxh + 1) / 2), dim3(128), 0, f->stream, a, G, chat, H,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, init_only);
  else if (f->col512)
    hipLaunchKernelGGL(f2_col512_direct_kernel<1>, dim3(a.nxh), dim3(64), 0, f->stream, a, G, chat, H,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, init_only);
  else
    hipLaunchKernelGGL(f2_col_kernel, dim3((a.nxh + CW - 1) / CW), dim3(CT * CW), f->lds_col, f->stream, a, G, chat, H,
                       (const double2*)f->twy, init_only);
}
}  // namespace

namespace {
// 3-D passes (512^3).  Rows: f2_row512_kernel over ny*nz/2 row pairs (its row index is the flattened (z, y) index).
void launch_row3(const Fused2D* f, const F2Args& a, const double2* H, const double* c_in, double* c_out, double2* G,
                 int from_spectrum, int use_fprime) {
  if (f->row512)
    hipLaunchKernelGGL(f2_row512_kernel<true>, dim3(a.ny * a.nz / 2 / RW), dim3(64 * RW), 0, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, from_spectrum, use_fprime);
  else
    hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny * a.nz / 2), dim3(RT), f->lds_row, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->twx, from_spectrum, use_fprime);
}
// where the columns of a pass live: N points along the column, nbatch batches of ceil(nxh / CW) column blocks
struct ColGeom {
  int N, lg;
  const double2* tw;
  int64_t col_stride, batch_stride;
  int nbatch;
  ColSplit sp;
};
ColGeom axis_geom(const Fused2D* f, const F2Args& a, int axis) {
  const int64_t row = a.pitch, plane = (int64_t)a.pitch * a.ny;
  ColGeom g;
  // axis 1: columns along y (stride one x-row), one batch per z-plane; axis 2: columns along z, one batch per y-row
  g.N = axis == 1 ? a.ny : a.nz;
  g.lg = axis == 1 ? a.lgy : f->lgz;
  g.tw = axis == 1 ? f->twy : f->twz;
  g.col_stride = axis == 1 ? row : plane;
  g.batch_stride = axis == 1 ? plane : row;
  g.nbatch = axis == 1 ? a.nz : a.ny;
  return g;
}
template <int MODE, int CWG, int G, int NMAX>
void launch_col3_g(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int nblk = (a.nxh + CWG - 1) / CWG;
  const size_t lds = sizeof(double2) * ((size_t)CWG * (g.N + g.N / 32 + 1) + g.N / 2);
  hipLaunchKernelGGL((f3_col_kernel<MODE, CWG, G, NMAX>), dim3(nblk * g.nbatch), dim3(G * CWG), lds, f->stream, a, A, chat,
                     H, g.col_stride, g.batch_stride, nblk, nblk * g.nbatch, g.N, g.lg, g.tw, g.sp);
}
template <int MODE, int CW3>
void launch_col3_t(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int nblk = (a.nxh + CW3 - 1) / CW3;
  const int nitems = nblk * g.nbatch;
  if (MODE == 2 && g_zearly)
    hipLaunchKernelGGL((f3_col512_kernel<MODE, CW3, true>), dim3(nitems), dim3(64 * CW3), 0, f->stream, a, A, chat, H,
                       g.col_stride, g.batch_stride, nblk, nitems, (const double2*)f->tw8a, (const double2*)f->tw8b, g.sp);
  else
    hipLaunchKernelGGL((f3_col512_kernel<MODE, CW3, false>), dim3(nitems), dim3(64 * CW3), 0, f->stream, a, A, chat, H,
                       g.col_stride, g.batch_stride, nblk, nitems, (const double2*)f->tw8a, (const double2*)f->tw8b, g.sp);
}
template <int MODE>
void launch_col3_geom(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int N = g.N;
  if (N != 512 || g_generic512) {
    if (N == 512) {
      if (g_cwg == 4)
        launch_col3_g<MODE, 4, 128, 512>(f, a, A, chat, H, g);
      else
        launch_col3_g<MODE, 8, 128, 512>(f, a, A, chat, H, g);
      return;
    }
    // columns per workgroup, measured at 256^3: 8 -> 0.363 ms per step, 4 -> 0.379 (rocFFT path 0.578)
    const int cwg = g_cwg ? g_cwg : 8;
    if (N > 512)  // 4 columns of 1024 points: 76 KB of LDS, 1024 threads
      launch_col3_g<MODE, 4, 256, 1024>(f, a, A, chat, H, g);
    else if (cwg == 8)
      launch_col3_g<MODE, 8, 64, 256>(f, a, A, chat, H, g);
    else
      launch_col3_g<MODE, 4, 64, 256>(f, a, A, chat, H, g);
    return;
  }
  // measured after the aligned pitch (rocprofv3, 512^3): z pass 999 us with 4 columns per workgroup (4 workgroups of 4
  // waves per CU) vs 1073 us with 8; y passes 441-452 us with 8 vs 466-471 us with 4
  const int cw = g_cw3 ? g_cw3 : ((MODE == 2 || MODE == 4 || MODE == 3) ? 4 : 8);
  if (cw == 4)
    launch_col3_t<MODE, 4>(f, a, A, chat, H, g);
  else
    launch_col3_t<MODE, 8>(f, a, A, chat, H, g);
}
template <int MODE>
void launch_col3(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, int axis) {
  launch_col3_geom<MODE>(f, a, A, chat, H, axis_geom(f, a, axis));
}
}  // namespace

// 512^3 periodic Poisson solve lap_h(phi) = -(k/eps) c with the same passes: x forward, y forward, z (forward, divide
// by the Laplacian's eigenvalue, inverse), y inverse, x inverse only.  W: nh complex work array.
int fused3d_poisson(Fused2D* f, const double* c, double* phi, double2* W, double k_over_eps, double inv_h2) {
  if (!f->cube512) return -3;
  if (!f->sym) {
    const int nn[3] = {f->a.nx, f->a.ny, f->a.nz};
    std::vector<double> t;
    for (int d = 0; d < 3; ++d)
      for (int m = 0; m < nn[d]; ++m) t.push_back(2.0 * std::cos(TWO_PI_F * m / nn[d]) - 2.0);
    if (hipMalloc(&f->sym, sizeof(double) * t.size()) != hipSuccess ||
        hipMemcpy(f->sym, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice) != hipSuccess)
      return -3;
  }
  F2Args a = f->a;
  a.dtM = inv_h2;
  a.dtMkappa = -k_over_eps;
  launch_row3(f, a, nullptr, c, nullptr, W, 0, 0);
  launch_col3<0>(f, a, W, nullptr, nullptr, 1);
  launch_col3<4>(f, a, W, reinterpret_cast<double2*>(f->sym), nullptr, 2);
  launch_col3<1>(f, a, W, nullptr, nullptr, 1);
  launch_row3(f, a, W, nullptr, phi, nullptr, 2, 0);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// chat <- 2-D spectrum of c (same as a rocFFT D2Z); G is clobbered
int fused2d_spectrum(Fused2D* f, const double* c, double2* chat, double2* G) {
  const F2Args& a = f->a;
  if (f->cube512) {
    launch_row3(f, a, nullptr, c, nullptr, G, 0, 0);
    launch_col3<0>(f, a, G, nullptr, nullptr, 1);
    launch_col3<3>(f, a, G, chat, nullptr, 2);
    f->g_valid = false;
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  launch_row(f, a, nullptr, c, nullptr, G, 0, 0);
  launch_col(f, a, G, chat, nullptr, 1);
  f->g_valid = false;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// one semi-implicit step c_in -> c_out; chat (resident, valid for c_in) is advanced; G, H are work arrays
int fused2d_step(Fused2D* f, const double* c_in, double* c_out, double2* chat, double2* G, double2* H, double dt,
                 double M, double kappa, double ca, double cb, double two_rho, double gam) {
  F2Args a = f->a;
  a.ca = ca;
  a.cb = cb;
  a.two_rho = two_rho;
  a.dtM = dt * M;
  a.dtMkappa = dt * M * kappa;
  a.gam = gam;
  if (f->cube512) {
    if (!f->g_valid) {  // x- and y-transform of f'(c_in) (first step, or after the field was replaced)
      launch_row3(f, a, nullptr, c_in, nullptr, G, 0, 1);
      launch_col3<0>(f, a, G, nullptr, nullptr, 1);
    }
    launch_col3<2>(f, a, G, chat, H, 2);        // z: forward, k-space update of chat, inverse -> H
    launch_col3<1>(f, a, H, nullptr, nullptr, 1);  // y: inverse, in place
    launch_row3(f, a, H, nullptr, c_out, G, 1, 1);  // x: inverse -> c_out, f'(c_out), forward -> G
    launch_col3<0>(f, a, G, nullptr, nullptr, 1);  // y: forward, in place (ready for the next step's z pass)
    f->g_valid = true;
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (!f->g_valid)  // row transform of f'(c_in) (first step, or after the field was replaced)
    launch_row(f, a, nullptr, c_in, nullptr, G, 0, 1);
  launch_col(f, a, G, chat, H, 0);
  launch_row(f, a, H, nullptr, c_out, G, 1, 1);
  f->g_valid = true;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- slab-decomposed transforms (csrc/slabfft.hip) ----------------------------------------------------------------------
// The local passes of a rank that owns nzl z-planes for the x and y transforms and, after the all-to-all, nyl = ny / P
// y-rows of every z-plane for the z transform ("T" layout [z][yq][kx]).  Every spectrum array uses the padded pitch.
bool fusedslab_supported(int nx, int ny, int nz, int P) {
  return P >= 1 && ny % P == 0 && nz % P == 0 && ilog2(P) >= 0 && fused2d_supported(3, nx, ny, nz);
}
namespace {
ColGeom slab_y_geom(const Fused2D* f, const F2Args& a, int nzl, int P, int on) {
  ColGeom g = axis_geom(f, a, 1);  // a.nz == nzl: one batch per local plane
  const int nyl = a.ny / P;
  g.sp.on = on;
  g.sp.lg = ilog2(nyl);
  g.sp.chunk = (int64_t)nzl * nyl * a.pitch;
  g.sp.bstride = (int64_t)nyl * a.pitch;
  return g;
}
}  // namespace
// real planes -> x rows (of f'(c) if use_fprime) -> tmp [zl][y][kx] -> y columns -> A [q][zl][yq][kx] (all-to-all layout)
int fusedslab_forward_xy(Fused2D* f, const double* real_in, double2* tmp, double2* A, int nzl, int P, int use_fprime,
                         double ca, double cb, double two_rho) {
  F2Args a = f->a;
  a.nz = nzl;
  a.ca = ca;
  a.cb = cb;
  a.two_rho = two_rho;
  launch_row3(f, a, nullptr, real_in, nullptr, tmp, 0, use_fprime);
  launch_col3_geom<0>(f, a, tmp, nullptr, A, slab_y_geom(f, a, nzl, P, 1));
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// A [q][zl][yq][kx] -> inverse y columns -> tmp [zl][y][kx] -> inverse x rows -> real planes
int fusedslab_inverse_yx(Fused2D* f, double2* A, double2* tmp, double* real_out, int nzl, int P) {
  F2Args a = f->a;
  a.nz = nzl;
  launch_col3_geom<1>(f, a, A, nullptr, tmp, slab_y_geom(f, a, nzl, P, 2));
  launch_row3(f, a, tmp, nullptr, real_out, nullptr, 2, 0);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// z columns of B (T layout, nyl local y-rows starting at global row yoff), in place.  mode 0: forward; 1: inverse
// (unnormalised); 2: forward -> k-space update of the resident chat -> inverse of chat / N; 3: forward, stored to chat.
int fusedslab_z(Fused2D* f, double2* B, double2* chat, int mode, int nyl, int yoff, double dtM, double dtMkappa) {
  F2Args a = f->a;
  a.yoff = yoff;
  a.dtM = dtM;
  a.dtMkappa = dtMkappa;
  a.gam = 0.0;
  ColGeom g;
  g.N = a.nz;
  g.lg = f->lgz;
  g.tw = f->twz;
  g.col_stride = (int64_t)nyl * a.pitch;
  g.batch_stride = a.pitch;
  g.nbatch = nyl;
  if (mode == 0)
    launch_col3_geom<0>(f, a, B, nullptr, nullptr, g);
  else if (mode == 1)
    launch_col3_geom<1>(f, a, B, nullptr, nullptr, g);
  else if (mode == 2)
    launch_col3_geom<2>(f, a, B, chat, B, g);
  else
    launch_col3_geom<3>(f, a, B, chat, nullptr, g);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace pfhip
