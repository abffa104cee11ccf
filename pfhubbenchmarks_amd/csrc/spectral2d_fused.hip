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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pfhip_internal.h"
#include "fft512_wave.h"

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
  int nyp = 0;       // rows RESERVED per z-plane of a half-spectrum array (0 = ny): single-GPU 3-D boxes keep one pad row per
                     // plane, see fused_spectrum_layout()
};

// Row (z, y) of a half-spectrum array starts at (z nyp + y) pitch complex elements (nyp = ny + pad rows); `row` = the
// flattened index z ny + y (2-D: z = 0).
__host__ __device__ __forceinline__ int64_t spec_row_flat(const F2Args& a, int row) {
  if (a.nyp == 0 || a.nyp == a.ny) return (int64_t)row * a.pitch;
  const int z = a.lgy >= 0 ? row >> a.lgy : row / a.ny;  // ny a power of two except on the mixed-radix path
  return ((int64_t)row + (int64_t)z * (a.nyp - a.ny)) * a.pitch;
}
__host__ __device__ __forceinline__ int64_t spec_plane(const F2Args& a) { return (int64_t)(a.nyp ? a.nyp : a.ny) * a.pitch; }

// Where element r of the column of batch b lives (complex elements, k_x added by the caller):
//   (r >> lr) rchunk + (r & (2^lr - 1)) rstride + (b >> lb) bchunk + (b & (2^lb - 1)) bstride        (lr / lb = 31: no split)
// covers the plain layout for both column directions and the all-to-all layout
// [row / split][batch][row % split][kx] that the column passes of the slab-decomposed transforms (csrc/slabfft.hip) read /
// write on the fly.  A pass gets two maps: `mn` for its natural side and `ms` for the other side when spon != 0
// (spon = 1: the store side of a MODE 0 pass goes to H through ms; spon = 2: the load side of a MODE 1 pass comes from A
// through ms and the result goes to H through mn).
struct ColMap {
  int lr = 31, lb = 31;
  int64_t rchunk = 0, rstride = 0, bchunk = 0, bstride = 0;
};
__device__ __forceinline__ int64_t col_addr(const ColMap& m, int b, int r) {
  return (int64_t)(r >> m.lr) * m.rchunk + (int64_t)(r & (int)((1u << m.lr) - 1u)) * m.rstride +
         (int64_t)(b >> m.lb) * m.bchunk + (int64_t)(b & (int)((1u << m.lb) - 1u)) * m.bstride;
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
__global__ __launch_bounds__(RT) void f2_row_kernel(const F2Args a, const double2* H,  // (H == G allowed)
                                                    const double* __restrict__ c_in, double* __restrict__ c_out,
                                                    double2* G, const double2* __restrict__ twx_g,
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
      const double2 p = H[spec_row_flat(a, y0) + k], q = H[spec_row_flat(a, y1) + k];
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
    G[spec_row_flat(a, y0) + k] = make_double2(0.5 * (w.x + m.x), 0.5 * (w.y - m.y));
    G[spec_row_flat(a, y1) + k] = make_double2(0.5 * (w.y + m.y), -0.5 * (w.x - m.x));
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
// LDS layout of the exchanges: 8 blocks of 72 slots (64 + 8 pad); exchange 1 at [q*72 + l + l/8] (input index l skewed by
// one slot per 8: conflict-free for BOTH lane -> l maps in use, m = lane and the chained m = T(lane) -- the unskewed form
// put the 8 lanes of a 16-byte write group of the chained map on two bank groups, a 4-way conflict on every store of the
// row pass's forward transform: round 3, SQ_LDS_BANK_CONFLICT = 50 % of its LDS cycles), exchange 2 at [q*72 + 9*l0 + s]:
// both directions of both exchanges are bank-conflict free per group of 16-byte accesses.
// Row kernel, nx == 512: one wave per pair of rows; same contract as f2_row_kernel.
constexpr int RW = 1;  // row pairs (waves) per workgroup
// LAZY: twiddles fetched right before each use (fft512_wave_tw) instead of held in 84 VGPRs -- the 512^3 pass streams
// from HBM and wants occupancy; the latency-bound 2-D step keeps them resident (loaded beside the data at kernel entry).
template <bool LAZY>
__global__ __launch_bounds__(64 * RW) void f2_row512_kernel(const F2Args a, const double2* H,  // (H == G allowed)
                                                            const double* __restrict__ c_in,
                                                            double* __restrict__ c_out, double2* G,
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
      const double2 p = H[spec_row_flat(a, y0) + kk], q = H[spec_row_flat(a, y1) + kk];
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
      G[spec_row_flat(a, y0) + k] = make_double2(0.5 * (w.x + mm.x), 0.5 * (w.y - mm.y));
      G[spec_row_flat(a, y1) + k] = make_double2(0.5 * (w.y + mm.y), -0.5 * (w.x - mm.x));
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


// ---- per-XCD work queues for the 3-D column passes ---------------------------------------------------------------------
// A 4-column workgroup moves 64-byte half lines; the two workgroups that share a 128-byte line only save the second HBM
// fetch if they run on the same XCD (same L2) at about the same time.  A static blockIdx -> item map cannot promise
// that once the dispatcher refills the XCDs out of order (round 2: the z pass fetched 1.57x its bytes).  Instead every
// workgroup is persistent, reads the id of the XCD it landed on and pulls items from THAT XCD's queue: an XCD owns a
// contiguous run of items and hands them out in order, so neighbouring items are taken by the same L2 within a fraction
// of a microsecond of each other, wherever the dispatcher put the workgroups.  An XCD whose queue is empty steals from
// the others (tail balance).  Placement only affects speed, never results: every item is processed exactly once.
// Counters: two sets of 8 heads (one 128-byte line each); launch n uses set n & 1 and block 0 zeroes the other set for
// launch n + 1 (launches of one Fused2D are ordered on one stream).
constexpr int QSTRIDE = 32;  // ints per head: one 128-byte line
struct XcdQueue {
  int* head;
  int per, nitems, xcc, k0;
  __device__ __forceinline__ void init(int* set, int* other, int n) {
    head = set;
    nitems = n;
    per = ((n + 7) / 8 + 1) & ~1;  // even: a line-sharing pair never straddles two XCDs' runs
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc = id & 7;
    k0 = 0;
    if (blockIdx.x == 0 && threadIdx.x < 8) other[threadIdx.x * QSTRIDE] = 0;
  }
  // the same for ONE wave (workgroups whose waves work independently): lane 0 pops, the value is broadcast
  __device__ __forceinline__ int pop_wave() {
    int it = nitems;
    if ((threadIdx.x & 63) == 0) {
      for (; k0 < 8; ++k0) {
        const int x = (xcc + k0) & 7;
        const int lo = x * per, hi = lo + per < nitems ? lo + per : nitems;
        if (lo >= hi) continue;
        const int t = __hip_atomic_fetch_add(head + x * QSTRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lo + t < hi) {
          it = lo + t;
          break;
        }
      }
    }
    return __builtin_amdgcn_readfirstlane(it);
  }
  // next item for this workgroup, or nitems when every queue is empty (called by all threads; two barriers)
  __device__ __forceinline__ int pop(int* s_item) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int it = nitems;
      for (; k0 < 8; ++k0) {  // queues only ever run dry, so the scan never goes back
        const int x = (xcc + k0) & 7;
        const int lo = x * per, hi = lo + per < nitems ? lo + per : nitems;
        if (lo >= hi) continue;
        const int t = __hip_atomic_fetch_add(head + x * QSTRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lo + t < hi) {
          it = lo + t;
          break;
        }
      }
      *s_item = it;
    }
    __syncthreads();
    return *s_item;
  }
};

// Row pass of a time step (inverse x -> c -> f'(c) -> forward x) with the NEXT row pair's half-spectrum rows already in
// flight: one persistent wave per workgroup walks row pairs pair, pair + W, pair + 2 W, ...; right after a pair's 16 loads
// have been combined into the transform's input, the loads of the wave's next pair are issued into the same registers and
// stay in flight during the two transforms (16 KB per wave at all times; 3 waves per SIMD at <= 168 VGPRs).
template <int MINW>
__global__ __launch_bounds__(64, MINW) void f3_row512p_kernel(const F2Args a, const double2* H,  // H == G allowed (in place)
                                                              double* __restrict__ c_out, double2* G,
                                                              const double2* __restrict__ twA_g,
                                                              const double2* __restrict__ twB_g, int npairs) {
  __shared__ __attribute__((aligned(16))) double2 L[W8];
  constexpr int N = 512;
  const int lane = threadIdx.x;
  const int T = (lane >> 3) + 8 * (lane & 7);
  double2 p[8], q[8];
  // MINW == 2 (256 VGPRs per wave): the twiddle rows of this lane stay in registers for the wave's whole life (it is
  // persistent) -- 28 L1 loads of 1 KB per row pair less, more bytes than the row pair itself (the lazy form re-reads them
  // from the 9 KB tables before every use to fit 4-5 waves per SIMD)
  constexpr bool HOLD = MINW <= 2;
  double2 twN[7], twT[7], twB[7];
  if (HOLD) {
    load_tw(twN, twA_g, lane);
    load_tw(twT, twA_g, T);
    load_tw(twB, twB_g, lane & 7);
  }
  int pair = blockIdx.x;
  auto issue = [&](int pr) {
    const int64_t r0 = spec_row_flat(a, 2 * pr), r1 = spec_row_flat(a, 2 * pr + 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = lane + 64 * j;
      const int kk = k > N / 2 ? N - k : k;
      p[j] = H[r0 + kk];
      q[j] = H[r1 + kk];
    }
  };
  if (pair < npairs) issue(pair);
  while (pair < npairs) {
    asm volatile("" : "+s"(twA_g), "+s"(twB_g));
    double2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool upper = lane + 64 * j > N / 2;
      v[j] = upper ? make_double2(p[j].x + q[j].y, q[j].x - p[j].y) : make_double2(p[j].x - q[j].y, p[j].y + q[j].x);
    }
    const int y0 = 2 * pair, y1 = y0 + 1;
    const int next = pair + gridDim.x;
    if (next < npairs) issue(next);
    if (HOLD)
      fft512_wave<+1>(v, L, lane, twN, twB, lane);
    else
      fft512_wave_tw<+1>(v, L, lane, twA_g, twB_g, lane);
    if (c_out) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        c_out[(int64_t)y0 * N + T + 64 * t] = v[t].x;
        c_out[(int64_t)y1 * N + T + 64 * t] = v[t].y;
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = make_double2(fp2(v[j].x, a), fp2(v[j].y, a));
    if (HOLD)
      fft512_wave<-1>(v, L, T, twT, twB, lane);
    else
      fft512_wave_tw<-1>(v, L, T, twA_g, twB_g, lane);
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = T + 64 * t;
      L[k + (k >> 3)] = v[t];
    }
    wave_lds_sync();
    const int64_t g0 = spec_row_flat(a, y0), g1 = spec_row_flat(a, y1);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int k = lane + 64 * t;
      if (k <= N / 2) {
        const int km = (N - k) & (N - 1);
        const double2 w = L[k + (k >> 3)], mm = L[km + (km >> 3)];
        G[g0 + k] = make_double2(0.5 * (w.x + mm.x), 0.5 * (w.y - mm.y));
        G[g1 + k] = make_double2(0.5 * (w.y + mm.y), -0.5 * (w.x - mm.x));
      }
    }
    wave_lds_sync();
    pair = next;
  }
}

// The one-directional row passes (Poisson solve: real c -> forward x; ... -> inverse x -> real phi; initialisation: real
// field -> forward x of it or of f'(it)) in the same persistent, prefetching form.  Round 3 ran them on f2_row512_kernel<true>
// (one short-lived wave per row pair, twiddles re-read before every use, no load in flight during the transform): 1.56 ms
// of the 2.96 ms BM6 FD + Poisson step for 4.3 GB, 2.8 TB/s (profiles/r04/summary_bm6_fd_512c_before_row_kernels.json).
//   FWD: rows y0, y1 of a real field (8-byte loads, 512 B per wave instruction) -> [f'] -> forward x -> G rows
//   INV: H rows -> inverse x -> rows of a real field
template <bool FWD>
__global__ __launch_bounds__(64, 2) void f3_row512d_kernel(const F2Args a, const double2* __restrict__ H,
                                                           const double* __restrict__ r_in, double* __restrict__ r_out,
                                                           double2* __restrict__ G, const double2* __restrict__ twA_g,
                                                           const double2* __restrict__ twB_g, int npairs, int use_fprime) {
  __shared__ __attribute__((aligned(16))) double2 L[W8];
  constexpr int N = 512;
  const int lane = threadIdx.x;
  const int T = (lane >> 3) + 8 * (lane & 7);
  double2 twN[7], twB[7];
  load_tw(twN, twA_g, lane);
  load_tw(twB, twB_g, lane & 7);
  double2 p[8], q[8];  // INV: the two half-spectrum rows; FWD: p[j] = (row y0, row y1) at x = lane + 64 j (q unused)
  auto issue = [&](int pr) {
    if (FWD) {
      const double *r0 = r_in + (int64_t)(2 * pr) * N + lane, *r1 = r0 + N;
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = make_double2(r0[64 * j], r1[64 * j]);
    } else {
      const int64_t r0 = spec_row_flat(a, 2 * pr), r1 = spec_row_flat(a, 2 * pr + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = lane + 64 * j;
        const int kk = k > N / 2 ? N - k : k;
        p[j] = H[r0 + kk];
        q[j] = H[r1 + kk];
      }
    }
  };
  int pair = blockIdx.x;
  if (pair < npairs) issue(pair);
  while (pair < npairs) {
    double2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (FWD) {
        v[j] = use_fprime ? make_double2(fp2(p[j].x, a), fp2(p[j].y, a)) : p[j];
      } else {
        const bool upper = lane + 64 * j > N / 2;
        v[j] = upper ? make_double2(p[j].x + q[j].y, q[j].x - p[j].y) : make_double2(p[j].x - q[j].y, p[j].y + q[j].x);
      }
    }
    const int y0 = 2 * pair, y1 = y0 + 1;
    const int next = pair + gridDim.x;
    if (next < npairs) issue(next);
    fft512_wave<FWD ? -1 : +1>(v, L, lane, twN, twB, lane);
    if (FWD) {
      // Hermitian separation of the two real rows: needs X[k] and X[N - k] -> one more exchange (skewed: k + k / 8)
      wave_lds_sync();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = T + 64 * t;
        L[k + (k >> 3)] = v[t];
      }
      wave_lds_sync();
      const int64_t g0 = spec_row_flat(a, y0), g1 = spec_row_flat(a, y1);
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const int k = lane + 64 * t;
        if (k <= N / 2) {
          const int km = (N - k) & (N - 1);
          const double2 w = L[k + (k >> 3)], mm = L[km + (km >> 3)];
          G[g0 + k] = make_double2(0.5 * (w.x + mm.x), 0.5 * (w.y - mm.y));
          G[g1 + k] = make_double2(0.5 * (w.y + mm.y), -0.5 * (w.x - mm.x));
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        r_out[(int64_t)y0 * N + T + 64 * t] = v[t].x;
        r_out[(int64_t)y1 * N + T + 64 * t] = v[t].y;
      }
    }
    wave_lds_sync();
    pair = next;
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
template <int MODE, int CW3>
__global__ __launch_bounds__(64 * CW3, 4) void f3_col512_kernel(const F2Args a, double2* A,  // (A == H allowed)
                                                             double2* __restrict__ chat, double2* H,
                                                             const ColMap mn, const ColMap ms, int spon, int nblk,
                                                             int nitems, const double2* __restrict__ twA_g,
                                                             const double2* __restrict__ twB_g,
                                                             int* __restrict__ qset, int* __restrict__ qother) {
  __shared__ __attribute__((aligned(16))) double2 Lall[CW3 * W8C];
  __shared__ __attribute__((aligned(16))) double2 TWB[72];  // the 64-entry second twiddle table, rows skewed to 9 slots (no bank
                                                            // conflicts for the 8 distinct rows a wave reads): 7 of the 14
                                                            // twiddle loads of a transform leave the L1 path
  __shared__ int s_item;
  if (threadIdx.x < 64) TWB[(threadIdx.x >> 3) * 9 + (threadIdx.x & 7)] = twB_g[threadIdx.x];
  constexpr int N = 512, NT = 64 * CW3, PER = 8;
  auto nat = [](int n) { return n + (n >> 3); };
  // One work item (batch b, block of CW3 k_x columns) per workgroup.  (A persistent, software-pipelined form -- next
  // item's loads in flight during the transforms -- was measured and is slower: 3.85 vs 3.25 ms per step; its extra 32
  // VGPRs cost a wave per SIMD, and short-lived workgroups already overlap through the dispatcher.)
  // qset != null: persistent workgroups fed by the per-XCD queues above (items in natural order: neighbours in time and
  // L2); null: one item per workgroup, blockIdx -> item
  XcdQueue xq;
  int item = blockIdx.x;
  if (qset) {
    xq.init(qset, qother, nitems);
    item = xq.pop(&s_item);
  }
  while (item < nitems) {
    // the twiddle tables are re-read from L1 right before each use BY DESIGN (fft512_wave_tw); hide the pointers from the
    // optimiser once per item so that it cannot hoist those loads out of this loop into 56+ live VGPRs (= scratch)
    asm volatile("" : "+s"(twA_g), "+s"(twB_g));
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // same for the per-thread index arithmetic (cheap to redo, expensive to keep live)
    const int lane = tid & 63, wave = tid >> 6;
    double2* L = Lall + wave * W8C + 4 * wave;
    const int T = (lane >> 3) + 8 * (lane & 7);
    const int ci = tid % CW3;
    double2* Lc = Lall + ci * W8C + 4 * ci;
    double2 v[8], ch[PER];
    const int b = item / nblk, kx = (item % nblk) * CW3 + ci;
    const bool on = kx < a.nxh;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int r = (tid + NT * i) / CW3;
      const int64_t src = col_addr((MODE == 1 && spon == 2) ? ms : mn, b, r) + kx;
      v[i] = on ? A[src] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) Lc[nat((tid + NT * i) / CW3)] = v[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
    if (MODE == 1)
      fft512_wave_tw<+1, 9>(v, L, lane, twA_g, TWB, lane);
    else
      fft512_wave_tw<-1, 9>(v, L, lane, twA_g, TWB, lane);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
    if (MODE == 2) {  // the resident spectrum, fetched only now: 108 VGPRs (requesting it before the forward transform needs
                      // scratch at the 128-VGPR budget of two workgroups per CU and measured no better)
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        ch[i] = on ? chat[col_addr(mn, b, r) + kx] : make_double2(0.0, 0.0);
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
      fft512_wave_tw<+1, 9>(v, L, lane, twA_g, TWB, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        if (on) A[col_addr(mn, b, r) + kx] = Lc[nat(r)];
      }
    } else if (MODE == 0 || MODE == 1 || MODE == 3) {
      double2* dst = MODE == 3 ? chat : (spon ? H : A);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        const int64_t di = col_addr((MODE == 0 && spon == 1) ? ms : mn, b, r) + kx;
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
        if (on) chat[col_addr(mn, b, kz) + kx] = r;
        Lc[nat(kz)] = make_double2(r.x * a.inv_n, r.y * a.inv_n);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
      fft512_wave_tw<+1, 9>(v, L, lane, twA_g, TWB, lane);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int r = (tid + NT * i) / CW3;
        if (on) H[col_addr(mn, b, r) + kx] = Lc[nat(r)];
      }
    }
    if (!qset) break;
    item = xq.pop(&s_item);
  }
}

// Column passes for the other power-of-two axis lengths (128, 256, 1024): the multi-stage radix-2^2 LDS transform of the
// 2-D path (fft_inplace), G threads per column (one radix-4 group each: 64 up to 256 points, 256 for 1024 -- sized so
// that the kernels stay under 64 VGPRs and fill the CU: the first form, one wave per 1024-capable column, needed 162-178
// VGPRs), CWG adjacent k_x columns per workgroup moved with the column index fastest.  Same MODEs as f3_col512_kernel.
template <int MODE, int CWG, int G, int NMAX>  // G threads per column, axis length N <= NMAX
__global__ __launch_bounds__(G * CWG, 8) void f3_col_kernel(const F2Args a, double2* A,  // (A == H allowed)
                                                          double2* __restrict__ chat, double2* H,
                                                          const ColMap mn, const ColMap ms, int spon, int nblk, int nitems,
                                                          int N, int lg, const double2* __restrict__ tw_g) {
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
  for (int k = tid; k < N / 2; k += NT) TW[k] = tw_g[k];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int r = (tid + NT * i) / CWG;
    const int64_t src = col_addr((MODE == 1 && spon == 2) ? ms : mn, b, r) + kx;
    if (r < N) X[ci * NP + px(brev(r, lg))] = on ? A[src] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  if (MODE == 1)
    fft_inplace<+1, G, NMAX>(X + wave * NP, TW, N, lg, lane);
  else
    fft_inplace<-1, G, NMAX>(X + wave * NP, TW, N, lg, lane);
  if (MODE == 0 || MODE == 1 || MODE == 3) {
    double2* dst = MODE == 3 ? chat : (spon ? H : A);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int r = (tid + NT * i) / CWG;
      const int64_t di = col_addr((MODE == 0 && spon == 1) ? ms : mn, b, r) + kx;
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
        const double2 ch = on ? chat[col_addr(mn, b, kz) + kx] : make_double2(0.0, 0.0);
        double2 r;
        r.x = fma(-num, gh.x, ch.x) * den;
        r.y = fma(-num, gh.y, ch.y) * den;
        if (on) chat[col_addr(mn, b, kz) + kx] = r;
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
    if (r < N && on) dst[col_addr(mn, b, r) + kx] = X[ci * NP + px(r)];
  }
}

// =====================================================================================================================
// MIXED-RADIX passes: axes whose length factors into 2, 3 and 5 (8 .. 1024 points) -- the reference's own lattices (100
// intervals -> 200 / 400 points, dolfin/bench1.py:21-23) and any other such box -- on hand-written transforms too, so that
// the library (rocFFT, fftplan.hip) is left with sizes that have a larger prime factor.  Same pass structure, same
// spectrum layout (rocFFT's D2Z), same k-space arithmetic as the power-of-two kernels above; the transform is a
// Stockham autosort FFT in LDS (two buffers per transform, natural order in and out, radix 4 / 2 / 3 / 5 stages, twiddles
// from a full-circle table staged in LDS), G threads per transform.
struct Radices {
  int n = 0;
  int r[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// X -> (result in the buffer X points at on return; Y is the other one).  Called by all G threads of the transform's
// group (t = index in the group); every stage ends with a workgroup barrier (the groups of a workgroup run in lockstep).
template <int SIGN, int G>
__device__ __forceinline__ void fft_stockham(double2*& X, double2*& Y, const double2* TW, int N, const Radices& rd, int t) {
  constexpr double S3 = 0.86602540378443864676372317075294;   // sin(pi/3)
  constexpr double C51 = 0.30901699437494742410229341718282;  // cos(2 pi/5)
  constexpr double C52 = -0.80901699437494742410229341718282; // cos(4 pi/5)
  constexpr double S51 = 0.95105651629515357211643933337938;  // sin(2 pi/5)
  constexpr double S52 = 0.58778525229247312916870595463907;  // sin(4 pi/5)
  auto tw = [&](int idx) {  // W_N^idx for the direction of this transform
    double2 w = TW[idx];
    if (SIGN > 0) w.y = -w.y;
    return w;
  };
  // i * SIGN * v  (forward: -i v ... the rotation every odd butterfly output needs)
  auto rot = [](double2 v) { return SIGN < 0 ? make_double2(v.y, -v.x) : make_double2(-v.y, v.x); };
  int Ns = 1;
  for (int st = 0; st < rd.n; ++st) {
    const int R = rd.r[st], M = N / R, step = N / (Ns * R);
    for (int j = t; j < M; j += G) {
      const int k = j % Ns, base = (j - k) * R + k;
      if (R == 4) {
        double2 a0 = X[j], a1 = X[j + M], a2 = X[j + 2 * M], a3 = X[j + 3 * M];
        if (Ns > 1) {
          a1 = cmul2(tw(k * step), a1);
          a2 = cmul2(tw(2 * k * step), a2);
          a3 = cmul2(tw(3 * k * step), a3);
        }
        const double2 s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = rot(csub(a1, a3));
        Y[base] = cadd(s0, s1);
        Y[base + Ns] = cadd(d0, d1);
        Y[base + 2 * Ns] = csub(s0, s1);
        Y[base + 3 * Ns] = csub(d0, d1);
      } else if (R == 2) {
        double2 a0 = X[j], a1 = X[j + M];
        if (Ns > 1) a1 = cmul2(tw(k * step), a1);
        Y[base] = cadd(a0, a1);
        Y[base + Ns] = csub(a0, a1);
      } else if (R == 3) {
        double2 a0 = X[j], a1 = X[j + M], a2 = X[j + 2 * M];
        if (Ns > 1) {
          a1 = cmul2(tw(k * step), a1);
          a2 = cmul2(tw(2 * k * step), a2);
        }
        const double2 sm = cadd(a1, a2), df = rot(csub(a1, a2));
        const double2 m = make_double2(a0.x - 0.5 * sm.x, a0.y - 0.5 * sm.y);
        Y[base] = cadd(a0, sm);
        Y[base + Ns] = make_double2(m.x + S3 * df.x, m.y + S3 * df.y);
        Y[base + 2 * Ns] = make_double2(m.x - S3 * df.x, m.y - S3 * df.y);
      } else {  // 5
        double2 a0 = X[j], a1 = X[j + M], a2 = X[j + 2 * M], a3 = X[j + 3 * M], a4 = X[j + 4 * M];
        if (Ns > 1) {
          a1 = cmul2(tw(k * step), a1);
          a2 = cmul2(tw(2 * k * step), a2);
          a3 = cmul2(tw(3 * k * step), a3);
          a4 = cmul2(tw(4 * k * step), a4);
        }
        const double2 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = rot(csub(a1, a4)), t4 = rot(csub(a2, a3));
        const double2 m1 = make_double2(a0.x + C51 * t1.x + C52 * t2.x, a0.y + C51 * t1.y + C52 * t2.y);
        const double2 m2 = make_double2(a0.x + C52 * t1.x + C51 * t2.x, a0.y + C52 * t1.y + C51 * t2.y);
        const double2 n1 = make_double2(S51 * t3.x + S52 * t4.x, S51 * t3.y + S52 * t4.y);
        const double2 n2 = make_double2(S52 * t3.x - S51 * t4.x, S52 * t3.y - S51 * t4.y);
        Y[base] = make_double2(a0.x + t1.x + t2.x, a0.y + t1.y + t2.y);
        Y[base + Ns] = cadd(m1, n1);
        Y[base + 2 * Ns] = cadd(m2, n2);
        Y[base + 3 * Ns] = csub(m2, n2);
        Y[base + 4 * Ns] = csub(m1, n1);
      }
    }
    __syncthreads();
    double2* sw = X;
    X = Y;
    Y = sw;
    Ns *= R;
  }
}

// Row pass (x): 256 threads per pair of rows; contract of f2_row_kernel.  twx_g: nx entries e^{-2 pi i k / nx}.
__global__ __launch_bounds__(RT) void mx_row_kernel(const F2Args a, const double2* H,  // (H == G allowed)
                                                    const double* __restrict__ c_in, double* __restrict__ c_out,
                                                    double2* G, const double2* __restrict__ twx_g,
                                                    int from_spectrum, int use_fprime, const Radices rx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.nx, lane = threadIdx.x;
  double2* X = reinterpret_cast<double2*>(smem_raw);
  double2* Y = X + N + 1;
  double2* TW = Y + N + 1;
  const int y0 = 2 * blockIdx.x, y1 = y0 + 1;
  for (int k = lane; k < N; k += RT) TW[k] = twx_g[k];
  constexpr int MAXP = 1024 / RT;
  double2 z[MAXP];
  if (from_spectrum) {
    for (int k = lane; k <= N / 2; k += RT) {
      const double2 p = H[spec_row_flat(a, y0) + k], q = H[spec_row_flat(a, y1) + k];
      X[k] = make_double2(p.x - q.y, p.y + q.x);                                // p + i q
      if (k > 0 && 2 * k < N) X[N - k] = make_double2(p.x + q.y, q.x - p.y);  // conj(p) + i conj(q)
    }
    __syncthreads();
    fft_stockham<+1, RT>(X, Y, TW, N, rx, lane);
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int x = lane + RT * i;
      if (x < N) {
        z[i] = X[x];
        if (c_out) {
          c_out[(int64_t)y0 * N + x] = z[i].x;
          c_out[(int64_t)y1 * N + x] = z[i].y;
        }
      }
    }
    if (from_spectrum == 2) return;
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
    if (x < N) X[x] = use_fprime ? make_double2(fp2(z[i].x, a), fp2(z[i].y, a)) : z[i];
  }
  __syncthreads();
  fft_stockham<-1, RT>(X, Y, TW, N, rx, lane);
  for (int k = lane; k <= N / 2; k += RT) {
    const double2 w = X[k], m = X[k == 0 ? 0 : N - k];
    G[spec_row_flat(a, y0) + k] = make_double2(0.5 * (w.x + m.x), 0.5 * (w.y - m.y));
    G[spec_row_flat(a, y1) + k] = make_double2(0.5 * (w.y + m.y), -0.5 * (w.x - m.x));
  }
}

// Column pass (y, or z of a 3-D box): MXC adjacent k_x columns per workgroup, MXG threads per column; MODEs and address
// maps of f3_col_kernel.  two_d: the column axis is y of a 2-D grid (the k-space arithmetic then has k_y = row, k_z = 0).
constexpr int MXC = 4, MXG = 64;
template <int MODE>
__global__ __launch_bounds__(MXC * MXG) void mx_col_kernel(const F2Args a, double2* A,  // (A == H allowed)
                                                            double2* __restrict__ chat, double2* H,
                                                            const ColMap mn, const ColMap ms, int spon, int nblk, int N,
                                                            const double2* __restrict__ tw_g, const Radices rd, int two_d,
                                                            int ext) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int NP = N + 1;
  double2* S = reinterpret_cast<double2*>(smem_raw);  // [MXC][2][NP]
  double2* TW = S + MXC * 2 * NP;
  constexpr int NT = MXC * MXG;
  const int tid = threadIdx.x, lane = tid % MXG, grp = tid / MXG, ci = tid % MXC;
  const int lb = blockIdx.x;
  const int b = lb / nblk, kx = (lb % nblk) * MXC + ci;
  const bool on = kx < a.nxh;
  for (int k = tid; k < N; k += NT) TW[k] = tw_g[k];
  for (int idx = tid; idx < N * MXC; idx += NT) {
    const int r = idx / MXC;
    // ext: the column holds the N/2 + 1 physical nodes of a no-flux axis; the transform runs on its EVEN extension
    const int rs = (ext && 2 * r > N) ? N - r : r;
    const int64_t src = col_addr((MODE == 1 && spon == 2) ? ms : mn, b, rs) + kx;
    S[ci * 2 * NP + r] = on ? A[src] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  double2* X = S + grp * 2 * NP;
  double2* Y = X + NP;
  if (MODE == 1)
    fft_stockham<+1, MXG>(X, Y, TW, N, rd, lane);
  else
    fft_stockham<-1, MXG>(X, Y, TW, N, rd, lane);
  const int res = (int)(X - (S + grp * 2 * NP));  // 0 or NP: which buffer holds the result (the same for every group)
  if (MODE == 0 || MODE == 1 || MODE == 3) {
    double2* dst = MODE == 3 ? chat : (spon ? H : A);
    for (int idx = tid; idx < N * MXC; idx += NT) {
      const int r = idx / MXC;
      const int64_t di = col_addr((MODE == 0 && spon == 1) ? ms : mn, b, r) + kx;
      if (on && !(ext && 2 * r > N)) dst[di] = S[ci * 2 * NP + res + r];
    }
    return;
  }
  // k-space stage on the transformed columns; the result goes into the OTHER buffer (input of the inverse transform)
  const int oth = res ? 0 : NP;
  for (int idx = tid; idx < N * MXC; idx += NT) {
    const int r = idx / MXC;
    const double2 gh = S[ci * 2 * NP + res + r];
    double2 o;
    if (MODE == 2) {
      const int ky_i = two_d ? r : b + a.yoff, kz_i = two_d ? 0 : r;
      const int my = 2 * ky_i > a.ny ? ky_i - a.ny : ky_i;
      const int mz = 2 * kz_i > a.nz ? kz_i - a.nz : kz_i;
      const double kxv = a.kx0 * kx, kyv = a.ky0 * my, kzv = a.kz0 * mz;
      const double k2 = (kxv * kxv + kyv * kyv) + kzv * kzv;  // same grouping as spectral.hip's ksq
      const double num = a.dtM * k2;
      const double den = 1.0 / fma(a.dtMkappa, k2 * k2, 1.0 + (k2 > 0.0 ? a.gam : 0.0));
      const int64_t ca = col_addr(mn, b, r) + kx;
      const double2 ch = on ? chat[ca] : make_double2(0.0, 0.0);
      double2 rr;
      rr.x = fma(-num, gh.x, ch.x) * den;
      rr.y = fma(-num, gh.y, ch.y) * den;
      if (on) chat[ca] = rr;
      o = make_double2(rr.x * a.inv_n, rr.y * a.inv_n);
    } else {  // MODE 4: Poisson, divide by the eigenvalue of the 5- / 7-point Laplacian (tables in `chat`, see f3_col512_kernel)
      const double* sym = reinterpret_cast<const double*>(chat);
      const double lam = two_d ? (sym[on ? kx : 0] + sym[a.nx + r]) * a.dtM
                               : ((sym[on ? kx : 0] + sym[a.nx + b]) + sym[a.nx + a.ny + r]) * a.dtM;
      const bool zero = kx == 0 && (two_d ? r == 0 : (b == 0 && r == 0));
      const double sc = zero ? 0.0 : (a.inv_n / lam) * a.dtMkappa;
      o = make_double2(gh.x * sc, gh.y * sc);
      if (ext) {
        // sine / cosine transforms on the physical nodes (fused_poisson_dirichlet): a complex element is TWO real
        // columns, k_x = 2 kx and 2 kx + 1, each with its own eigenvalue; a.nxh counts the complex columns
        const int ka = 2 * kx, kb = 2 * kx + 1;
        const double rest = two_d ? sym[a.nx + r] : sym[a.nx + b] + sym[a.nx + a.ny + r];
        const double la = (sym[on ? ka : 0] + rest) * a.dtM, lbv = (sym[on && kb < a.nx ? kb : 0] + rest) * a.dtM;
        o = make_double2(la != 0.0 ? gh.x * (a.inv_n / la) : 0.0, lbv != 0.0 ? gh.y * (a.inv_n / lbv) : 0.0);
      }
    }
    S[ci * 2 * NP + oth + r] = o;
  }
  __syncthreads();
  X = S + grp * 2 * NP + oth;
  Y = S + grp * 2 * NP + res;
  fft_stockham<+1, MXG>(X, Y, TW, N, rd, lane);
  const int res2 = (int)(X - (S + grp * 2 * NP));
  double2* dst = MODE == 2 ? H : A;
  for (int idx = tid; idx < N * MXC; idx += NT) {
    const int r = idx / MXC;
    if (on && !(ext && 2 * r > N)) dst[col_addr(mn, b, r) + kx] = S[ci * 2 * NP + res2 + r];
  }
}

// ---- Poisson problem of the reference's BM6: phi = 0 on x = 0, phi = sin(y / 7) on x = Lx, no flux elsewhere ------------
// (dolfin/bench6.py:77-90, pfbase.py:410-421) as a sine transform along x and cosine transforms along y (and z) ON THE
// PHYSICAL NODES: the odd / even extensions exist only inside LDS, HBM sees the (Nx + 1)(Ny + 1)(Nz + 1) node values once
// per pass -- a quarter (2-D) or an eighth (3-D) of the lattice the round-2 solver transformed, and the right-hand side,
// the eigenvalue division and the Dirichlet values are fused into the passes (three library calls and three pointwise
// kernels before).  S: real array [zp][yp][kx], kx = 0 .. Nx (+ pad to an even count), read as complex by the cosine passes.
struct DirGeom {
  int npx, npy, npz;   // physical nodes per axis (lattice n = 2 (np - 1)); npz = 1 in 2-D
  int spitch;          // doubles per row of S (even)
  double h, k_over_eps, inv_h2;
};
__device__ __forceinline__ double dir_right(double y) { return sin(y / 7.0); }  // bench6.py:84 phi_right

// rows (yp, zp) in pairs: odd extension of the right-hand side -> forward FFT -> S = its (imaginary) sine coefficients
__global__ __launch_bounds__(RT) void dst_row_fwd_kernel(const F2Args a, const DirGeom g, const double* __restrict__ c,
                                                         double* __restrict__ S, const double2* __restrict__ twx_g,
                                                         const Radices rx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.nx, Np = N / 2, lane = threadIdx.x;
  double2* X = reinterpret_cast<double2*>(smem_raw);
  double2* Y = X + N + 1;
  double2* TW = Y + N + 1;
  for (int k = lane; k < N; k += RT) TW[k] = twx_g[k];
  const int nrows = g.npy * g.npz, r0 = 2 * blockIdx.x, r1 = r0 + 1;
  const bool two = r1 < nrows;
  const int y0 = r0 % g.npy, z0 = r0 / g.npy, y1 = two ? r1 % g.npy : 0, z1 = two ? r1 / g.npy : 0;
  const double* c0 = c + ((int64_t)z0 * a.ny + y0) * N;   // physical nodes are the first np lattice points of each axis
  const double* c1 = c + ((int64_t)z1 * a.ny + y1) * N;
  const double g0 = dir_right(y0 * g.h) * g.inv_h2, g1 = dir_right(y1 * g.h) * g.inv_h2;
  for (int x = lane; x < N; x += RT) {
    const int xr = x <= Np ? x : N - x;
    double va = 0.0, vb = 0.0;
    if (xr != 0 && xr != Np) {
      va = -g.k_over_eps * c0[xr];
      vb = two ? -g.k_over_eps * c1[xr] : 0.0;
      if (xr == Np - 1) {  // the Dirichlet value of x = Lx moved to the right-hand side
        va -= g0;
        if (two) vb -= g1;
      }
      if (x > Np) {
        va = -va;
        vb = -vb;
      }
    }
    X[x] = make_double2(va, vb);
  }
  __syncthreads();
  fft_stockham<-1, RT>(X, Y, TW, N, rx, lane);
  for (int k = lane; k <= Np; k += RT) {
    const double2 w = X[k], m = X[k == 0 ? 0 : N - k];
    S[(int64_t)r0 * g.spitch + k] = 0.5 * (w.y - m.y);             // Im of the first row's transform
    if (two) S[(int64_t)r1 * g.spitch + k] = -0.5 * (w.x - m.x);   // Im of the second row's
  }
}

// S (solution's sine coefficients along x, already back in real space along y, z) -> phi on the whole lattice: even in x
// with the Dirichlet values on x = 0, Lx (the Cahn-Hilliard kernel needs mu mirror symmetric), even in y and z
__global__ __launch_bounds__(RT) void dst_row_inv_kernel(const F2Args a, const DirGeom g, const double* __restrict__ S,
                                                         double* __restrict__ phi, const double2* __restrict__ twx_g,
                                                         const Radices rx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int N = a.nx, Np = N / 2, lane = threadIdx.x;
  double2* X = reinterpret_cast<double2*>(smem_raw);
  double2* Y = X + N + 1;
  double2* TW = Y + N + 1;
  for (int k = lane; k < N; k += RT) TW[k] = twx_g[k];
  const int nrows = g.npy * g.npz, r0 = 2 * blockIdx.x, r1 = r0 + 1;
  const bool two = r1 < nrows;
  for (int k = lane; k <= Np; k += RT) {
    const double sa = S[(int64_t)r0 * g.spitch + k], sb = two ? S[(int64_t)r1 * g.spitch + k] : 0.0;
    X[k] = make_double2(-sb, sa);                            // i sa + i (i sb)
    if (k > 0 && k < Np) X[N - k] = make_double2(sb, -sa);   // conj(i sa) + i conj(i sb)
  }
  __syncthreads();
  fft_stockham<+1, RT>(X, Y, TW, N, rx, lane);
  for (int rr = 0; rr < (two ? 2 : 1); ++rr) {
    const int r = r0 + rr, yp = r % g.npy, zp = r / g.npy;
    const double right = dir_right(yp * g.h);
    const int ym = (yp > 0 && yp < g.npy - 1) ? a.ny - yp : -1;   // mirror images on the lattice (none for the wall nodes)
    const int zm = (g.npz > 1 && zp > 0 && zp < g.npz - 1) ? a.nz - zp : -1;
    for (int x = lane; x <= Np; x += RT) {
      const double2 zz = X[x];
      const double v = x == 0 ? 0.0 : (x == Np ? right : (rr ? zz.y : zz.x));
      const int xm = (x > 0 && x < Np) ? N - x : -1;
      for (int iz = 0; iz < 2; ++iz) {
        const int zl = iz ? zm : zp;
        if (zl < 0) continue;
        for (int iy = 0; iy < 2; ++iy) {
          const int yl = iy ? ym : yp;
          if (yl < 0) continue;
          double* row = phi + ((int64_t)zl * a.ny + yl) * N;
          row[x] = v;
          if (xm >= 0) row[xm] = v;
        }
      }
    }
  }
}

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

}  // namespace

// n = product of radices 4, 2, 3, 5 (false if n has another prime factor)
bool factor235(int n, Radices* out) {
  Radices r;
  while (n % 4 == 0 && r.n < 12) {
    r.r[r.n++] = 4;
    n /= 4;
  }
  for (int p : {2, 3, 5})
    while (n % p == 0 && r.n < 12) {
      r.r[r.n++] = p;
      n /= p;
    }
  if (out) *out = r;
  return n == 1;
}
bool smooth_axis(int n) { return n >= 8 && n <= 1024 && factor235(n, nullptr); }
bool pow2_axis(int n) { return ilog2(n) >= 7 && ilog2(n) <= 10; }

struct Fused2D {
  F2Args a;
  bool mixed = false;  // some axis is not a power of two in 128..1024: every pass by the mixed-radix kernels (mx_*)
  Radices rx, ry, rz;
  double2 *twfx = nullptr, *twfy = nullptr, *twfz = nullptr;  // full-circle tables e^{-2 pi i k / n}, n entries (mx_* kernels)
  size_t lds_mrow = 0;
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
  int* queues = nullptr;  // 2 sets x 8 per-XCD heads (XcdQueue), zero-initialised; launch n uses set n & 1
  // z-chunked groups of passes (run_chunked): planes per chunk (0 = whole box, one launch per pass) and the side streams
  // the chunks are dealt to
  int chunk = 0, nside = 0;
  int chunk_min_planes = 0;   // calls over fewer planes run whole (default policy; 0 when PFHIP_FFT3D_CHUNK forces the chunking)
  char desc[256] = {0};  // fused2d_describe (a plain array: run_chunked copies the struct per chunk)
  hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  mutable unsigned qepoch = 0;
  int ncu = 256;
};

// 0: not here (library transforms); 1: power-of-two kernels (every axis 128, 256, 512 or 1024); 2: mixed-radix kernels
// (every axis 8 .. 1024 with prime factors 2, 3, 5 only; rows are transformed in pairs: nx and ny even)
int fused_path(int dim, int nx, int ny, int nz) {
  if (dim == 3) {
    const char* e = getenv("PFHIP_SPECTRAL_3D");  // "rocfft" forces the library path (A/B comparison)
    if (e && std::string(e) == "rocfft") return 0;
    // measured per step against the rocFFT path (profiles/r02/spectral3d_ab_sizes.log, bench_spectral_sizes.log):
    // 128^3 0.080 ms vs 0.088; 256^3 0.375 vs 0.567; 512^3 2.64 vs 3.5; 1024^3 24.9 vs 49.3
    if (pow2_axis(nx) && pow2_axis(ny) && pow2_axis(nz)) return 1;
    if (smooth_axis(nx) && smooth_axis(ny) && smooth_axis(nz) && nx % 2 == 0 && ny % 2 == 0) return 2;
    return 0;
  }
  if (dim != 2) return 0;
  if (pow2_axis(nx) && pow2_axis(ny)) return 1;
  const char* e = getenv("PFHIP_SPECTRAL_MIXED");  // "0": sizes that are not powers of two stay on the library (A/B)
  if (e && e[0] == '0') return 0;
  return (smooth_axis(nx) && smooth_axis(ny) && nx % 2 == 0 && ny % 2 == 0) ? 2 : 0;
}
bool fused2d_supported(int dim, int nx, int ny, int nz) {
  if (dim == 3) {
    const char* e = getenv("PFHIP_SPECTRAL_MIXED");
    if (e && e[0] == '0' && fused_path(dim, nx, ny, nz) == 2) return false;
  }
  return fused_path(dim, nx, ny, nz) != 0;
}

// Row pitch (complex elements) of every half-spectrum array this file touches.  2-D: nx/2 + 1, rocFFT's D2Z layout (the
// 2 MiB problem is L2-resident).  512^3: 264 instead of 257.  With 257 a column workgroup's 8 adjacent columns (128 bytes
// per row) start 16 bytes further into a 128-byte line on every row, so 7 of 8 rows straddle two lines: rocprofv3
// measured FETCH_SIZE x2 = 2.02 GB for the 1.08 GB a y pass needs (1.875x = (7*2 + 1)/8 exactly) and 4.09 GB for the z
// pass's 2.16 GB, with the passes already at 5.4-5.9 TB/s of HBM traffic (profiles/r02/spectral_512c_before_pitch.md).
// A pitch that is a multiple of 8 makes every (row, column-block) exactly one line.
int fused_spectrum_pitch(int dim, int nx, int ny, int nz) {
  const int nxh = nx / 2 + 1;
  if (dim == 3 && fused2d_supported(dim, nx, ny, nz)) return (nxh + 7) / 8 * 8;
  return nxh;
}

// Layout of every half-spectrum array handed to fused2d_* / fused3d_poisson: [z][nyp][pitch], and the rows to allocate.
// 3-D boxes reserve ONE PAD ROW PER PLANE.  The z passes walk columns whose rows are one plane apart; with ny and the
// pitch both multiples of 8 the plane stride is a multiple of 1 KB (512 x 264 x 16 B = 0x210000), and strides that are
// multiples of 1 KB spread a column's 128-byte lines badly over the HBM channels: a pure two-stream copy with the z pass's
// access pattern runs at 4.45 TB/s on physically contiguous memory and 4.65 on a fragmented allocation -- the "placement
// lottery" of round 3 -- against 5.0 on either once the plane stride is 256 B, 768 B or one 4224-byte row longer
// (tools/pairstream_probe.hip, profiles/r04/pairstream_probe.log).  The row stride of the y passes (4224 B) has that form.
// The whole step, six fresh processes on a box whose allocations were all of the slow kind without the pad row
// (profiles/r04/spectral_512c_planepad_ab.log): 2.52 -> 2.31 ms whole-box, 2.32 -> 2.10 chunked in 4-5 processes of 6 (the
// others stay slow: a second, physical-region effect of about 1.5 % on pure copies and up to 9 % on these latency-bound
// passes remains, profiles/r04/mempattern_probe_64_buffers.log); on a box of the fast kind the pad costs 0.7 %.
SpecLayout fused_spectrum_layout(int dim, int nx, int ny, int nz) {
  SpecLayout L;
  L.pitch = fused_spectrum_pitch(dim, nx, ny, nz);
  L.nyp = ny;
  if (dim == 3 && fused2d_supported(dim, nx, ny, nz)) {
    L.nyp = ny + 1;
  }
  L.rows = dim == 3 ? (int64_t)nz * L.nyp : (int64_t)ny;
  return L;
}

int fused2d_create(Fused2D** out, int nx, int ny, int nz, double h, hipStream_t stream, int want_mx) {
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
  a.nyp = fused_spectrum_layout(nz > 1 ? 3 : 2, nx, ny, nz).nyp;
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
    hipError_t e = pf_malloc(dev, sizeof(double2) * t.size());
    if (e != hipSuccess) return e;
    return hipMemcpy(*dev, t.data(), sizeof(double2) * t.size(), hipMemcpyHostToDevice);
  };
  if (table(nx, &f->twx) != hipSuccess || table(ny, &f->twy) != hipSuccess) return -3;
  if (nz > 1) {
    f->lgz = ilog2(nz);
    if (table(nz, &f->twz) != hipSuccess) return -3;
  }
  f->mixed = fused_path(nz > 1 ? 3 : 2, nx, ny, nz) == 2;
  if (f->mixed || nz == 1 || want_mx) {  // (2-D power-of-two grids too: their Poisson solve runs on the mixed-radix column kernel)
    auto full = [&](int N, double2** dev) -> hipError_t {
      std::vector<double2> t(N);
      for (int k = 0; k < N; ++k) {
        const double ang = TWO_PI_F * k / N;
        t[k] = make_double2(std::cos(ang), -std::sin(ang));
      }
      hipError_t e = pf_malloc(dev, sizeof(double2) * t.size());
      if (e != hipSuccess) return e;
      return hipMemcpy(*dev, t.data(), sizeof(double2) * t.size(), hipMemcpyHostToDevice);
    };
    if (!factor235(nx, &f->rx) || !factor235(ny, &f->ry) || (nz > 1 && !factor235(nz, &f->rz))) return -3;
    if (full(nx, &f->twfx) != hipSuccess || full(ny, &f->twfy) != hipSuccess ||
        (nz > 1 && full(nz, &f->twfz) != hipSuccess))
      return -3;
    f->lds_mrow = sizeof(double2) * (size_t)(2 * (nx + 1) + nx);
    const int nmax = ny > nz ? ny : nz;
    const size_t lds_mcol = sizeof(double2) * (size_t)(MXC * 2 * (nmax + 1) + nmax);
    auto big = [&](const void* k, size_t bytes) {
      return bytes <= 64 * 1024 ||
             hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    };
    if (!big(reinterpret_cast<const void*>(mx_row_kernel), f->lds_mrow) ||
        !big(reinterpret_cast<const void*>(dst_row_fwd_kernel), f->lds_mrow) ||
        !big(reinterpret_cast<const void*>(dst_row_inv_kernel), f->lds_mrow) ||
        !big(reinterpret_cast<const void*>(mx_col_kernel<0>), lds_mcol) ||
        !big(reinterpret_cast<const void*>(mx_col_kernel<1>), lds_mcol) ||
        !big(reinterpret_cast<const void*>(mx_col_kernel<2>), lds_mcol) ||
        !big(reinterpret_cast<const void*>(mx_col_kernel<3>), lds_mcol) ||
        !big(reinterpret_cast<const void*>(mx_col_kernel<4>), lds_mcol))
      return -3;
  }
  // 512-point axes: the one-wave radix-8 kernels (the multi-wave radix-2^2 kernels at 512 points were 8 % slower in 2-D,
  // profiles/r02; their A/B switch went in round 4)
  f->row512 = nx == 512 && (ny / 2) % RW == 0;
  f->col512 = ny == 512 || nz == 512;  // (3-D: "some column pass needs the radix-8 tables")
  if (f->mixed) f->row512 = f->col512 = false;
  if (f->cube512) {
    if (pf_malloc(&f->queues, sizeof(int) * 2 * 8 * QSTRIDE) != hipSuccess ||
        hipMemsetAsync(f->queues, 0, sizeof(int) * 2 * 8 * QSTRIDE, stream) != hipSuccess)   // (on the handle's stream: a
                                                // default-stream hipMemset is asynchronous and does not order against it)
      return -3;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      f->ncu = prop.multiProcessorCount;
  }
  if (f->cube512) {
    // Passes that only couple points of one z-plane (x rows, y columns) run chunk of planes by chunk of planes, one pass
    // after the other on the same chunk (run_chunked): a chunk that fits the 256 MiB Infinity Cache is still on the die
    // when the next pass reads it.  Default (where chunking is on): chunks of ~64 MiB of half spectrum dealt to two streams
    // (the tail of one chunk's launch overlaps the head of the next chunk's).
    // PFHIP_FFT3D_CHUNK="planes[,streams]" overrides (0 = whole box per launch): measured at 512^3 in one process,
    // profiles/r04/spectral_512c_chunk_ab.log.
    const double plane_mb = (double)ny * a.pitch * sizeof(double2) / (1024.0 * 1024.0);
    int ns = 2;
    // ... by default only where it was measured to pay: 512 x 512 planes (the one-wave radix-8 kernels, which run near the
    // copy rate of their access patterns, so the bytes they move are what counts) and a pass over at least 768 MiB of half
    // spectrum (run_chunked looks at the planes of the call: a rank's 64 planes of a 512^3 box are not chunked).
    // Measured in one process (profiles/r04/spectral_chunk_sizes.log): 512^3 2.30-2.54 -> 2.05-2.12 ms; 512 x 512 x 256
    // 1.138 -> 1.151, x 128 0.554 -> 0.598 (most of such a box is on the die between whole-box passes anyway, the chunk
    // launches only add tails); 256^3 0.341 -> 0.398; 1024^3 25.7 -> 26.6 (radix-2^2 kernels: LDS-bound, not memory-bound).
    if (nx == 512 && ny == 512 && f->row512 && f->col512) {
      f->chunk = (int)(64.0 / plane_mb + 0.5);
      if (f->chunk < 2) f->chunk = 2;
      f->chunk_min_planes = (int)(768.0 / plane_mb);
    }
    if (const char* e = getenv("PFHIP_FFT3D_CHUNK")) {
      int c = 0, n2 = ns;
      const int got = sscanf(e, "%d,%d", &c, &n2);
      if (got >= 1 && c >= 0) {
        f->chunk = c;
        f->chunk_min_planes = 0;   // forced: every call with more planes than a chunk
      }
      if (got >= 2 && n2 >= 0) ns = n2;
    }
    if (f->chunk >= nz) f->chunk = 0;
    if (f->chunk > 0 && ns > 1) {
      // lanes of the chunked groups: the handle's stream + (ns - 1) side streams that REALLY run beside it
      if (hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming) != hipSuccess) return -3;
      f->nside = pf_acquire_side_streams(stream, (ns > 4 ? 4 : ns) - 1, f->side);
      for (int k = 0; k < f->nside; ++k)
        if (hipEventCreateWithFlags(&f->ev_join[k], hipEventDisableTiming) != hipSuccess) return -3;
    }
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
    if (pf_malloc(&f->tw8a, sizeof(double2) * 512) != hipSuccess ||
        pf_malloc(&f->tw8b, sizeof(double2) * 64) != hipSuccess ||
        hipMemcpy(f->tw8a, ta.data(), sizeof(double2) * 512, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->tw8b, tb.data(), sizeof(double2) * 64, hipMemcpyHostToDevice) != hipSuccess)
      return -3;
  }
  {
    auto kind = [&](int n, bool is512) -> std::string {
      if (f->mixed) return "mixed-radix Stockham (" + std::to_string(n) + ")";
      return is512 ? "one-wave radix-8 (512)" : "radix-2^2 LDS (" + std::to_string(n) + ")";
    };
    std::string d = "hand-written LDS-FFT passes: x " + kind(nx, f->row512) + ", y " + kind(ny, ny == 512 && f->col512);
    if (nz > 1) {
      d += ", z " + kind(nz, nz == 512 && f->col512);
      d += f->chunk > 0 ? "; plane-local passes in chunks of " + std::to_string(f->chunk) + " planes on " +
                              std::to_string(f->nside + 1) + " stream(s)" +
                              (f->chunk_min_planes > 0 ? " (calls over >= " + std::to_string(f->chunk_min_planes) + " planes)" : "")
                        : "; whole box per launch";
    }
    snprintf(f->desc, sizeof f->desc, "%s", d.c_str());
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
    if (!big(reinterpret_cast<const void*>(f3_col_kernel<0, 4, 256, 1024>)) ||
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
  if (f->twz) (void)pf_free(f->twz);
  if (f->twfx) (void)pf_free(f->twfx);
  if (f->twfy) (void)pf_free(f->twfy);
  if (f->twfz) (void)pf_free(f->twfz);
  if (f->twx) (void)pf_free(f->twx);
  if (f->twy) (void)pf_free(f->twy);
  if (f->tw8a) (void)pf_free(f->tw8a);
  if (f->tw8b) (void)pf_free(f->tw8b);
  if (f->sym) (void)pf_free(f->sym);
  if (f->queues) (void)pf_free(f->queues);
  for (int k = 0; k < 4; ++k) {
    if (f->side[k]) (void)hipStreamDestroy(f->side[k]);
    if (f->ev_join[k]) (void)hipEventDestroy(f->ev_join[k]);
  }
  if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
  delete f;
}

void fused2d_invalidate(Fused2D* f) { f->g_valid = false; }
const char* fused2d_describe(const Fused2D* f) { return f->desc; }

namespace {
void launch_row(const Fused2D* f, const F2Args& a, const double2* H, const double* c_in, double* c_out, double2* G,
                int from_spectrum, int use_fprime) {
  if (f->mixed)
    hipLaunchKernelGGL(mx_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_mrow, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->twfx, from_spectrum, use_fprime, f->rx);
  else if (f->row512)
    hipLaunchKernelGGL(f2_row512_kernel<false>, dim3(a.ny / 2 / RW), dim3(64 * RW), 0, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->tw8a, (const double2*)f->tw8b, from_spectrum, use_fprime);
  else
    hipLaunchKernelGGL(f2_row_kernel, dim3(a.ny / 2), dim3(RT), f->lds_row, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->twx, from_spectrum, use_fprime);
}
// 2-D column pass by the mixed-radix kernel: one batch, rows one pitch apart
template <int MODE>
void launch_mx_col2d(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H) {
  ColMap m;
  m.rstride = a.pitch;
  const int nblk = (a.nxh + MXC - 1) / MXC;
  const size_t lds = sizeof(double2) * (size_t)(MXC * 2 * (a.ny + 1) + a.ny);
  hipLaunchKernelGGL(mx_col_kernel<MODE>, dim3(nblk), dim3(MXC * MXG), lds, f->stream, a, A, chat, H, m, ColMap(), 0, nblk,
                     a.ny, (const double2*)f->twfy, f->ry, 1, 0);
}
void launch_col(const Fused2D* f, const F2Args& a, const double2* G, double2* chat, double2* H, int init_only) {
  if (f->mixed) {
    if (init_only)
      launch_mx_col2d<3>(f, a, const_cast<double2*>(G), chat, nullptr);
    else
      launch_mx_col2d<2>(f, a, const_cast<double2*>(G), chat, H);
    return;
  }
  if (f->col512)  // one column per workgroup: 13.05 us per 512^2 step (2 columns: 14.3, 4: 17.5 -- parallelism beats line sharing)
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
  if (f->mixed) {
    hipLaunchKernelGGL(mx_row_kernel, dim3(a.ny * a.nz / 2), dim3(RT), f->lds_mrow, f->stream, a, H, c_in, c_out, G,
                       (const double2*)f->twfx, from_spectrum, use_fprime, f->rx);
    return;
  }
  if (f->row512 && from_spectrum == 1 && use_fprime) {
    // the step's row pass: persistent one-wave workgroups, 8 per CU (2 waves per SIMD: twiddles held in registers), the next
    // row pair's loads in flight during the two transforms.  H == G is allowed (in place: a row pair has one owner).
    const int npairs = a.ny * a.nz / 2;
    const int grid = f->ncu * 8 < npairs ? f->ncu * 8 : npairs;
    hipLaunchKernelGGL(f3_row512p_kernel<2>, dim3(grid), dim3(64), 0, f->stream, a, H, c_out, G, (const double2*)f->tw8a,
                       (const double2*)f->tw8b, npairs);
    return;
  }
  if (f->row512 && (from_spectrum == 0 || from_spectrum == 2)) {
    // the one-directional row passes (real field -> forward x; inverse x -> real field): same persistent form
    const int npairs = a.ny * a.nz / 2;
    const int grid = f->ncu * 8 < npairs ? f->ncu * 8 : npairs;
    if (from_spectrum == 0)
      hipLaunchKernelGGL(f3_row512d_kernel<true>, dim3(grid), dim3(64), 0, f->stream, a, H, c_in, c_out, G,
                         (const double2*)f->tw8a, (const double2*)f->tw8b, npairs, use_fprime);
    else
      hipLaunchKernelGGL(f3_row512d_kernel<false>, dim3(grid), dim3(64), 0, f->stream, a, H, c_in, c_out, G,
                         (const double2*)f->tw8a, (const double2*)f->tw8b, npairs, use_fprime);
    return;
  }
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
  ColMap mn, ms;  // natural side / the other side of the slab-decomposed passes (spon)
  int spon = 0;
  int nbatch;
  const double2* twf = nullptr;  // mixed-radix path: full-circle table and radices of this axis
  Radices rd;
};
ColGeom axis_geom(const Fused2D* f, const F2Args& a, int axis) {
  const int64_t row = a.pitch;
  ColGeom g;
  // axis 1: columns along y, one batch per z-plane; axis 2: columns along z, one batch per y-row 
  g.N = axis == 1 ? a.ny : a.nz;
  g.lg = axis == 1 ? a.lgy : f->lgz;
  g.tw = axis == 1 ? f->twy : f->twz;
  g.nbatch = axis == 1 ? a.nz : a.ny;
  g.twf = axis == 1 ? f->twfy : f->twfz;
  g.rd = axis == 1 ? f->ry : f->rz;
  const int64_t plane = spec_plane(a);
  ColMap m;
  if (axis == 1) {  // r = y, b = z
    m.rstride = row;
    m.bstride = plane;
  } else {  // r = z, b = y
    m.rstride = plane;
    m.bstride = row;
  }
  g.mn = m;
  return g;
}
template <int MODE, int CWG, int G, int NMAX>
void launch_col3_g(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int nblk = (a.nxh + CWG - 1) / CWG;
  const size_t lds = sizeof(double2) * ((size_t)CWG * (g.N + g.N / 32 + 1) + g.N / 2);
  hipLaunchKernelGGL((f3_col_kernel<MODE, CWG, G, NMAX>), dim3(nblk * g.nbatch), dim3(G * CWG), lds, f->stream, a, A, chat,
                     H, g.mn, g.ms, g.spon, nblk, nblk * g.nbatch, g.N, g.lg, g.tw);
}
template <int MODE, int CW3>
void launch_col3_t(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int nblk = (a.nxh + CW3 - 1) / CW3;
  const int nitems = nblk * g.nbatch;
  // z-type passes (two transforms and / or the resident spectrum per item): persistent workgroups, two per CU (what the CU
  // holds at 4 waves per SIMD and 74 KB of LDS), fed by the per-XCD queues -- 0.19 ms per 512^3 step faster than the static
  // map (profiles/r03/spectral_512c_alignment_and_queue.log); the y passes: one item per workgroup (queues measured slower)
  const bool queued = f->queues && (MODE == 2 || MODE == 3 || MODE == 4);
  int* qset = nullptr;
  int* qother = nullptr;
  int grid = nitems;
  if (queued) {
    const unsigned e = f->qepoch++;
    qset = f->queues + (e & 1) * 8 * QSTRIDE;
    qother = f->queues + ((e + 1) & 1) * 8 * QSTRIDE;
    grid = f->ncu * 2 < nitems ? f->ncu * 2 : nitems;
  }
  hipLaunchKernelGGL((f3_col512_kernel<MODE, CW3>), dim3(grid), dim3(64 * CW3), 0, f->stream, a, A, chat, H, g.mn, g.ms,
                     g.spon, nblk, nitems, (const double2*)f->tw8a, (const double2*)f->tw8b, qset, qother);
}
template <int MODE>
void launch_col3_geom(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, const ColGeom& g) {
  const int N = g.N;
  if (f->mixed) {
    const int nblk = (a.nxh + MXC - 1) / MXC;
    const size_t lds = sizeof(double2) * (size_t)(MXC * 2 * (N + 1) + N);
    hipLaunchKernelGGL(mx_col_kernel<MODE>, dim3(nblk * g.nbatch), dim3(MXC * MXG), lds, f->stream, a, A, chat, H, g.mn, g.ms,
                       g.spon, nblk, N, g.twf, g.rd, 0, 0);
    return;
  }
  if (N != 512) {
    // radix-2^2 LDS transforms: 8 columns per workgroup (256^3: 0.363 ms per step; 4 columns: 0.379), 4 columns of 1024
    // points (76 KB of LDS, 1024 threads)
    if (N > 512)
      launch_col3_g<MODE, 4, 256, 1024>(f, a, A, chat, H, g);
    else
      launch_col3_g<MODE, 8, 64, 256>(f, a, A, chat, H, g);
    return;
  }
  // 512 points: one wave per column, 8 columns per workgroup = one full 128-byte line per row on every pass (4-column
  // workgroups fetched half lines 1.57x in round 2 and were slower on every box of round 3)
  launch_col3_t<MODE, 8>(f, a, A, chat, H, g);
}
template <int MODE>
void launch_col3(const Fused2D* f, const F2Args& a, double2* A, double2* chat, double2* H, int axis) {
  launch_col3_geom<MODE>(f, a, A, chat, H, axis_geom(f, a, axis));
}

// A group of passes that only couple points of one z-plane (x rows, y columns), chunk of planes by chunk of planes:
// body(fc, ac, z0) launches the whole group on planes [z0, z0 + ac.nz) with the launch descriptor fc (its stream: chunks
// are dealt round-robin to the side streams, which fork from and join f->stream, so the group is one node in the handle's
// stream order).  A chunk that fits the 256 MiB Infinity Cache is still on the die when the group's next pass reads it,
// and an intermediate that the next pass overwrites in place never reaches HBM: at 512^3 the three middle passes of a
// spectral step cost 6 array moves of 1.08 GB whole-box and about 2 chunked (mempattern_probe: 1275 -> 980 us for pure
// copies with these access patterns; the step: 2.30-2.54 -> 2.05 ms, profiles/r04/spectral_512c_chunk_ab.log).
template <class Body>
int run_chunked(const Fused2D* f, const F2Args& a, Body body) {
  if (!(f->chunk > 0 && f->chunk < a.nz && a.nz >= f->chunk_min_planes)) {
    body(*f, a, 0);
    return 0;
  }
  if (f->nside > 0) {
    if (hipEventRecord(f->ev_fork, f->stream) != hipSuccess) return -3;
    for (int k = 0; k < f->nside; ++k)
      if (hipStreamWaitEvent(f->side[k], f->ev_fork, 0) != hipSuccess) return -3;
  }
  // equal chunks (the remainder spread one plane each): 512 planes at 31 -> 17 chunks of 31 / 30; a rank's 64 planes -> 22, 21, 21
  const int nchunks = (a.nz + f->chunk - 1) / f->chunk, base = a.nz / nchunks, rem = a.nz % nchunks;
  for (int k = 0, z0 = 0; k < nchunks; ++k) {
    F2Args ac = a;
    ac.nz = base + (k < rem ? 1 : 0);
    Fused2D fc = *f;  // launch descriptor only: the launchers read geometry, tables and the stream from it
    const int lane = k % (f->nside + 1);   // lane 0 = the handle's stream, the others = the side streams tested at create
    if (lane > 0) fc.stream = f->side[lane - 1];
    body(fc, ac, z0);
    f->qepoch = fc.qepoch;
    z0 += ac.nz;
  }
  for (int j = 0; j < f->nside; ++j)
    if (hipEventRecord(f->ev_join[j], f->side[j]) != hipSuccess || hipStreamWaitEvent(f->stream, f->ev_join[j], 0) != hipSuccess)
      return -3;
  return 0;
}
}  // namespace

// 512^3 periodic Poisson solve lap_h(phi) = -(k/eps) c with the same passes: x forward, y forward, z (forward, divide
// by the Laplacian's eigenvalue, inverse), y inverse, x inverse only.  W: nh complex work array.
namespace {
// 2 cos(2 pi m / n) - 2 for the x, y and z axis back to back: the eigenvalues of the 5- / 7-point Laplacian (times h^2)
int ensure_sym(Fused2D* f) {
  if (f->sym) return 0;
  const int nn[3] = {f->a.nx, f->a.ny, f->a.nz};
  std::vector<double> t;
  for (int d = 0; d < 3; ++d)
    for (int m = 0; m < nn[d]; ++m) t.push_back(2.0 * std::cos(TWO_PI_F * m / nn[d]) - 2.0);
  if (pf_malloc(&f->sym, sizeof(double) * t.size()) != hipSuccess ||
      hipMemcpy(f->sym, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice) != hipSuccess)
    return -3;
  return 0;
}
}  // namespace

// Lattices (2 (np - 1) points per axis) the sine / cosine solver below can run on
bool fused_dirichlet_supported(int dim, int nx, int ny, int nz) {
  const char* e = getenv("PFHIP_POISSON_DIRICHLET");  // "rocfft": the round-2 solver (library transforms of the extension)
  if (e && std::string(e) == "rocfft") return false;
  return smooth_axis(nx) && smooth_axis(ny) && (dim == 2 || smooth_axis(nz)) && nx % 2 == 0 && ny % 2 == 0 &&
         (dim == 2 || nz % 2 == 0);
}
int64_t fused_dirichlet_work_doubles(int npx, int npy, int npz) { return (int64_t)((npx + 1) / 2 * 2) * npy * npz; }

// phi (whole lattice) <- solution of the reference's BM6 Poisson problem for c (lattice, even extension); S: work array
// of fused_dirichlet_work_doubles().  f: created for the LATTICE sizes with want_mx = 1.
int fused_poisson_dirichlet(Fused2D* f, const double* c, double* phi, double* S, int npx, int npy, int npz, double h,
                            double k_over_eps) {
  if (ensure_sym(f) != 0) return -3;
  const F2Args& a = f->a;
  DirGeom g;
  g.npx = npx;
  g.npy = npy;
  g.npz = npz;
  g.spitch = (npx + 1) / 2 * 2;
  g.h = h;
  g.k_over_eps = k_over_eps;
  g.inv_h2 = 1.0 / (h * h);
  const int nrows = npy * npz, grid = (nrows + 1) / 2;
  hipLaunchKernelGGL(dst_row_fwd_kernel, dim3(grid), dim3(RT), f->lds_mrow, f->stream, a, g, c, S, (const double2*)f->twfx, f->rx);
  F2Args ac = a;                 // the cosine passes read S as complex: one element = two real k_x columns
  const int spc = g.spitch / 2;
  ac.nxh = spc;
  ac.dtM = g.inv_h2;
  ac.dtMkappa = 1.0;
  double2* Sc = reinterpret_cast<double2*>(S);
  double2* sym = reinterpret_cast<double2*>(f->sym);
  const int nblk = (spc + MXC - 1) / MXC;
  auto lds = [](int N) { return sizeof(double2) * (size_t)(MXC * 2 * (N + 1) + N); };
  if (npz == 1) {
    ColMap m;
    m.rstride = spc;
    hipLaunchKernelGGL(mx_col_kernel<4>, dim3(nblk), dim3(MXC * MXG), lds(a.ny), f->stream, ac, Sc, sym, nullptr, m, ColMap(), 0,
                       nblk, a.ny, (const double2*)f->twfy, f->ry, 1, 1);
  } else {
    ColMap my, mz;
    my.rstride = spc;                     // y columns: r = yp, one batch per physical plane
    my.bstride = (int64_t)npy * spc;
    mz.rstride = (int64_t)npy * spc;      // z columns: r = zp, one batch per k_y
    mz.bstride = spc;
    hipLaunchKernelGGL(mx_col_kernel<0>, dim3(nblk * npz), dim3(MXC * MXG), lds(a.ny), f->stream, ac, Sc, nullptr, nullptr, my,
                       ColMap(), 0, nblk, a.ny, (const double2*)f->twfy, f->ry, 0, 1);
    hipLaunchKernelGGL(mx_col_kernel<4>, dim3(nblk * npy), dim3(MXC * MXG), lds(a.nz), f->stream, ac, Sc, sym, nullptr, mz,
                       ColMap(), 0, nblk, a.nz, (const double2*)f->twfz, f->rz, 0, 1);
    hipLaunchKernelGGL(mx_col_kernel<1>, dim3(nblk * npz), dim3(MXC * MXG), lds(a.ny), f->stream, ac, Sc, nullptr, nullptr, my,
                       ColMap(), 0, nblk, a.ny, (const double2*)f->twfy, f->ry, 0, 1);
  }
  hipLaunchKernelGGL(dst_row_inv_kernel, dim3(grid), dim3(RT), f->lds_mrow, f->stream, a, g, (const double*)S, phi,
                     (const double2*)f->twfx, f->rx);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int fused3d_poisson(Fused2D* f, const double* c, double* phi, double2* W, double k_over_eps, double inv_h2) {
  if (ensure_sym(f) != 0) return -3;
  F2Args a = f->a;
  a.dtM = inv_h2;
  a.dtMkappa = -k_over_eps;
  if (!f->cube512) {  // 2-D periodic box: x forward, y (forward, divide, inverse) by the mixed-radix column kernel, x inverse
    launch_row(f, a, nullptr, c, nullptr, W, 0, 0);
    launch_mx_col2d<4>(f, a, W, reinterpret_cast<double2*>(f->sym), nullptr);
    launch_row(f, a, W, nullptr, phi, nullptr, 2, 0);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  const int64_t plane = spec_plane(a), rplane = (int64_t)a.ny * a.nx;
  if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
        launch_row3(&fc, ac, nullptr, c + z0 * rplane, nullptr, W + z0 * plane, 0, 0);
        launch_col3<0>(&fc, ac, W + z0 * plane, nullptr, nullptr, 1);
      }) != 0)
    return -3;
  launch_col3<4>(f, a, W, reinterpret_cast<double2*>(f->sym), nullptr, 2);
  if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
        launch_col3<1>(&fc, ac, W + z0 * plane, nullptr, nullptr, 1);
        launch_row3(&fc, ac, W + z0 * plane, nullptr, phi + z0 * rplane, nullptr, 2, 0);
      }) != 0)
    return -3;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// chat <- 2-D spectrum of c (same as a rocFFT D2Z); G is clobbered
int fused2d_spectrum(Fused2D* f, const double* c, double2* chat, double2* G) {
  const F2Args& a = f->a;
  if (f->cube512) {
    const int64_t plane = spec_plane(a), rplane = (int64_t)a.ny * a.nx;
    if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
          launch_row3(&fc, ac, nullptr, c + z0 * rplane, nullptr, G + z0 * plane, 0, 0);
          launch_col3<0>(&fc, ac, G + z0 * plane, nullptr, nullptr, 1);
        }) != 0)
      return -3;
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
    // Two work arrays: G (y-transformed f'(c), consumed by the z pass) and H (z pass output -> y inverse in place -> read by
    // the x pass, which writes G again).  Measured against ONE array with every pass in place (one process, same
    // allocations, profiles/r04/spectral_512c_work_arrays_ab.log): the in-place x pass costs 0.03-0.06 ms per step, the
    // in-place z pass nothing -- the saved write-back of the dead intermediate does not pay for it.
    const int64_t plane = spec_plane(a), rplane = (int64_t)a.ny * a.nx;
    if (!f->g_valid) {  // x- and y-transform of f'(c_in) (first step, or after the field was replaced)
      if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
            launch_row3(&fc, ac, nullptr, c_in + z0 * rplane, nullptr, G + z0 * plane, 0, 1);
            launch_col3<0>(&fc, ac, G + z0 * plane, nullptr, nullptr, 1);
          }) != 0)
        return -3;
    }
    launch_col3<2>(f, a, G, chat, H, 2);  // z: forward, k-space update of chat, inverse -> H (whole box: a column spans all planes)
    // y inverse (in place on H) -> x (inverse -> c_out, f'(c_out), forward -> G) -> y forward (in place on G, ready for the
    // next step's z pass), chunk of planes by chunk of planes
    if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
          double2 *Gc = G + z0 * plane, *Hc = H + z0 * plane;
          launch_col3<1>(&fc, ac, Hc, nullptr, nullptr, 1);
          launch_row3(&fc, ac, Hc, nullptr, c_out ? c_out + z0 * rplane : nullptr, Gc, 1, 1);
          launch_col3<0>(&fc, ac, Gc, nullptr, nullptr, 1);
        }) != 0)
      return -3;
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
  return P >= 1 && ny % P == 0 && nz % P == 0 && ilog2(P) >= 0 && ilog2(ny / P) >= 0 && fused2d_supported(3, nx, ny, nz);
}
namespace {
ColGeom slab_y_geom(const Fused2D* f, const F2Args& a, int nzl, int P, int on) {
  ColGeom g = axis_geom(f, a, 1);  // one batch per plane of this launch (a.nz planes of the rank's nzl: the layout below
                                   // speaks of the whole local slab, the caller offsets the base pointers by the first plane)
  const int nyl = a.ny / P;
  g.spon = on;
  g.ms.lr = ilog2(nyl);            // element r of a column goes to chunk r / nyl of the all-to-all layout
  g.ms.rchunk = (int64_t)nzl * nyl * a.pitch;
  g.ms.rstride = a.pitch;
  g.ms.bstride = (int64_t)nyl * a.pitch;
  return g;
}
}  // namespace
// real planes -> x rows (of f'(c) if use_fprime) -> tmp [zl][y][kx] -> y columns -> A [q][zl][yq][kx] (all-to-all layout)
int fusedslab_forward_xy(Fused2D* f, const double* real_in, double2* tmp, double2* A, int nzl, int P, int use_fprime,
                         double ca, double cb, double two_rho) {
  F2Args a = f->a;
  a.nyp = a.ny;  // slab-decomposed passes: planes without pad rows
  a.nz = nzl;
  a.ca = ca;
  a.cb = cb;
  a.two_rho = two_rho;
  // both passes are plane-local: chunk of planes by chunk of planes (run_chunked), tmp stays on the die in between
  const int64_t plane = spec_plane(a), rplane = (int64_t)a.ny * a.nx, aplane = (int64_t)(a.ny / P) * a.pitch;
  if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
        launch_row3(&fc, ac, nullptr, real_in + z0 * rplane, nullptr, tmp + z0 * plane, 0, use_fprime);
        launch_col3_geom<0>(&fc, ac, tmp + z0 * plane, nullptr, A + z0 * aplane, slab_y_geom(&fc, ac, nzl, P, 1));
      }) != 0)
    return -3;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// A [q][zl][yq][kx] -> inverse y columns -> tmp [zl][y][kx] -> inverse x rows -> real planes
int fusedslab_inverse_yx(Fused2D* f, double2* A, double2* tmp, double* real_out, int nzl, int P) {
  F2Args a = f->a;
  a.nyp = a.ny;  // slab-decomposed passes: planes without pad rows
  a.nz = nzl;
  const int64_t plane = spec_plane(a), rplane = (int64_t)a.ny * a.nx, aplane = (int64_t)(a.ny / P) * a.pitch;
  if (run_chunked(f, a, [&](const Fused2D& fc, const F2Args& ac, int z0) {
        launch_col3_geom<1>(&fc, ac, A + z0 * aplane, nullptr, tmp + z0 * plane, slab_y_geom(&fc, ac, nzl, P, 2));
        launch_row3(&fc, ac, tmp + z0 * plane, nullptr, real_out + z0 * rplane, nullptr, 2, 0);
      }) != 0)
    return -3;
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// z columns of B (T layout, nyl local y-rows starting at global row yoff), in place.  mode 0: forward; 1: inverse
// (unnormalised); 2: forward -> k-space update of the resident chat -> inverse of chat / N; 3: forward, stored to chat.
int fusedslab_z(Fused2D* f, double2* B, double2* chat, int mode, int nyl, int yoff, double dtM, double dtMkappa) {
  F2Args a = f->a;
  a.nyp = a.ny;  // slab-decomposed passes: planes without pad rows
  a.yoff = yoff;
  a.dtM = dtM;
  a.dtMkappa = dtMkappa;
  a.gam = 0.0;
  ColGeom g;
  g.N = a.nz;
  g.lg = f->lgz;
  g.tw = f->twz;
  g.mn.rstride = (int64_t)nyl * a.pitch;
  g.mn.bstride = a.pitch;
  g.twf = f->twfz;
  g.rd = f->rz;
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
