// One-wave 512-point FFT (three radix-8 passes in registers, two wave-private LDS exchanges) -- shared by the 2-D / 3-D
// spectral passes (spectral2d_fused.hip), the pipelined 512-point column passes (spectral3d_pipe.hip) and tools/.
//
//   n = l + 64 j          stage A: lane holds x[l + 64 j], j = 0..7 -> radix-8 over j -> y[l][q], times W_512^(l q)
//   l = l0 + 8 l1         stage B: lane (q, l0) gathers l1 = 0..7 -> radix-8 -> z[q][l0][s], times W_64^(l0 s)
//   k = q + 8 s + 64 t    stage C: lane (q, s) gathers l0 = 0..7 -> radix-8 -> X[q + 8 s + 64 t], t = 0..7
// Physical lane p = 8 q + s ends up holding X[T(p) + 64 t] with T(p) = q + 8 s (the two octal digits of p swapped).
#ifndef PFHIP_FFT512_WAVE_H
#define PFHIP_FFT512_WAVE_H
#include <hip/hip_runtime.h>

namespace pfhip {
namespace {

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
  for (int q = 0; q < 8; ++q) L[q * 72 + m + (m >> 3)] = v[q];
  wave_lds_sync();
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) v[l1] = L[hi * 72 + lo + 9 * l1];
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
template <int SIGN, int BS = 8>  // BS: row stride of the second table (8: the global table; 9: its bank-skewed copy in LDS)
__device__ __forceinline__ void fft512_wave_tw(double2 (&v)[8], double2* L, int m, const double2* __restrict__ twA_g,
                                               const double2* twB_g, int lane) {
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
  for (int q = 0; q < 8; ++q) L[q * 72 + m + (m >> 3)] = v[q];
  wave_lds_sync();
#pragma unroll
  for (int l1 = 0; l1 < 8; ++l1) v[l1] = L[hi * 72 + lo + 9 * l1];
  {
    double2 tw[7];
#pragma unroll
    for (int sx = 1; sx < 8; ++sx) tw[sx - 1] = twB_g[lo * BS + sx];
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

}  // namespace
}  // namespace pfhip
#endif
