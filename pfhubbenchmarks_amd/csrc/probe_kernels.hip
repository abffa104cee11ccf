// Measurement probes (not on any product path): the shader clock the chip actually holds, read inside a kernel.
//   clock = delta s_memtime (shader cycles) / delta s_memrealtime (constant 100 MHz) -- MI355X_MICROARCH.md, "DVFS
//   give-back" item 6.  Used by tools/ramp_probe.py to find what the first launches after an idle period pay for.
#include "pfhip_internal.h"

namespace pfhip {
namespace {

// busy == 0: lane 0 of every workgroup sleeps between time checks (the chip sees an almost idle kernel);
// busy != 0: every lane runs a dependent fp64 fma chain between the checks (all SIMDs issuing, no memory traffic).
// The spin is bounded by `spin_ticks` of the 100 MHz counter AND by an iteration cap.
__global__ __launch_bounds__(256) void clock_probe_kernel(double* __restrict__ out, unsigned spin_ticks, int busy) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  double acc = (double)threadIdx.x;
  for (int it = 0; it < (1 << 22) && (unsigned)(r1 - r0) < spin_ticks; ++it) {
    if (busy) {
#pragma unroll
      for (int k = 0; k < 64; ++k) acc = __builtin_fma(acc, 1.0000001, 1e-9);
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[(size_t)blockIdx.x * 2 + 0] = r1 > r0 ? (double)(c1 - c0) / (double)(r1 - r0) * 100.0 : 0.0;  // MHz
    out[(size_t)blockIdx.x * 2 + 1] = (double)r0 * 1e-5;                                                // ms, free-running
  }
  if (acc == -1.0) out[0] = acc;  // keeps the chain alive
}

// Single-XCD barrier probe (round 3): what a persistent multi-step kernel would pay per phase boundary
// if ALL its workgroups sat on one XCD -- one L2, so a hand-off needs no L2 write-back, only loads that bypass the
// per-CU L1 (sc1) and an arrival counter.  The grid is launched chip-wide; every workgroup registers the XCD it landed
// on (HW_REG_XCC_ID), the ones that are not on XCD `target` leave at once, the others run
// `iters` rounds of { write a 128-byte record (plain stores) -> vmcnt(0) -> workgroup barrier -> arrive (agent-scope
// atomic add) -> poll the counter (sc1 loads) -> read the NEXT participant's record with sc1 loads and check it }.
// ctl: [0] registered workgroups, [32 + x] census of XCD x, [64] arrival counter, [96] stale records seen, [128] bounded
// spin tripped, [160] participants; out[0] = microseconds per round (participant 0, 100 MHz counter).
__global__ __launch_bounds__(256) void xcd_barrier_probe_kernel(unsigned* __restrict__ ctl, unsigned long long* __restrict__ rec,
                                                               double* __restrict__ out, int iters, int target, int handoff) {
  __shared__ int s_me, s_n, s_bad;
  int id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const int xcc = id & 7;
  if (threadIdx.x == 0) {
    s_bad = 0;
    const unsigned me = __hip_atomic_fetch_add(ctl + 32 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    s_me = (int)me;
    if (xcc == target) {  // wait until every workgroup of the grid has registered: the census of this XCD is then final
      int spins = 0;
      while (__hip_atomic_load(ctl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {
          s_bad = 1;
          break;
        }
      }
      s_n = (int)__hip_atomic_load(ctl + 32 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (xcc != target) return;
  const int me = s_me, n = s_n;
  int bad = s_bad;
  unsigned long long t0 = 0;
  if (threadIdx.x == 0 && me == 0) {
    ctl[160] = (unsigned)n;
    t0 = __builtin_amdgcn_s_memrealtime();
  }
  unsigned stale = 0;
  for (int it = 1; it <= iters && !bad; ++it) {
    if (handoff && threadIdx.x < 16) rec[(size_t)me * 16 + threadIdx.x] = ((unsigned long long)it << 32) | (unsigned)(me * 16 + threadIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctl + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)it * (unsigned)n;
      int spins = 0;
      while (__hip_atomic_load(ctl + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 24)) {
          s_bad = 1;
          break;
        }
      }
    }
    __syncthreads();
    bad = s_bad;
    if (handoff && threadIdx.x < 16) {
      const int nb = (me + 1) % n;
      const unsigned long long v =
          __hip_atomic_load(rec + (size_t)nb * 16 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: bypasses L1
      if (v != (((unsigned long long)it << 32) | (unsigned)(nb * 16 + threadIdx.x))) ++stale;
    }
    // second barrier of the round (the neighbour may not overwrite its record before it has been read): same cost again,
    // counted in the per-round figure
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctl + 65, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)it * (unsigned)n;
      int spins = 0;
      while (__hip_atomic_load(ctl + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1 << 24)) {
          s_bad = 1;
          break;
        }
      }
    }
    __syncthreads();
    bad = s_bad;
  }
  if (stale) atomicAdd(ctl + 96, stale);
  if (threadIdx.x == 0 && bad) ctl[128] = 1;
  if (threadIdx.x == 0 && me == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = (double)(t1 - t0) * 0.01 / (double)iters;  // us per round = two barriers + the hand-off
  }
}

}  // namespace

// us_per_round: one round = two single-XCD barriers (+ a 128-byte record hand-off per workgroup when handoff != 0)
hipError_t run_xcd_barrier_probe(int nblocks, int nthreads, int iters, int handoff, double* us_per_round, int* participants,
                                 int* stale, int* ok) {
  unsigned* ctl = nullptr;
  unsigned long long* rec = nullptr;
  double* out = nullptr;
  hipError_t e;
  if ((e = pf_malloc(&ctl, sizeof(unsigned) * 256)) != hipSuccess) return e;
  if ((e = pf_malloc(&rec, sizeof(unsigned long long) * 16 * (size_t)nblocks)) != hipSuccess) return e;
  if ((e = pf_malloc(&out, sizeof(double) * 8)) != hipSuccess) return e;
  unsigned h_ctl[256];
  double h_out = 0.0;
  for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {  // the second repetition is the reported one
    (void)hipMemset(ctl, 0, sizeof(unsigned) * 256);
    (void)hipMemset(rec, 0, sizeof(unsigned long long) * 16 * (size_t)nblocks);
    (void)hipMemset(out, 0, sizeof(double) * 8);
    hipLaunchKernelGGL(xcd_barrier_probe_kernel, dim3(nblocks), dim3(nthreads), 0, nullptr, ctl, rec, out, iters, 0, handoff);
    if ((e = hipGetLastError()) != hipSuccess) break;
    if ((e = hipDeviceSynchronize()) != hipSuccess) break;
    if ((e = hipMemcpy(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost)) != hipSuccess) break;
    e = hipMemcpy(&h_out, out, sizeof(double), hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess) {
    *us_per_round = h_out;
    *participants = (int)h_ctl[160];
    *stale = (int)h_ctl[96];
    *ok = h_ctl[128] == 0;
  }
  (void)pf_free(ctl);
  (void)pf_free(rec);
  (void)pf_free(out);
  return e;
}

hipError_t launch_clock_probe(double* out, int nblocks, int spin_us, int busy, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(nblocks), dim3(busy ? 256 : 64), 0, stream, out,
                     (unsigned)spin_us * 100u, busy);
  return hipGetLastError();
}

}  // namespace pfhip
