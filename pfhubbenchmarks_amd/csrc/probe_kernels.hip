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

}  // namespace

hipError_t launch_clock_probe(double* out, int nblocks, int spin_us, int busy, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(nblocks), dim3(busy ? 256 : 64), 0, stream, out,
                     (unsigned)spin_us * 100u, busy);
  return hipGetLastError();
}

}  // namespace pfhip
