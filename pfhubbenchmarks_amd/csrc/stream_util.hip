// Streams that do not share a hardware queue (stream_util.hip).
//
// ROCm multiplexes HIP streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4; tools/stream_queue_probe.hip shows
// which streams of a process share one).  Two streams on one hardware queue run their kernels one behind the other, so the
// chunk lanes of the spectral step (run_chunked) would not overlap: 2.55 instead of 2.25 ms per 512^3 step with
// GPU_MAX_HW_QUEUES=1, and from handle to handle whenever the runtime happens to put a side stream on the handle stream's
// queue (profiles/r04/stream_queues.log).  HIP has no call that names a hardware queue, but sharing can be TESTED: two
// 120 us spin kernels launched at once finish in ~140 us side by side and in ~260 us on one queue.  pf_acquire_stream
// creates streams until one runs beside every stream of `avoid` and beside the legacy default stream; the rejected ones are
// destroyed at the end, so that the runtime does not hand the same queue out again.  When none qualifies (a runtime with
// one hardware queue, a profiler that serialises kernels) the caller runs without side lanes: correctness never depends
// on the outcome.  (The number of OTHER live streams in the process has no effect on kernel times once the arrays are
// placed reproducibly -- csrc/device_alloc.hip; an apparent effect of that kind in this round's notes was the allocator.)
#include <chrono>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {

__global__ __launch_bounds__(64) void spin_us_kernel(int us) {
  const long long t0 = wall_clock64();   // 100 MHz constant clock
  while (wall_clock64() - t0 < 100ll * us) __builtin_amdgcn_s_sleep(16);
}

bool pf_streams_overlap(hipStream_t a, hipStream_t b) {
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, a, 120);
  hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, b, 120);
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  return us < 200.0;
}

// a new non-blocking stream that runs beside every stream of avoid[0 .. navoid) and beside the default stream; *found = whether
// such a stream was found (else the returned stream is simply the last candidate)
hipError_t pf_acquire_stream(const hipStream_t* avoid, int navoid, hipStream_t* out, bool* found) {
  std::vector<hipStream_t> rejected;
  hipStream_t got = nullptr;
  bool ok = false;
  for (int tries = 0; tries < 10 && !ok; ++tries) {
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
      return e;
    }
    ok = pf_streams_overlap(nullptr, s);
    for (int j = 0; ok && j < navoid; ++j) ok = pf_streams_overlap(avoid[j], s);
    if (ok || tries == 9)
      got = s;
    else
      rejected.push_back(s);
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  (void)hipGetLastError();
  *out = got;
  if (found) *found = ok;
  return hipSuccess;
}

// up to `want` side streams for the handle stream `main`: each runs beside `main`, the default stream and the others
int pf_acquire_side_streams(hipStream_t main, int want, hipStream_t* out) {
  int have = 0;
  for (; have < want; ++have) {
    std::vector<hipStream_t> avoid(1, main);
    for (int j = 0; j < have; ++j) avoid.push_back(out[j]);
    hipStream_t s = nullptr;
    bool found = false;
    if (pf_acquire_stream(avoid.data(), (int)avoid.size(), &s, &found) != hipSuccess) break;
    if (!found) {
      if (s) (void)hipStreamDestroy(s);
      break;
    }
    out[have] = s;
  }
  return have;
}

}  // namespace pfhip
