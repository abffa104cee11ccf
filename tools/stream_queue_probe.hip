// stream_queue_probe -- are all HIP streams equal?  ROCm multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
// default 4).  (1) the same back-to-back kernel sequence timed on each of 10 streams created in a row; (2) for every pair
// of streams, two long kernels launched at once: elapsed ~1x = they overlap (different hardware queues), ~2x = they share one.
// Round 4 found the 512^3 spectral step alternating between 2.09 and 2.31 ms from handle to handle on the SAME memory:
// its two side streams sometimes share a hardware queue.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_queue_probe.hip -o tools/bin/stream_queue_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// occupies `nwg` CUs for ~us microseconds without touching memory (half the chip: two of them fit side by side)
__global__ __launch_bounds__(64) void spin_kernel(int us) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 100ll * us) __builtin_amdgcn_s_sleep(16);
}

int main() {
  const size_t n = (size_t)64 << 20;  // 1 GiB per array
  d2 *A, *B;
  CK(hipMalloc(&A, n * sizeof(d2)));
  CK(hipMalloc(&B, n * sizeof(d2)));
  CK(hipMemset(A, 0, n * sizeof(d2)));
  CK(hipMemset(B, 0, n * sizeof(d2)));
  const int NS = 10;
  std::vector<hipStream_t> st(NS);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st[0], A, B, n);
  CK(hipDeviceSynchronize());
  printf("(1) 40 back-to-back 1 GiB copies per stream:\n");
  for (int s = 0; s < NS; ++s) {
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st[s], A, B, n);
    CK(hipEventRecord(e0, st[s]));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st[s], A, B, n);
    CK(hipEventRecord(e1, st[s]));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  stream %d: %.1f us per copy, %.0f GB/s\n", s, ms / 40 * 1e3, 2.0 * n * 16 / (ms / 40 * 1e-3) / 1e9);
  }
  printf("(2) two 200 us spin kernels of 64 workgroups at once, streams i and j: elapsed us (~200: concurrent, ~400: one hardware queue)\n     ");
  for (int j = 0; j < NS; ++j) printf("%6d", j);
  printf("\n");
  for (int i = 0; i < NS; ++i) {
    printf("  %2d ", i);
    for (int j = 0; j < NS; ++j) {
      if (j <= i) {
        printf("%6s", "");
        continue;
      }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, st[i], 200);
      hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, st[j], 200);
      CK(hipStreamSynchronize(st[i]));
      CK(hipStreamSynchronize(st[j]));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%6.0f", ms * 1e3);
    }
    printf("\n");
  }
  return 0;
}
