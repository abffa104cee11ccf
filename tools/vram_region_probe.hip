// vram_region_probe -- is the bandwidth of an MI355X allocation a property of WHERE in VRAM it lands?
// Allocates `nblk` blocks of `gb` GiB one after the other (all held), and times on each block
//   copy : a streaming copy of the first half into the second half (16 B per lane, grid-stride)
//   col  : the access pattern of the z pass of the spectral step, in place: a workgroup moves 512 segments of 256 B that lie
//          one plane (2 166 912 B) apart through registers and writes them back
// Usage: vram_region_probe [nblk=20] [gb=4] [mode=0] [free_between=0] [chunk_mb=2] [shuffle=1]
//   mode 0 = hipMalloc, 1 = hipExtMallocWithFlags(hipDeviceMallocContiguous), 2 = virtual memory management: the block is
//   built from chunk_mb-MiB physical allocations (hipMemCreate) mapped into one reserved address range in shuffled order
// Build: hipcc -O3 --offload-arch=gfx950 tools/vram_region_probe.hip -o tools/bin/vram_region_probe
#include <hip/hip_runtime.h>

#include <algorithm>
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

__global__ __launch_bounds__(256) void copy_kernel(const d2* __restrict__ a, d2* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}

// plane stride in d2 elements: 513 rows x 264; a workgroup of 256 lanes handles 16 columns-of-segments: lane -> (z group, 16 B piece)
__global__ __launch_bounds__(256) void col_kernel(d2* __restrict__ a, long pstride, int nseg) {
  const int piece = threadIdx.x & 15, zl = threadIdx.x >> 4;  // 16 lanes cover one 256 B segment; 16 planes per sweep
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    d2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = a[(long)(zl + 16 * k) * pstride + (long)seg * 16 + piece];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[(long)(zl + 16 * k) * pstride + (long)seg * 16 + piece] = v[k] * 1.0000001;
  }
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 20;
  const double gb = argc > 2 ? atof(argv[2]) : 4.0;
  const int contig = argc > 3 ? atoi(argv[3]) : 0;
  const int free_between = argc > 4 ? atoi(argv[4]) : 0;
  const int chunk_mb = argc > 5 ? atoi(argv[5]) : 2;
  const int shuffle = argc > 6 ? atoi(argv[6]) : 1;
  const size_t bytes = (size_t)(gb * (1ull << 30));
  const long pstride = 513L * 264;               // d2 per plane
  const int nseg = (int)(pstride / 16);          // 256 B segments per plane
  if ((size_t)512 * pstride * 16 > bytes) { fprintf(stderr, "block too small for the column pattern\n"); return 2; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<void*> held;
  for (int b = 0; b < nblk; ++b) {
    void* p = nullptr;
    hipError_t e = hipSuccess;
    if (contig == 2) {
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned;
      prop.location.type = hipMemLocationTypeDevice;
      prop.location.id = 0;
      size_t gran = 0;
      CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
      const size_t chunk = (size_t)chunk_mb << 20;
      if (b == 0) printf("granularity %zu bytes, chunk %zu bytes\n", gran, chunk);
      const size_t nch = bytes / chunk;
      hipDeviceptr_t va = nullptr;
      CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
      std::vector<size_t> order(nch);
      for (size_t i = 0; i < nch; ++i) order[i] = i;
      if (shuffle) {  // deterministic shuffle (LCG)
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        for (size_t i = nch - 1; i > 0; --i) {
          st = st * 6364136223846793005ull + 1442695040888963407ull;
          std::swap(order[i], order[(size_t)((st >> 33) % (i + 1))]);
        }
      }
      for (size_t i = 0; i < nch; ++i) {
        hipMemGenericAllocationHandle_t hnd;
        CK(hipMemCreate(&hnd, chunk, &prop, 0));
        CK(hipMemMap((hipDeviceptr_t)((char*)va + order[i] * chunk), chunk, 0, hnd, 0));
        CK(hipMemRelease(hnd));
      }
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, bytes, &acc, 1));
      p = (void*)va;
    } else {
      e = contig ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) : hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) { printf("block %d: allocation failed (%s)\n", b, hipGetErrorString(e)); break; }
    CK(hipMemset(p, 0, bytes));
    const long n = (long)(bytes / 32);   // d2 elements in half a block
    float best_copy = 1e9f, best_col = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
      float ms;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, (const d2*)p, (d2*)p + n, n);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 4) best_copy = std::min(best_copy, ms);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(col_kernel, dim3(256 * 8), dim3(256), 0, 0, (d2*)p, pstride, nseg);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 4) best_col = std::min(best_col, ms);
    }
    printf("block %3d at %p  copy %7.1f GB/s   col %7.1f GB/s\n", b, p, 2.0 * n * 16 / best_copy / 1e6,
           2.0 * 512 * pstride * 16 / best_col / 1e6);
    fflush(stdout);
    if (contig == 2) continue;   // the probe leaves the mapped ranges to process exit
    if (free_between) CK(hipFree(p)); else held.push_back(p);
  }
  for (void* p : held) CK(hipFree(p));
  return 0;
}
