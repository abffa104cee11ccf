// pairstream_probe -- does the rate of a column pass that streams TWO arrays at once (the z pass of the spectral step reads
// the work array and the resident spectrum of the same columns and writes both) depend on the DISTANCE between the arrays?
// One allocation (physically contiguous if the runtime grants hipDeviceMallocContiguous), A at its start, B = A + one array
// + delta; items as in the column passes (8 columns = 128 bytes x 512 rows, stride one row (y) or one plane (z)); load both,
// LDS round trip, store both.  Prints GB/s per delta.
// Build: hipcc -O3 --offload-arch=gfx950 tools/pairstream_probe.hip -o tools/bin/pairstream_probe
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

struct Pat {
  int64_t rstride, bstride;
  int nblk, nitems;
};

__global__ __launch_bounds__(512, 4) void pair_kernel(d2* __restrict__ A, d2* __restrict__ B, const Pat p) {
  __shared__ __attribute__((aligned(16))) d2 L[4096];
  const int tid = threadIdx.x, item = blockIdx.x;
  const int64_t base = (int64_t)(item / p.nblk) * p.bstride + (int64_t)(item % p.nblk) * 8;
  d2 v[8], w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 512 * i;
    v[i] = A[base + (e >> 3) * p.rstride + (e & 7)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) L[tid + 512 * i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = L[(tid + 512 * i) ^ 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {   // the second stream is requested late, like the resident spectrum in the z pass
    const int e = (tid + 512 * i) ^ 1;
    w[i] = B[base + (e >> 3) * p.rstride + (e & 7)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = (tid + 512 * i) ^ 1;
    const int64_t o = base + (e >> 3) * p.rstride + (e & 7);
    B[o] = w[i] + v[i];
    A[o] = v[i] - w[i];
  }
}

int main(int argc, char** argv) {
  const int n = 512, pitch = 264;
  const size_t elems = (size_t)n * n * pitch;
  const size_t slot = elems * sizeof(d2), extra = (size_t)96 << 20;
  unsigned char* blk = nullptr;
  bool contig = true;
  if (argc > 1 && atoi(argv[1]) == 0) contig = false;
  if (!(contig && hipExtMallocWithFlags((void**)&blk, 2 * slot + extra, hipDeviceMallocContiguous) == hipSuccess)) {
    (void)hipGetLastError();
    contig = false;
    CK(hipMalloc(&blk, 2 * slot + extra));
  }
  CK(hipMemset(blk, 0, 2 * slot + extra));
  printf("block %p (%s), slot %zu bytes = 0x%zx\n", (void*)blk, contig ? "physically contiguous" : "plain hipMalloc", slot, slot);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int64_t plane = (int64_t)n * pitch;
  const Pat py{pitch, plane, 33, 33 * n}, pz{plane, pitch, 33, 33 * n};
  d2* A = reinterpret_cast<d2*>(blk);
  for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(pair_kernel, dim3(py.nitems), dim3(512), 0, 0, A, reinterpret_cast<d2*>(blk + slot), py);
  CK(hipDeviceSynchronize());
  const size_t K = 1024, M = K * K;
  const size_t deltas[] = {0, 256, 4 * K, 16 * K, 64 * K, 128 * K, 192 * K, 256 * K, 384 * K, 512 * K, 768 * K, M, M + 64 * K, 2 * M,
                           2 * M + 64 * K, 3 * M, 4 * M, 6 * M, 8 * M, 12 * M, 16 * M, 16 * M + 64 * K, 24 * M, 32 * M, 33 * M,
                           48 * M, 64 * M, 64 * M + 4 * K, 65 * M + 320 * K};
  printf("%-22s %10s %10s\n", "delta (B - A - slot)", "y GB/s", "z GB/s");
  for (size_t d : deltas) {
    d2* B = reinterpret_cast<d2*>(blk + slot + d);
    double r[2];
    int k = 0;
    for (const Pat* pp : {&py, &pz}) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(pair_kernel, dim3(pp->nitems), dim3(512), 0, 0, A, B, *pp);
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(pair_kernel, dim3(pp->nitems), dim3(512), 0, 0, A, B, *pp);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      r[k++] = 4.0 * 33.0 * n * 512 * 128 / (ms / 20 * 1e-3) / 1e9;
    }
    printf("%10zu B %8.2f MB %10.0f %10.0f\n", d, d / 1048576.0, r[0], r[1]);
    fflush(stdout);
  }
  // ---- second table: plane stride (rows of a z column) and small deltas, same kernel ------------------------------------------
  {
    printf("z pattern: plane stride = 2162688 B + pad; delta = B - A - slot\n%-12s %-12s %10s\n", "pad", "delta", "z GB/s");
    const size_t pads[] = {0, 256, 512, 768, 1024, 2048, 4096, 4224, 8192, 16384, 32768, 65536};
    const size_t dl[] = {0, 256, 512, 1024, 2048};
    for (size_t pad : pads)
      for (size_t d : dl) {
        Pat p{plane + (int64_t)(pad / 16), pitch, 33, 33 * n};
        d2* B = reinterpret_cast<d2*>(blk + slot + ((size_t)48 << 20) + d);   // (room for the padded planes of A)
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(pair_kernel, dim3(p.nitems), dim3(512), 0, 0, A, B, p);
        CK(hipEventRecord(e0, 0));
        for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(pair_kernel, dim3(p.nitems), dim3(512), 0, 0, A, B, p);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12zu %-12zu %10.0f\n", pad, d, 4.0 * 33.0 * n * 512 * 128 / (ms / 20 * 1e-3) / 1e9);
        fflush(stdout);
      }
  }
  return 0;
}
