// mempattern_probe -- what the MI355X memory system gives the access patterns of the spectral passes, measured with pure
// copies (no transform): every workgroup moves "items" of ROWS x W bytes (W contiguous bytes per row, rows `rstride` apart)
// through LDS, in place or into a second array, with the load-all / barrier / store-all structure of the column passes or
// with a register pipeline.  The numbers decide the layout and tiling of csrc/spectral3d_pipe.hip.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mempattern_probe.hip -o tools/bin/mempattern_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

// a native vector type: with HIP_vector_type (d2) `v[i] = A[k]` is a struct copy that SROA leaves in scratch memory
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
  int64_t rstride;   // d2 elements between rows of an item
  int64_t bstride;   // between batches (inner level: batch % nb1)
  int64_t kstride;   // between the segment blocks of one batch (W/16 for [row][kx] layouts)
  int nblk, nitems, rows, wseg;  // wseg = W / 16 (d2 per row segment)
  int nb1 = 1 << 30;     // batches per outer group
  int64_t bstride2 = 0;  // between outer groups (batch / nb1)
  int lw = 0;            // log2(wseg), set by main
  int64_t off = 0;       // first element (chunked passes)
};

__device__ __forceinline__ int64_t item_base(const Pat& p, int item) {
  const int b = item / p.nblk;
  return p.off + (int64_t)(b % p.nb1) * p.bstride + (int64_t)(b / p.nb1) * p.bstride2 + (int64_t)(item % p.nblk) * p.kstride;
}

// load all (PER x 16 B per thread) -> LDS -> barrier -> store all; one item per workgroup (NT threads), or persistent
template <int NT, int PER, bool PERSIST>
__global__ __launch_bounds__(NT) void lsb_kernel(const d2* __restrict__ A, d2* __restrict__ B, const Pat p) {
  extern __shared__ __attribute__((aligned(16))) d2 L[];
  const int tid = threadIdx.x;
  for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
    const int64_t base = item_base(p, item);
    d2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + NT * i, r = e >> p.lw, c = e & (p.wseg - 1);
      v[i] = A[base + r * p.rstride + c];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) L[tid + NT * i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = L[(tid + NT * i) ^ 1];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (tid + NT * i) ^ 1, r = e >> p.lw, c = e & (p.wseg - 1);
      B[base + r * p.rstride + c] = v[i];
    }
    if (!PERSIST) break;
  }
}

// persistent, register-pipelined: the next item's loads are in flight while the current one is stored
template <int NT, int PER>
__global__ __launch_bounds__(NT) void pipe_kernel(const d2* __restrict__ A, d2* __restrict__ B, const Pat p) {
  const int tid = threadIdx.x;
  d2 v[PER], w[PER];
  int item = blockIdx.x;
  auto load = [&](int it, d2(&d)[PER]) {
    const int64_t base = item_base(p, it);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + NT * i, r = e >> p.lw, c = e & (p.wseg - 1);
      d[i] = A[base + r * p.rstride + c];
    }
  };
  if (item < p.nitems) load(item, v);
  while (item < p.nitems) {
    const int next = item + gridDim.x;
    if (next < p.nitems) load(next, w);
    const int64_t base = item_base(p, item);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + NT * i, r = e >> p.lw, c = e & (p.wseg - 1);
      B[base + r * p.rstride + c] = v[i];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = w[i];
    item = next;
  }
}

int main(int argc, char** argv) {
  const int nrep = argc > 1 ? atoi(argv[1]) : 100;
  const int n = 512, pitch = 264;
  const size_t elems = (size_t)n * n * pitch;  // 1.107 GB per array
  d2 *A, *B;
  CK(hipMalloc(&A, elems * sizeof(d2)));
  CK(hipMalloc(&B, elems * sizeof(d2)));
  CK(hipMemset(A, 0, elems * sizeof(d2)));
  CK(hipMemset(B, 0, elems * sizeof(d2)));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)lsb_kernel<512, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)lsb_kernel<512, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  // pre-heat: ~0.3 s of copies (the chip answers a load step from idle with a reduced clock for ~25 ms)
  {
    Pat p{pitch, (int64_t)n * pitch, 8, 33, 33 * n, 512, 8};
    p.lw = 3;
    for (int i = 0; i < 600; ++i) hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, A, B, p);
    CK(hipDeviceSynchronize());
  }
  struct Case {
    const char* name;
    Pat p;
  };
  std::vector<Case> cases;
  const int64_t plane = (int64_t)n * pitch;
  // the passes as they are: 8 columns (128 B) x 512 rows
  cases.push_back({"y-pass  128B x 512 rows, stride 4224 B", Pat{pitch, plane, 8, 33, 33 * n, 512, 8}});
  cases.push_back({"z-pass  128B x 512 rows, stride 2.06 MB", Pat{plane, pitch, 8, 33, 33 * n, 512, 8}});
  // wider segments, same 64 KB items (fewer rows): what the segment width is worth at the y / z strides
  for (int w : {16, 32, 64, 256}) {
    static char nm[8][64];
    static int k = 0;
    const int rows = 4096 / w;
    const int nblk = w == 256 ? 1 : 264 / w;  // w = 256: the whole 4224-byte row (264 elements) -> handled as 256 + rest ignored
    snprintf(nm[k], 64, "y-like  %4dB x %3d rows, stride 4224 B", w * 16, rows);
    cases.push_back({nm[k++], Pat{pitch, (int64_t)rows * pitch, w, nblk, nblk * (n * n / rows), rows, w}});
    snprintf(nm[k], 64, "z-like  %4dB x %3d rows, stride 2.06 MB", w * 16, rows);
    // rows along z: take `rows` planes apart... emulate with the same plane stride; batches enumerate (y, z-group)
    cases.push_back({nm[k++], Pat{plane, pitch, w, nblk, nblk * n * (n / rows), rows, w, n, (int64_t)rows * plane}});
  }
  // contiguous 64 KB spans (the ideal): rows of 4096 elements... = one item is 64 KB contiguous
  cases.push_back({"span    64 KB contiguous per item", Pat{4096, 4096, 4096, 1, (int)(elems / 4096), 1, 4096}});
  const double GB = 1e9;
  printf("device %s, %d CUs, nrep %d\n", prop.name, ncu, nrep);
  printf("%-44s %10s %10s %10s %10s %10s\n", "pattern", "lsb", "lsb-inpl", "lsb-pers", "pipe", "pipe-inpl");
  for (auto& c : cases) {
    Pat& p = c.p;
    for (p.lw = 0; (1 << p.lw) < p.wseg; ++p.lw) {}
    const double bytes = 2.0 * (double)p.nitems * p.rows * p.wseg * 16.0;
    double res[5];
    for (int form = 0; form < 5; ++form) {
      auto launch = [&]() {
        switch (form) {
          case 0: hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, A, B, p); break;
          case 1: hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, A, A, p); break;
          case 2: hipLaunchKernelGGL((lsb_kernel<512, 8, true>), dim3(2 * ncu), dim3(512), 65536, 0, A, A, p); break;
          case 3: hipLaunchKernelGGL((pipe_kernel<512, 8>), dim3(2 * ncu), dim3(512), 0, 0, A, B, p); break;
          default: hipLaunchKernelGGL((pipe_kernel<512, 8>), dim3(2 * ncu), dim3(512), 0, 0, A, A, p); break;
        }
      };
      for (int w = 0; w < 5; ++w) launch();
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < nrep; ++r) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      res[form] = bytes / (ms / nrep * 1e-3) / GB;
    }
    printf("%-44s %10.0f %10.0f %10.0f %10.0f %10.0f   (%.3f GB per pass)\n", c.name, res[0], res[1], res[2], res[3], res[4],
           bytes / GB);
    fflush(stdout);
  }
  // ---- chunked passes: does the 256 MiB Infinity Cache keep a z-chunk between the passes of a step? ------------------------
  // per chunk of P planes: Y (in place on A, 128-byte segments along y) -> X (A -> B, whole contiguous planes) -> Y (in
  // place on B); against the same three passes over the whole box one after the other
  {
    auto ypass = [&](d2* X, int z0, int P, hipStream_t st) {
      Pat p{pitch, plane, 8, 33, 33 * P, 512, 8};
      p.lw = 3;
      p.off = (int64_t)z0 * plane;
      hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, st, X, X, p);
    };
    auto xpass = [&](int z0, int P, hipStream_t st) {
      Pat p{4096, 4096, 4096, 1, (int)(P * plane / 4096), 1, 4096};
      p.lw = 12;
      p.off = (int64_t)z0 * plane;
      hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, st, A, B, p);
    };
    hipStream_t st[3];
    for (auto& s_ : st) CK(hipStreamCreate(&s_));
    const double bytes3 = 3.0 * 2.0 * (double)n * plane * 16.0;
    printf("three passes (Y in place, X A->B, Y in place), %.2f GB moved per round:\n", bytes3 / GB);
    for (int P : {512, 128, 64, 32, 16, 8}) {
      for (int mode = 0; mode < 2; ++mode) {   // 0: one stream; 1: chunks round-robin over three streams
        if (P == 512 && mode == 1) continue;
        auto round = [&]() {
          int k = 0;
          for (int z0 = 0; z0 < n; z0 += P, ++k) {
            hipStream_t s_ = mode ? st[k % 3] : st[0];
            ypass(A, z0, P, s_);
            xpass(z0, P, s_);
            ypass(B, z0, P, s_);
          }
        };
        for (int w = 0; w < 3; ++w) round();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        // (mode 1: the other streams are idle at this point and join at the end through the device sync below)
        const int reps = 20;
        for (int r = 0; r < reps; ++r) round();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  chunk %3d planes (%6.1f MB per array), %s: %8.1f us per round, %6.0f GB/s\n", P, P * plane * 16.0 / 1e6,
               mode ? "3 streams" : "1 stream ", ms / reps * 1e3, bytes3 / (ms / reps * 1e-3) / GB);
        fflush(stdout);
      }
    }
  }
  // ---- placement: is the rate of a pattern a property of the ALLOCATION? ----------------------------------------------------
  // `nbuf` buffers of one half-spectrum array each, all held; the y- and z-pattern in-place copies and the contiguous copy
  // on each (round 3 saw the spectral step alternate between 2.43 and 2.62 ms from handle to handle)
  {
    const int nbuf = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<d2*> bufs;
    for (int i = 0; i < nbuf; ++i) {
      d2* X = nullptr;
      if (hipMalloc(&X, elems * sizeof(d2)) != hipSuccess) break;
      CK(hipMemset(X, 0, elems * sizeof(d2)));
      bufs.push_back(X);
    }
    printf("placement: %zu buffers of %.2f GB; GB/s of the in-place y-pattern / z-pattern / z-pattern 256 B / contiguous copy\n",
           bufs.size(), elems * 16.0 / 1e9);
    for (size_t i = 0; i < bufs.size(); ++i) {
      d2* X = bufs[i];
      Pat py{pitch, plane, 8, 33, 33 * n, 512, 8};
      py.lw = 3;
      Pat pz{plane, pitch, 8, 33, 33 * n, 512, 8};
      pz.lw = 3;
      Pat pz2{plane, pitch, 16, 16, 16 * n * 2, 256, 16, n, (int64_t)256 * plane};
      pz2.lw = 4;
      Pat ps{4096, 4096, 4096, 1, (int)(elems / 4096), 1, 4096};
      ps.lw = 12;
      double r[4];
      int k = 0;
      for (Pat* pp : {&py, &pz, &pz2, &ps}) {
        const Pat p = *pp;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, X, X, p);
        CK(hipEventRecord(e0, 0));
        for (int q = 0; q < 20; ++q) hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, X, X, p);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        r[k++] = 2.0 * (double)p.nitems * p.rows * p.wseg * 16.0 / (ms / 20 * 1e-3) / GB;
      }
      // does the Infinity Cache serve THIS buffer?  y pass in place on the whole array against the same pass on a 32-plane
      // piece repeated 16 times (69 MB: resident after the first round if the pages may allocate there)
      double mall[2];
      for (int m = 0; m < 2; ++m) {
        Pat p{pitch, plane, 8, 33, 33 * (m ? 32 : n), 512, 8};
        p.lw = 3;
        const int reps = m ? 16 * 20 : 20;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, X, X, p);
        CK(hipEventRecord(e0, 0));
        for (int q = 0; q < reps; ++q) hipLaunchKernelGGL((lsb_kernel<512, 8, false>), dim3(p.nitems), dim3(512), 65536, 0, X, X, p);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        mall[m] = 2.0 * (double)p.nitems * p.rows * p.wseg * 16.0 / (ms / reps * 1e-3) / GB;
      }
      printf("  buffer %zu at %p: y %6.0f  z %6.0f  z256 %6.0f  span %6.0f | y on a resident 32-plane piece %6.0f (whole %6.0f)\n", i,
             (void*)X, r[0], r[1], r[2], r[3], mall[1], mall[0]);
      fflush(stdout);
    }
  }
  return 0;
}
