// colpass_probe -- stand-alone measurement of the 512-point column pass of the 3-D spectral step (csrc/spectral2d_fused.hip,
// csrc/spectral3d_pipe.hip) on the 512^3 half spectrum [z][y][pitch 264] of complex doubles: one in-place FFT per column
// along y (stride one row, 4224 B) or z (stride one plane, 2.06 MB), 8 adjacent k_x columns (one 128-byte line per row) per
// work item.  Forms:
//   reg    one item per workgroup: coalesced register loads -> LDS -> one wave per column transforms -> LDS -> coalesced
//          stores (the round-2/3 structure, 2 workgroups per CU)
//   dma    persistent workgroups, one per CU: the NEXT item streams into a second LDS buffer by LDS-DMA
//          (global_load_lds_dwordx4, each wave fetching its own column) while the current one is transformed and stored
//   dmau   the same with every wave storing its own column straight from registers (no workgroup barrier at all)
//   copy*  the same data movement without the transform (what the memory system gives the access pattern)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I pfhubbenchmarks_amd/csrc tools/colpass_probe.hip -o gpurun_out/colpass_probe
// Run:   colpass_probe [nrep]      prints one line per (form, axis): us per pass, GB/s of (read + write) bytes, max |diff| vs reg
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft512_wave.h"

using namespace pfhip;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Geom {
  int64_t rstride, bstride;  // complex elements between rows of a column / between batches
  int nbatch, nblk, nxh, nitems;
};

__device__ __forceinline__ int nat(int n) { return n + (n >> 3); }

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- reg: the round-3 structure ------------------------------------------------------------------------------------------
constexpr int W8C = W8 + 32;
__device__ int g_stagger_us = 0;   // > 0: the second workgroup of every CU (blocks 256..511) starts that much later
template <int SIGN, bool FFT>
__global__ __launch_bounds__(512, 4) void reg_kernel(double2* __restrict__ A, const Geom g,
                                                     const double2* __restrict__ twA_g, const double2* __restrict__ twB_g) {
  __shared__ __attribute__((aligned(16))) double2 Lall[8 * W8C];
  __shared__ __attribute__((aligned(16))) double2 TWB[72];
  if (g_stagger_us > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    const long long t0 = __builtin_readcyclecounter();   // s_memtime: 100 MHz
    while (__builtin_readcyclecounter() - t0 < 100ll * g_stagger_us) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x < 64) TWB[(threadIdx.x >> 3) * 9 + (threadIdx.x & 7)] = twB_g[threadIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double2* L = Lall + wave * W8C + 4 * wave;
  const int T = (lane >> 3) + 8 * (lane & 7);
  const int ci = tid & 7;
  double2* Lc = Lall + ci * W8C + 4 * ci;
  const int item = blockIdx.x;
  const int b = item / g.nblk, kx = (item % g.nblk) * 8 + ci;
  const bool on = kx < g.nxh;
  double2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (tid + 512 * i) >> 3;
    v[i] = on ? A[b * g.bstride + r * g.rstride + kx] : make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) Lc[nat((tid + 512 * i) >> 3)] = v[i];
  __syncthreads();
  if (FFT) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = L[nat(lane + 64 * j)];
    fft512_wave_tw<SIGN, 9>(v, L, lane, twA_g, TWB, lane);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) L[nat(T + 64 * t)] = v[t];
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (tid + 512 * i) >> 3;
    if (on) A[b * g.bstride + r * g.rstride + kx] = Lc[nat(r)];
  }
}

// ---- dma: persistent, double-buffered by LDS-DMA ---------------------------------------------------------------------------
// one 16-byte LDS-DMA load per lane: LDS destination = lds_byte (wave-uniform) + 16 * lane
__device__ __forceinline__ void glds16(const double2* gsrc, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int COLS = 576;  // slots per column region: the first 512 receive the column, all 576 serve the FFT exchanges
// UNCO: every wave stores its own column from registers (16 bytes per lane, 64 lines per instruction); else cooperative
// full-line stores behind a workgroup barrier
template <int SIGN, bool FFT, bool UNCO>
__global__ __launch_bounds__(512, 2) void dma_kernel(double2* __restrict__ A, const Geom g,
                                                     const double2* __restrict__ twA_g, const double2* __restrict__ twB_g) {
  extern __shared__ __attribute__((aligned(16))) double2 smem[];  // [2][8][COLS] + TWB[72]
  double2* TWB = smem + 2 * 8 * COLS;
  if (threadIdx.x < 64) TWB[(threadIdx.x >> 3) * 9 + (threadIdx.x & 7)] = twB_g[threadIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (lane >> 3) + 8 * (lane & 7);
  const int ci = tid & 7;
  const unsigned smem_byte = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  double2 twA[7], twB[7];
  load_tw(twA, twA_g, lane);
  load_tw(twB, twB_g, lane & 7);
  auto issue = [&](int item, int buf) {
    const int b = item / g.nblk, kx = (item % g.nblk) * 8 + wave;
    const double2* src = A + b * g.bstride + kx + (int64_t)lane * g.rstride;
    const unsigned dst = __builtin_amdgcn_readfirstlane(smem_byte + (unsigned)((buf * 8 + wave) * COLS) * 16u);
#pragma unroll
    for (int k = 0; k < 8; ++k) glds16(src + (int64_t)(64 * k) * g.rstride, dst + (unsigned)(64 * k) * 16u);
  };
  int item = blockIdx.x, cur = 0;
  bool first = true;
  if (item < g.nitems) issue(item, 0);
  __syncthreads();  // TWB
  while (item < g.nitems) {
    const int next = item + gridDim.x;
    // in the queue, oldest first: this item's 8 DMA loads, the previous item's 8 stores (none in the first round), the 8
    // DMA loads of the next item issued here -- wait for exactly the first group
    if (next < g.nitems) {
      issue(next, cur ^ 1);
      if (first)
        wait_vm<8>();
      else
        wait_vm<16>();
    } else {
      if (first)
        wait_vm<0>();
      else
        wait_vm<8>();
    }
    first = false;
    double2* L = smem + (cur * 8 + wave) * COLS;
    double2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = L[lane + 64 * j];
    if (FFT) {
      fft512_wave<SIGN>(v, L, lane, twA, twB, lane);
    }
    const int b = item / g.nblk, kx0 = (item % g.nblk) * 8;
    if (UNCO) {
      // (the last block's waves 1..7 write pad columns: allocated, never read -- every wave issues 8 stores per item, so
      // the vmcnt arithmetic above holds)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int r = FFT ? T + 64 * t : lane + 64 * t;
        A[b * g.bstride + r * g.rstride + kx0 + wave] = v[t];
      }
      wave_lds_sync();
    } else {
      wave_lds_sync();
#pragma unroll
      for (int t = 0; t < 8; ++t) L[nat(FFT ? T + 64 * t : lane + 64 * t)] = v[t];
      lds_barrier();
      const double2* Lc = smem + (cur * 8 + ci) * COLS;
      const bool on = kx0 + ci < g.nxh;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = (tid + 512 * i) >> 3;
        if (on) A[b * g.bstride + r * g.rstride + kx0 + ci] = Lc[nat(r)];
      }
      lds_barrier();  // every read of this buffer is done before the next iteration's DMA overwrites it
    }
    item = next;
    cur ^= 1;
  }
}

static void tables(std::vector<double2>& ta, std::vector<double2>& tb) {
  const double TWO_PI = 6.283185307179586476925286766559;
  ta.resize(512);
  tb.resize(64);
  for (int l = 0; l < 64; ++l)
    for (int q = 0; q < 8; ++q) {
      const double ang = TWO_PI * (double)(l * q) / 512.0;
      ta[l * 8 + q] = make_double2(std::cos(ang), -std::sin(ang));
    }
  for (int l0 = 0; l0 < 8; ++l0)
    for (int s = 0; s < 8; ++s) {
      const double ang = TWO_PI * (double)(l0 * s) / 64.0;
      tb[l0 * 8 + s] = make_double2(std::cos(ang), -std::sin(ang));
    }
}

int main(int argc, char** argv) {
  const int nrep = argc > 1 ? atoi(argv[1]) : 20;
  const int n = 512, nxh = 257, pitch = 264;
  const size_t elems = (size_t)n * n * pitch;
  double2 *A, *B, *ref, *twa, *twb;
  CK(hipMalloc(&A, elems * sizeof(double2)));
  CK(hipMalloc(&B, elems * sizeof(double2)));
  CK(hipMalloc(&ref, elems * sizeof(double2)));
  std::vector<double2> ta, tb;
  tables(ta, tb);
  CK(hipMalloc(&twa, 512 * sizeof(double2)));
  CK(hipMalloc(&twb, 64 * sizeof(double2)));
  CK(hipMemcpy(twa, ta.data(), 512 * sizeof(double2), hipMemcpyHostToDevice));
  CK(hipMemcpy(twb, tb.data(), 64 * sizeof(double2), hipMemcpyHostToDevice));
  {  // input: a deterministic pattern (host-generated once)
    std::vector<double2> h(elems);
    unsigned s = 12345u;
    for (size_t i = 0; i < elems; ++i) {
      s = s * 1664525u + 1013904223u;
      const double x = (double)(s >> 8) / 16777216.0 - 0.5;
      s = s * 1664525u + 1013904223u;
      h[i] = make_double2(x, (double)(s >> 8) / 16777216.0 - 0.5);
    }
    CK(hipMemcpy(B, h.data(), elems * sizeof(double2), hipMemcpyHostToDevice));
  }
  int ncu = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  ncu = prop.multiProcessorCount;
  const size_t lds = (2 * 8 * COLS + 72) * sizeof(double2);
  auto big = [&](const void* k) { CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); };
  big((const void*)dma_kernel<-1, true, false>);
  big((const void*)dma_kernel<-1, true, true>);
  big((const void*)dma_kernel<-1, false, false>);
  big((const void*)dma_kernel<-1, false, true>);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 2.0 * 512.0 * (argc > 2 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 512.0) * nxh * 16.0;
  printf("device %s, %d CUs; bytes per pass (read + write) %.3f GB; dma LDS %zu B\n", prop.name, ncu, bytes / 1e9, lds);
  {  // pre-heat (~0.2 s): the chip answers a load step from idle with a reduced clock for ~25 ms
    Geom g{pitch, (int64_t)n * pitch, n, 33, nxh, 33 * n};
    CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((reg_kernel<-1, false>), dim3(g.nitems), dim3(512), 0, 0, A, g, twa, twb);
    CK(hipDeviceSynchronize());
  }
  const int resident = argc > 2 ? atoi(argv[2]) : 0;   // > 0: the y pass on a piece of that many planes only (stays in the
                                                        // 256 MiB Infinity Cache between repetitions: the chunked regime)
  for (int axis = 1; axis <= (resident ? 1 : 2); ++axis) {
    Geom g;
    g.nxh = nxh;
    g.nblk = (nxh + 7) / 8;
    g.nbatch = resident ? resident : n;
    g.nitems = g.nblk * g.nbatch;
    g.rstride = axis == 1 ? pitch : (int64_t)n * pitch;
    g.bstride = axis == 1 ? (int64_t)n * pitch : pitch;
    for (int form = 0; form < 6; ++form) {
      const char* names[6] = {"reg", "dma", "dmau", "copy_reg", "copy_dma", "copy_dmau"};
      auto launch = [&]() {
        switch (form) {
          case 0: hipLaunchKernelGGL((reg_kernel<-1, true>), dim3(g.nitems), dim3(512), 0, 0, A, g, twa, twb); break;
          case 1: hipLaunchKernelGGL((dma_kernel<-1, true, false>), dim3(ncu), dim3(512), lds, 0, A, g, twa, twb); break;
          case 2: hipLaunchKernelGGL((dma_kernel<-1, true, true>), dim3(ncu), dim3(512), lds, 0, A, g, twa, twb); break;
          case 3: hipLaunchKernelGGL((reg_kernel<-1, false>), dim3(g.nitems), dim3(512), 0, 0, A, g, twa, twb); break;
          case 4: hipLaunchKernelGGL((dma_kernel<-1, false, false>), dim3(ncu), dim3(512), lds, 0, A, g, twa, twb); break;
          default: hipLaunchKernelGGL((dma_kernel<-1, false, true>), dim3(ncu), dim3(512), lds, 0, A, g, twa, twb); break;
        }
      };
      // correctness: one pass on a fresh copy of the input, compared with form 0's result (form 3..5: with the input)
      CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
      launch();
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      double maxdiff = 0.0;
      if (form == 0) {
        CK(hipMemcpy(ref, A, elems * sizeof(double2), hipMemcpyDeviceToDevice));
      } else {
        const double2* cmp = form < 3 ? ref : B;
        std::vector<double2> x(1 << 20), y(1 << 20);
        for (size_t off = 0; off + (1 << 20) <= elems; off += (size_t)37 << 20) {
          CK(hipMemcpy(x.data(), A + off, x.size() * sizeof(double2), hipMemcpyDeviceToHost));
          CK(hipMemcpy(y.data(), cmp + off, y.size() * sizeof(double2), hipMemcpyDeviceToHost));
          for (size_t i = 0; i < x.size(); ++i) {
            if ((off + i) % pitch >= (size_t)nxh) continue;  // pad columns
            maxdiff = std::fmax(maxdiff, std::fmax(std::fabs(x[i].x - y[i].x), std::fabs(x[i].y - y[i].y)));
          }
        }
      }
      // timing: data content does not matter (values grow under repeated unnormalised transforms -> rescale by reloading)
      for (int w = 0; w < 3; ++w) launch();
      CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
      for (int w = 0; w < 2; ++w) launch();
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < nrep; ++r) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("axis %c %-10s %8.1f us/pass %7.1f GB/s  maxdiff %.3g\n", axis == 1 ? 'y' : 'z', names[form], ms / nrep * 1e3,
             bytes / (ms / nrep * 1e-3) / 1e9, maxdiff);
      fflush(stdout);
    }
  }
  if (!resident) {   // is the y pass (real transform) equally fast on every stream?  (streams share a few hardware queues)
    Geom g;
    g.nxh = nxh;
    g.nblk = 33;
    g.nbatch = n;
    g.nitems = g.nblk * g.nbatch;
    g.rstride = pitch;
    g.bstride = (int64_t)n * pitch;
    for (int sidx = 0; sidx < 10; ++sidx) {
      hipStream_t st;
      CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((reg_kernel<-1, true>), dim3(g.nitems), dim3(512), 0, st, A, g, twa, twb);
      CK(hipMemcpyAsync(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice, st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((reg_kernel<-1, true>), dim3(g.nitems), dim3(512), 0, st, A, g, twa, twb);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stream %d (kept alive): y pass reg %8.1f us/pass\n", sidx, ms / 20 * 1e3);
    }
  }
  if (resident) {
    Geom g;
    g.nxh = nxh;
    g.nblk = 33;
    g.nbatch = resident;
    g.nitems = g.nblk * g.nbatch;
    g.rstride = pitch;
    g.bstride = (int64_t)n * pitch;
    for (int st : {0, 2, 4, 6, 8, 12}) {
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger_us), &st, sizeof(int)));
      for (int form = 0; form < 2; ++form) {
        auto launch = [&]() {
          if (form == 0)
            hipLaunchKernelGGL((reg_kernel<-1, true>), dim3(g.nitems), dim3(512), 0, 0, A, g, twa, twb);
          else
            hipLaunchKernelGGL((reg_kernel<-1, false>), dim3(g.nitems), dim3(512), 0, 0, A, g, twa, twb);
        };
        CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
        for (int w = 0; w < 5; ++w) launch();
        CK(hipMemcpy(A, B, elems * sizeof(double2), hipMemcpyDeviceToDevice));
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < nrep; ++r) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stagger %2d us  %-8s %8.1f us/pass %7.1f GB/s\n", st, form ? "copy_reg" : "reg", ms / nrep * 1e3,
               bytes / (ms / nrep * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
