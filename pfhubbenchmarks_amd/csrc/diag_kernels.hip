// Diagnostics reductions and the initial-condition kernel (fp64, gfx950).
//
//   total_solute       = int c dx                                   dolfin/bench1.py:121-122
//   total_free_energy  = int f_chem + kappa/2 |grad c|^2 (+ k c phi/2) dx   dolfin/bench1.py:124-125, bench6.py:155-165
//   IC                 = pfbase.py:187-189 (BM1), :332-334 (BM6); extruded along z as in dolfin/b13d.py:55
//
// The kernel produces RAW sums {sum c, sum f_chem, sum of squared forward differences, sum c*phi, min c, max c};
// the host scales them (h^d, kappa/(2 h^2), k/2).  Two-stage, fixed-order reduction (wave shuffles -> LDS ->
// per-block partials -> one final block): no float atomics, so results are bitwise reproducible run to run.
#include "pfhip_internal.h"

namespace pfhip {
namespace {

constexpr int DIAG_BLOCK = 256;
constexpr int DIAG_MAX_BLOCKS = 8192;

__device__ __forceinline__ int wrapi(int i, int n) {
  i %= n;
  return i < 0 ? i + n : i;
}

// zends: bit 0 / bit 1 = local plane 0 / nz-1 is a no-flux wall of a z-line decomposition (mirror bc without the even
// extension along z): trapezoid weight 1/2 for the wall planes, and no forward z-difference out of the top wall.
// zends == 0 leaves every product a multiplication by 1.0, i.e. the sums are bitwise those of the unweighted form.
__device__ __forceinline__ double zweight(int z, int nz, int zends) {
  return ((z == 0 && (zends & 1)) || (z == nz - 1 && (zends & 2))) ? 0.5 : 1.0;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// reduce 6 values over a 256-thread block; result valid in thread 0
__device__ __forceinline__ void block_reduce6(double v[6], double* sh /* 4 waves x 6 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double r[6];
  r[0] = wave_sum(v[0]);
  r[1] = wave_sum(v[1]);
  r[2] = wave_sum(v[2]);
  r[3] = wave_sum(v[3]);
  r[4] = wave_min(v[4]);
  r[5] = wave_max(v[5]);
  if (lane == 0)
    for (int q = 0; q < 6; ++q) sh[wave * 6 + q] = r[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) v[q] = ((sh[q] + sh[6 + q]) + (sh[12 + q] + sh[18 + q]));
    v[4] = fmin(fmin(sh[4], sh[10]), fmin(sh[16], sh[22]));
    v[5] = fmax(fmax(sh[5], sh[11]), fmax(sh[17], sh[23]));
  }
}

// Generic partial kernel (any nx): one thread per cell, grid-stride; neighbours through the caches.
__global__ __launch_bounds__(DIAG_BLOCK) void diag_partial_kernel(const double* __restrict__ c,
                                                                 const double* __restrict__ phi, int nx, int ny,
                                                                 int nz, int ghost, int zwrap, int zends, double rho,
                                                                 double ca, double cb,
                                                                 double* __restrict__ partials) {
  __shared__ double sh[24];
  const int64_t plane = (int64_t)nx * ny;
  const int64_t total = plane * nz;
  double v[6] = {0.0, 0.0, 0.0, 0.0, INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * DIAG_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * DIAG_BLOCK) {
    const int x = (int)(i % nx);
    const int y = (int)((i / nx) % ny);
    const int z = (int)(i / plane);
    const int zp = zwrap ? wrapi(z + 1, nz) : z + 1;
    const double* p0 = c + (int64_t)(z + ghost) * plane;
    const double* pp = c + (int64_t)(zp + ghost) * plane;
    const int xp = x + 1 == nx ? 0 : x + 1;
    const int yp = y + 1 == ny ? 0 : y + 1;
    const double w = p0[(int64_t)y * nx + x];
    const double a = w - ca, b = cb - w, ab = a * b;
    const double dx = p0[(int64_t)y * nx + xp] - w;
    const double dy = p0[(int64_t)yp * nx + x] - w;
    const double dz = pp[(int64_t)y * nx + x] - w;
    const double wz = zweight(z, nz, zends), wd = (z == nz - 1 && (zends & 2)) ? 0.0 : 1.0;
    v[0] += wz * w;
    v[1] += wz * (rho * (ab * ab));
    v[2] += wz * (dx * dx + dy * dy) + wd * (dz * dz);
    if (phi) v[3] += wz * (w * phi[(int64_t)(z + ghost) * plane + (int64_t)y * nx + x]);
    v[4] = fmin(v[4], w);
    v[5] = fmax(v[5], w);
  }
  block_reduce6(v, sh);
  if (threadIdx.x == 0)
    for (int q = 0; q < 6; ++q) partials[(int64_t)blockIdx.x * 6 + q] = v[q];
}

// Streaming partial kernel (even nx): a wave owns one y-row of 128 cells (16-byte pairs per lane) and marches through a
// z-chunk keeping plane z in registers while plane z+1 arrives, so c is read from HBM once (8 B/cell algorithmic):
// the z+1 difference comes from registers, x+1 from the neighbouring lane (wave shuffle; the last lane loads one
// extra double), y+1 from the next row (an L1/L2 hit: the neighbouring wave streams it at the same time).
constexpr int DS_ROWS = 4;  // rows per 256-thread block
__global__ __launch_bounds__(DIAG_BLOCK) void diag_stream_kernel(const double* __restrict__ c,
                                                                const double* __restrict__ phi, int nx, int ny, int nz,
                                                                int ghost, int zwrap, int zends, int ntx, int nty,
                                                                int zchunk, double rho, double ca, double cb,
                                                                double* __restrict__ partials) {
  __shared__ double sh[24];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int tx = t % ntx, ty = (t / ntx) % nty, ch = t / (ntx * nty);
  const int x = tx * 128 + 2 * lane, y = ty * DS_ROWS + wave;
  const bool on = x < nx && y < ny;
  const int xc = on ? x : 0, yc = on ? y : 0;
  const int yp = yc + 1 == ny ? 0 : yc + 1;
  const int xn = xc + 2 >= nx ? 0 : xc + 2;  // first cell of the next pair (periodic)
  const int64_t plane = (int64_t)nx * ny;
  const int64_t o0 = (int64_t)yc * nx + xc, oy = (int64_t)yp * nx + xc, ox = (int64_t)yc * nx + xn;
  const int zs = ch * zchunk, ze = min(nz, zs + zchunk);
  double v[6] = {0.0, 0.0, 0.0, 0.0, INFINITY, -INFINITY};
  auto pl = [&](int z) { return c + (int64_t)((zwrap ? wrapi(z, nz) : z) + ghost) * plane; };
  double2 cur = *reinterpret_cast<const double2*>(pl(zs) + o0);
  for (int z = zs; z < ze; ++z) {
    const double* p0 = pl(z);
    const double2 nxt = *reinterpret_cast<const double2*>(pl(z + 1) + o0);
    const double2 up = *reinterpret_cast<const double2*>(p0 + oy);
    double xr = __shfl_down(cur.x, 1, 64);  // lane+1's first cell
    if (lane == 63 || xc + 2 >= nx) xr = p0[ox];
    if (on) {
      const double wz = zweight(z, nz, zends), wd = (z == nz - 1 && (zends & 2)) ? 0.0 : 1.0;
      double a = cur.x - ca, b = cb - cur.x, ab = a * b;
      double f = rho * (ab * ab);
      double dx = cur.y - cur.x, dy = up.x - cur.x, dz = nxt.x - cur.x;
      double g = wz * (dx * dx + dy * dy) + wd * (dz * dz);
      a = cur.y - ca;
      b = cb - cur.y;
      ab = a * b;
      f += rho * (ab * ab);
      dx = xr - cur.y;
      dy = up.y - cur.y;
      dz = nxt.y - cur.y;
      g += wz * (dx * dx + dy * dy) + wd * (dz * dz);
      v[0] += wz * (cur.x + cur.y);
      v[1] += wz * f;
      v[2] += g;
      if (phi) {
        const double2 ph = *reinterpret_cast<const double2*>(phi + (int64_t)(z + ghost) * plane + o0);
        v[3] += wz * (cur.x * ph.x + cur.y * ph.y);
      }
      v[4] = fmin(v[4], fmin(cur.x, cur.y));
      v[5] = fmax(v[5], fmax(cur.x, cur.y));
    }
    cur = nxt;
  }
  block_reduce6(v, sh);
  if (threadIdx.x == 0)
    for (int q = 0; q < 6; ++q) partials[(int64_t)blockIdx.x * 6 + q] = v[q];
}

// ---- streaming partial kernel, second generation (round 2) ---------------------------------------------------------
// diag_stream_kernel above reaches 3.6 TB/s at 512^3: every row is loaded twice (once by its owner, once as the "up" row
// of the wave below, requested in the iteration that consumes it) and only one plane per wave is in flight, i.e. 32 KiB
// of HBM requests per CU -- at ~2 us loaded latency that is the 4 TB/s it delivers.  Here
//   * a wave owns ROWS adjacent rows of 128 cells (16-byte buffer loads, 1 KiB per wave instruction): the y+1 neighbour
//     comes from the wave's own registers except for ONE halo row per ROWS rows (read over-fetch (ROWS+1)/ROWS from
//     L2, not 2x), x+1 by wave shuffle, z+1 from the register ring;
//   * planes z .. z+DEPTH are in flight in a register ring (the slot of plane z is refilled with plane z+DEPTH+1 as soon
//     as it has been consumed), loads are branch-free through buffer descriptors (a plane past the chunk gets a
//     zero-length descriptor, an inactive lane an out-of-range offset);
//   * XCD-aware tile order (blocks b, b+8, .. share an L2; each XCD takes a contiguous run of tiles) so the halo row is
//     an L2 hit; ~4 workgroups per CU, all co-resident.
typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t DOOB = 0x80000000u;
__device__ __forceinline__ auto dplane_rsrc(const double* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, (int)bytes, 0x00020000);
}
template <class R>
__device__ __forceinline__ double2 dbld2(R rs, uint32_t boff) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff, 0, 0));
}
template <class R>
__device__ __forceinline__ double dbld1(R rs, uint32_t boff) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(rs, (int)boff, 0, 0));
}

struct DiagArgs {
  const double* c;
  const double* phi;
  int nx, ny, nz, ghost, zwrap, zends;
  int ntx, nty, zchunk, ntiles;
  double rho, ca, cb;
  double* partials;
};

template <int ROWS, int DEPTH, bool HAS_PHI>
__global__ __launch_bounds__(DIAG_BLOCK) void diag_stream2_kernel(const DiagArgs k) {
  __shared__ double sh[24];
  constexpr int NS = DEPTH + 1;  // ring slots: planes z .. z+DEPTH
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double v[6] = {0.0, 0.0, 0.0, 0.0, INFINITY, -INFINITY};
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (t < k.ntiles) {
    const int tx = t % k.ntx, ty = (t / k.ntx) % k.nty, ch = t / (k.ntx * k.nty);
    const int x = tx * 128 + 2 * lane, y0 = ty * (4 * ROWS) + wave * ROWS;
    const bool lane_on = x < k.nx;
    const bool last = x + 2 >= k.nx || lane == 63;  // this lane's x+1 neighbour of its second cell is not in lane+1
    const int64_t plane = (int64_t)k.nx * k.ny;
    const uint32_t plane_bytes = (uint32_t)(plane * 8);
    uint32_t off[ROWS + 1], offx[ROWS];
    bool row_on[ROWS];
#pragma unroll
    for (int r = 0; r <= ROWS; ++r) {
      int y = y0 + r;
      const bool own = lane_on && y < k.ny;                          // this wave accumulates row r
      const bool on = lane_on && (r == 0 ? y < k.ny : y - 1 < k.ny);  // loaded if owned OR the y+1 row of an owned row
      if (y >= k.ny) y -= k.ny;                                       // then y == ny: the periodic wrap to row 0
      off[r] = on ? (uint32_t)(((int64_t)y * k.nx + x) * 8) : DOOB;
      if (r < ROWS) {
        row_on[r] = own;
        const int xn = x + 2 >= k.nx ? 0 : x + 2;
        offx[r] = (own && last) ? (uint32_t)(((int64_t)y * k.nx + xn) * 8) : DOOB;
      }
    }
    const int zs = ch * k.zchunk, ze = min(k.nz, zs + k.zchunk);
    auto zoff = [&](int p) -> int64_t { return (int64_t)((k.zwrap ? wrapi(p, k.nz) : p) + k.ghost) * plane; };
    double2 R[NS][ROWS + 1], PH[NS][ROWS];
    double XR[NS][ROWS];
    auto load_plane = [&](int slot, int p) {
      const bool valid = p <= ze;  // plane ze is the z+1 neighbour of the chunk's last plane
      const auto rc = dplane_rsrc(k.c + zoff(valid ? p : ze), valid ? plane_bytes : 0u);
#pragma unroll
      for (int r = 0; r <= ROWS; ++r) R[slot][r] = dbld2(rc, off[r]);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) XR[slot][r] = dbld1(rc, offx[r]);
      if constexpr (HAS_PHI) {
        const bool pv = p < ze;
        const auto rp = dplane_rsrc(k.phi + zoff(pv ? p : zs), pv ? plane_bytes : 0u);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) PH[slot][r] = dbld2(rp, off[r]);
      }
    };
#pragma unroll
    for (int d = 0; d < NS; ++d) load_plane(d, zs + d);
    for (int z0 = zs; z0 < ze; z0 += NS) {
#pragma unroll
      for (int d = 0; d < NS; ++d) {
        const int z = z0 + d;
        if (z < ze) {
          const int nslot = (d + 1) % NS;
          const double wz = zweight(z, k.nz, k.zends), wd = (z == k.nz - 1 && (k.zends & 2)) ? 0.0 : 1.0;
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const double2 cur = R[d][r], up = R[d][r + 1], nxt = R[nslot][r];
            double xr = __shfl_down(cur.x, 1, 64);
            if (last) xr = XR[d][r];
            if (row_on[r]) {
              double a = cur.x - k.ca, b = k.cb - cur.x, ab = a * b;
              double f = k.rho * (ab * ab);
              double dx = cur.y - cur.x, dy = up.x - cur.x, dz = nxt.x - cur.x;
              double g = wz * (dx * dx + dy * dy) + wd * (dz * dz);
              a = cur.y - k.ca;
              b = k.cb - cur.y;
              ab = a * b;
              f += k.rho * (ab * ab);
              dx = xr - cur.y;
              dy = up.y - cur.y;
              dz = nxt.y - cur.y;
              g += wz * (dx * dx + dy * dy) + wd * (dz * dz);
              v[0] += wz * (cur.x + cur.y);
              v[1] += wz * f;
              v[2] += g;
              if constexpr (HAS_PHI) v[3] += wz * (cur.x * PH[d][r].x + cur.y * PH[d][r].y);
              v[4] = fmin(v[4], fmin(cur.x, cur.y));
              v[5] = fmax(v[5], fmax(cur.x, cur.y));
            }
          }
          load_plane(d, z + NS);  // refill the slot just consumed: planes z+1 .. z+DEPTH stay in flight meanwhile
        }
      }
    }
  }
  block_reduce6(v, sh);
  if (threadIdx.x == 0)
    for (int q = 0; q < 6; ++q) k.partials[(int64_t)blockIdx.x * 6 + q] = v[q];
}

__global__ __launch_bounds__(DIAG_BLOCK) void diag_final_kernel(const double* __restrict__ partials, int nblocks,
                                                               double* __restrict__ out6) {
  __shared__ double sh[24];
  double v[6] = {0.0, 0.0, 0.0, 0.0, INFINITY, -INFINITY};
  for (int b = threadIdx.x; b < nblocks; b += DIAG_BLOCK) {
    for (int q = 0; q < 4; ++q) v[q] += partials[(int64_t)b * 6 + q];
    v[4] = fmin(v[4], partials[(int64_t)b * 6 + 4]);
    v[5] = fmax(v[5], partials[(int64_t)b * 6 + 5]);
  }
  block_reduce6(v, sh);
  if (threadIdx.x == 0)
    for (int q = 0; q < 6; ++q) out6[q] = v[q];
}

// mnx / mny > 0: the lattice is the even (mirror) extension of a no-flux domain with mnx x mny nodes:
// lattice index i >= mn maps to node 2 (mn - 1) - i.
__global__ __launch_bounds__(256) void ic_kernel(double* __restrict__ c, int nx, int ny, int nz, int ghost, double h,
                                                 double c0, double amp, double w0, int mnx, int mny) {
  const int64_t plane = (int64_t)nx * ny;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % nx), y = (int)(i / nx);
    const int xe = (mnx > 0 && x >= mnx) ? 2 * (mnx - 1) - x : x;
    const int ye = (mny > 0 && y >= mny) ? 2 * (mny - 1) - y : y;
    const double X = xe * h, Y = ye * h;
    const double t2 = cos(0.13 * X) * cos(0.087 * Y);
    const double v =
        c0 + amp * (cos(w0 * X) * cos(0.11 * Y) + t2 * t2 + cos(0.025 * X - 0.15 * Y) * cos(0.07 * X - 0.02 * Y));
    for (int z = 0; z < nz; ++z) c[(int64_t)(z + ghost) * plane + i] = v;
  }
}

// z-line decomposition of a no-flux box: the ghost planes outside a wall are the mirror images of the owned planes
// next to it (node-centred even extension: z = -k <- z = +k, z = nzg-1+k <- z = nzg-1-k, k = 1..ghost).
__global__ __launch_bounds__(256) void reflect_ghosts_kernel(double* __restrict__ buf, int64_t plane, int nz, int ghost,
                                                             int ends) {
  const int k = blockIdx.y % ghost + 1, top = blockIdx.y / ghost;
  if (!(ends & (1 << top))) return;
  const int64_t src = top ? (int64_t)(ghost + nz - 1 - k) * plane : (int64_t)(ghost + k) * plane;
  const int64_t dst = top ? (int64_t)(ghost + nz - 1 + k) * plane : (int64_t)(ghost - k) * plane;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (int64_t)gridDim.x * 256)
    buf[dst + i] = buf[src + i];
}

// plain device copy, 16 B per lane per access: the 1R + 1W HBM ceiling (pfk_stream_copy).
// CHUNKED = false: grid-stride (consecutive workgroups touch consecutive 4 KB); true: every workgroup owns one
// contiguous span.  U = independent 16-byte loads in flight per lane.
template <bool CHUNKED, int U>
__global__ __launch_bounds__(256) void stream_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst,
                                                          int64_t n2) {
  int64_t i, end, stride;
  if (CHUNKED) {
    const int64_t per = ((n2 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    i = (int64_t)blockIdx.x * per + threadIdx.x;
    end = min(n2, (int64_t)(blockIdx.x + 1) * per);
    stride = 256;
  } else {
    i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    end = n2;
    stride = (int64_t)gridDim.x * 256;
  }
  for (; i + (U - 1) * stride < end; i += U * stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < end; i += stride) dst[i] = src[i];
}

// Grid-wide barrier for persistent kernels (all workgroups co-resident: cooperative launch).  Every thread publishes
// its global writes at agent scope, one thread per workgroup counts in and polls; the spin is BOUNDED (a lost
// workgroup sets *abort_flag and lets everybody fall through instead of hanging the GPU).
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* abort_flag) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// probe: `iters` grid barriers, each preceded by one global write and followed by one read of a neighbour's slot
// (so the release / acquire cache maintenance has something real to do)
__global__ __launch_bounds__(256) void grid_barrier_probe_kernel(unsigned* counter, int* abort_flag, double* slots,
                                                                 int iters) {
  const int nb = gridDim.x;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    if (threadIdx.x < 32) slots[(int64_t)blockIdx.x * 32 + threadIdx.x] = (double)(it + blockIdx.x);
    grid_barrier(counter, (unsigned)(it + 1) * (unsigned)nb, abort_flag);
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    acc += slots[(int64_t)((blockIdx.x + 17) % nb) * 32 + (threadIdx.x & 31)];
  }
  if (threadIdx.x == 0) slots[(int64_t)nb * 32 + blockIdx.x] = acc;
}

// ---- peer-copy halo transport (pfk_push_planes / pfk_wait_flag) --------------------------------------------------
// push: copy n doubles into a (possibly peer-mapped, IPC) destination, then publish `seq` in the destination rank's
// flag word -- written by whichever workgroup finishes last, after a system-scope release, so a reader that sees
// flag >= seq also sees the planes.  `ticket` is a zero-initialised counter owned by the sender (reset for reuse).
__global__ __launch_bounds__(256) void push_planes_kernel(const double2* __restrict__ src, double2* __restrict__ dst,
                                                          int64_t n2, long long* flag, long long seq,
                                                          unsigned* ticket) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the stores above are visible to the peer
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// wait: one lane polls this rank's own flag word until the neighbour has published `seq`.  BOUNDED: after about two
// seconds it gives up, raises *timeout and lets the stream continue (a lost neighbour must not hang the GPU).
__global__ void wait_flag_kernel(const long long* flag, long long seq, int* timeout) {
  if (threadIdx.x != 0) return;
  for (int spins = 0; spins < (1 << 20); ++spins) {  // ~1-2 us per poll (a system-scope load) -> 1-2 s
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) return;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// signal: *flag = seq with system-scope release, stream-ordered behind the kernels that read what the peer may overwrite next
__global__ void signal_flag_kernel(long long* flag, long long seq) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one 16-byte element per thread, one short-lived workgroup per 4 KB: the dispatcher walks the buffer in address order
__global__ __launch_bounds__(256) void stream_copy_flat_kernel(const double2* __restrict__ src,
                                                               double2* __restrict__ dst, int64_t n2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) dst[i] = src[i];
}

// measured (tools/diag_ab.py, profiles/r02/diag_ab_*.log; wall per pf_diagnostics call incl. launch + read-back):
// 512^3 0.277 ms (round-1 kernel) -> 0.196 ms <4,2> at 2 workgroups per CU; 1024^3 2.215 -> 1.485 ms; BM6 (c and phi
// streams) 0.454 -> 0.349 ms with <2,1>
int g_diag_variant = -1 /* auto: <4,2>, with phi <2,1> */, g_diag_target_blocks = 512;
int g_copy_wgs_per_cu = 16, g_copy_mode = 5;  // flat: 6.2 TB/s vs 5.2-5.7 for the looping forms (profiles/r01/copy_sweep.txt)

}  // namespace

void set_copy_tuning(int wgs_per_cu, int mode) {
  if (wgs_per_cu > 0) g_copy_wgs_per_cu = wgs_per_cu;
  if (mode >= 0) g_copy_mode = mode;
}

int g_push_wgs = 16;  // one GPU, self-exchange beside the 512^3 stencil: 1 WG 0.97 ms/step, 4: 0.456, 8: 0.423, 16: 0.419, 64: 0.438
void set_push_wgs(int n) {
  if (n > 0) g_push_wgs = n;
}

hipError_t launch_push_planes(const double* src, double* dst, int64_t n, long long* flag, long long seq,
                              unsigned* ticket, hipStream_t stream) {
  const int64_t n2 = n / 2;
  int64_t nb = (n2 + 255) / 256;
  if (nb > g_push_wgs) nb = g_push_wgs;  // few workgroups: the message has a whole interior launch to arrive, and
                                          // every CU it borrows turns into a straggler of the 1-WG-per-CU stencil
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(push_planes_kernel, dim3((int)nb), dim3(256), 0, stream, reinterpret_cast<const double2*>(src),
                     reinterpret_cast<double2*>(dst), n2, flag, seq, ticket);
  return hipGetLastError();
}

hipError_t launch_signal_flag(long long* flag, long long seq, hipStream_t stream) {
  hipLaunchKernelGGL(signal_flag_kernel, dim3(1), dim3(64), 0, stream, flag, seq);
  return hipGetLastError();
}

hipError_t launch_wait_flag(const long long* flag, long long seq, int* timeout, hipStream_t stream) {
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, stream, flag, seq, timeout);
  return hipGetLastError();
}

hipError_t launch_stream_copy(const double* src, double* dst, int64_t n, hipStream_t stream) {
  const int64_t n2 = n / 2;
  int64_t nb = (n2 + 255) / 256;
  if (nb > 256 * (int64_t)g_copy_wgs_per_cu) nb = 256 * (int64_t)g_copy_wgs_per_cu;
  if (nb < 1) nb = 1;
  auto s2 = reinterpret_cast<const double2*>(src);
  auto d2 = reinterpret_cast<double2*>(dst);
  if (g_copy_mode == 5) {
    const int64_t nflat = (n2 + 255) / 256;
    if (nflat > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stream_copy_flat_kernel, dim3((unsigned)nflat), dim3(256), 0, stream, s2, d2, n2);
    return hipGetLastError();
  }
  switch (g_copy_mode) {
    case 1: hipLaunchKernelGGL((stream_copy_kernel<true, 4>), dim3((int)nb), dim3(256), 0, stream, s2, d2, n2); break;
    case 2: hipLaunchKernelGGL((stream_copy_kernel<false, 8>), dim3((int)nb), dim3(256), 0, stream, s2, d2, n2); break;
    case 3: hipLaunchKernelGGL((stream_copy_kernel<true, 8>), dim3((int)nb), dim3(256), 0, stream, s2, d2, n2); break;
    case 4: hipLaunchKernelGGL((stream_copy_kernel<false, 1>), dim3((int)nb), dim3(256), 0, stream, s2, d2, n2); break;
    default: hipLaunchKernelGGL((stream_copy_kernel<false, 4>), dim3((int)nb), dim3(256), 0, stream, s2, d2, n2);
  }
  return hipGetLastError();
}

// returns milliseconds per barrier in *ms_per_barrier; *ok = 0 if the bounded spin tripped
hipError_t run_grid_barrier_probe(int nblocks, int nthreads, int iters, double* ms_per_barrier, int* ok) {
  unsigned* counter = nullptr;
  int* flag = nullptr;
  double* slots = nullptr;
  hipError_t e;
  if ((e = pf_malloc(&counter, sizeof(unsigned))) != hipSuccess) return e;
  if ((e = pf_malloc(&flag, sizeof(int))) != hipSuccess) return e;
  if ((e = pf_malloc(&slots, sizeof(double) * ((size_t)nblocks * 33 + 64))) != hipSuccess) return e;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {  // second repetition is the timed one
    (void)hipMemset(counter, 0, sizeof(unsigned));
    (void)hipMemset(flag, 0, sizeof(int));
    void* args[] = {&counter, &flag, &slots, &iters};
    (void)hipEventRecord(e0, nullptr);
    e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(grid_barrier_probe_kernel), dim3(nblocks),
                                   dim3(nthreads), args, 0, nullptr);
    if (e != hipSuccess) break;
    (void)hipEventRecord(e1, nullptr);
    if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  int h_flag = 1;
  if (e == hipSuccess) e = hipMemcpy(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost);
  *ok = h_flag == 0;
  *ms_per_barrier = ms / (double)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)pf_free(counter);
  (void)pf_free(flag);
  (void)pf_free(slots);
  return e;
}

hipError_t launch_reflect_ghosts(double* buf, int64_t plane, int nz, int ghost, int ends, hipStream_t stream) {
  if (!ends) return hipSuccess;
  int64_t nb = (plane + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(reflect_ghosts_kernel, dim3((int)nb, 2 * ghost), dim3(256), 0, stream, buf, plane, nz, ghost,
                     ends);
  return hipGetLastError();
}

int diag_partials_elems() { return DIAG_MAX_BLOCKS * 6; }

// pfk_set_tuning keys 8 / 9: 10 ROWS + DEPTH of diag_stream2_kernel (0 = the round-1 kernel), target workgroups
void set_diag_tuning(int variant, int target_blocks) {
  if (variant >= 0) g_diag_variant = variant;
  if (target_blocks > 0) g_diag_target_blocks = target_blocks;
}

hipError_t launch_diag(const double* c, const double* phi, int nx, int ny, int nz, int ghost, int zwrap, int zends,
                       double rho, double ca, double cb, double* partials, double* out6, hipStream_t stream) {
  const int64_t total = (int64_t)nx * ny * nz;
  int64_t nb;
  const bool aligned = (nx % 2 == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0) &&
                       (phi == nullptr || (reinterpret_cast<uintptr_t>(phi) & 15) == 0);
  const int variant = g_diag_variant < 0 ? (phi ? 21 : 42) : g_diag_variant;
  if (aligned && variant > 0 && (int64_t)nx * ny * 8 < (int64_t)DOOB) {
    const int rows = variant / 10;  // variant = 10 ROWS + DEPTH (validated by pfk_set_tuning)
    DiagArgs k;
    k.c = c;
    k.phi = phi;
    k.nx = nx;
    k.ny = ny;
    k.nz = nz;
    k.ghost = ghost;
    k.zwrap = zwrap;
    k.zends = zends;
    k.ntx = (nx + 127) / 128;
    k.nty = (ny + 4 * rows - 1) / (4 * rows);
    const int64_t xy = (int64_t)k.ntx * k.nty;
    int nchunk = (int)((g_diag_target_blocks + xy - 1) / xy);
    if (nchunk > nz) nchunk = nz;
    if (nchunk < 1) nchunk = 1;
    while (xy * nchunk > DIAG_MAX_BLOCKS - 8 && nchunk > 1) --nchunk;
    k.zchunk = (nz + nchunk - 1) / nchunk;
    nchunk = (nz + k.zchunk - 1) / k.zchunk;
    k.ntiles = (int)(xy * nchunk);
    k.rho = rho;
    k.ca = ca;
    k.cb = cb;
    k.partials = partials;
    nb = ((xy * nchunk + 7) / 8) * 8;  // XCD-aware order needs a multiple of 8; surplus blocks write neutral partials
    if (nb <= DIAG_MAX_BLOCKS) {
#define PF_DIAG_LAUNCH(RW, DP)                                                                                        \
  do {                                                                                                                \
    if (phi)                                                                                                          \
      hipLaunchKernelGGL((diag_stream2_kernel<RW, DP, true>), dim3((int)nb), dim3(DIAG_BLOCK), 0, stream, k);         \
    else                                                                                                              \
      hipLaunchKernelGGL((diag_stream2_kernel<RW, DP, false>), dim3((int)nb), dim3(DIAG_BLOCK), 0, stream, k);        \
  } while (0)
      switch (variant) {
        case 21: PF_DIAG_LAUNCH(2, 1); break;
        case 22: PF_DIAG_LAUNCH(2, 2); break;
        case 23: PF_DIAG_LAUNCH(2, 3); break;
        case 41: PF_DIAG_LAUNCH(4, 1); break;
        case 43: PF_DIAG_LAUNCH(4, 3); break;
        case 12: PF_DIAG_LAUNCH(1, 2); break;
        case 13: PF_DIAG_LAUNCH(1, 3); break;
        default: PF_DIAG_LAUNCH(4, 2); break;
      }
#undef PF_DIAG_LAUNCH
      hipLaunchKernelGGL(diag_final_kernel, dim3(1), dim3(DIAG_BLOCK), 0, stream, (const double*)partials, (int)nb,
                         out6);
      return hipGetLastError();
    }
  }
  if (aligned) {
    const int ntx = (nx + 127) / 128, nty = (ny + DS_ROWS - 1) / DS_ROWS;
    const int64_t xy = (int64_t)ntx * nty;
    int nchunk = (int)((2048 + xy - 1) / xy);  // aim at ~2048 blocks
    if (nchunk > nz) nchunk = nz;
    if (nchunk < 1) nchunk = 1;
    while (xy * nchunk > DIAG_MAX_BLOCKS && nchunk > 1) --nchunk;
    const int zchunk = (nz + nchunk - 1) / nchunk;
    nchunk = (nz + zchunk - 1) / zchunk;
    nb = xy * nchunk;
    if (nb <= DIAG_MAX_BLOCKS) {
      hipLaunchKernelGGL(diag_stream_kernel, dim3((int)nb), dim3(DIAG_BLOCK), 0, stream, c, phi, nx, ny, nz, ghost,
                         zwrap, zends, ntx, nty, zchunk, rho, ca, cb, partials);
      hipLaunchKernelGGL(diag_final_kernel, dim3(1), dim3(DIAG_BLOCK), 0, stream, (const double*)partials, (int)nb,
                         out6);
      return hipGetLastError();
    }
  }
  nb = (total + DIAG_BLOCK - 1) / DIAG_BLOCK;
  if (nb > DIAG_MAX_BLOCKS) nb = DIAG_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(diag_partial_kernel, dim3((int)nb), dim3(DIAG_BLOCK), 0, stream, c, phi, nx, ny, nz, ghost, zwrap,
                     zends, rho, ca, cb, partials);
  hipLaunchKernelGGL(diag_final_kernel, dim3(1), dim3(DIAG_BLOCK), 0, stream, (const double*)partials, (int)nb, out6);
  return hipGetLastError();
}

hipError_t launch_ic(double* c, int nx, int ny, int nz, int ghost, double h, double c0, double amp, double w0,
                     int mnx, int mny, hipStream_t stream) {
  const int64_t plane = (int64_t)nx * ny;
  int64_t nb = (plane + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(ic_kernel, dim3((int)nb), dim3(256), 0, stream, c, nx, ny, nz, ghost, h, c0, amp, w0, mnx,
                     mny);
  return hipGetLastError();
}

}  // namespace pfhip
