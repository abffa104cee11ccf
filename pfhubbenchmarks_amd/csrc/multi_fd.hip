// Explicit finite-difference schemes for the multi-field benchmarks (round 2):
//
//   BM2  dolfin/bench2.py:76-111   c_t = M lap(mu), mu = df/dc - kappa_c lap c;  eta_i,t = -L (df/deta_i - kappa_eta lap eta_i)
//        f = f_alpha (1 - h) + f_beta h + w g   (hinterp / double_well, bench2.py:76-96)
//   BM3  dolfin/bench3.py:66-97    tau phi_t = W^2 lap phi + dfdp,  dfdp = (phi - lam U (1 - phi^2)) (1 - phi^2);
//        U_t = D lap U + phi_t / 2                                   (anisotropy switched off there: a = 1)
//
// Same conventions as the BM1 / BM6 grid path: fp64, x fastest, periodic lattice (the reference's no-flux boxes run on
// their even extension), 5-point (2-D) / 7-point (3-D) lap_h, forward Euler, forward-difference discrete energy.  Fields
// are stored as one structure-of-arrays block u[f][cell] per time level (ping-pong = rollback state).  These are plain
// one-thread-per-cell kernels (neighbours through the caches): the benchmark problems are 200^2 / 960^2 cells, far from
// HBM-bound; the LDS-tiled streaming design of ch_fd_kernels.hip is what a 512^3 BM2 would take next.
// The operation order below is restated by oracle/multi_fd.py (numpy, no fma: this file is compiled with
// -ffp-contract=off) and compared bit for bit.
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "pfhip_internal.h"

namespace pfhip {

namespace {

struct MfdParams {
  int model, nf, nx, ny, nz;
  int gz;   // slab mode: ghost planes per side inside nz (refreshed by the caller before every step); diagnostics skip them
  int zlo = 0, zhi = 0;   // planes [zlo, zhi) to compute in this launch (multifd_step_range; a whole step: [0, nz))
  int64_t fs = 0;         // field stride in doubles: nx ny nz for the caller's buffers, padded for the library's own (multifd_create)
  double inv_h2;
  // BM2: ca, cb, rho2, kappa_c, M, kappa_eta, w, alpha, L     BM3: lam, 1/tau, W^2, D
  double q[9];
};

__device__ __forceinline__ int wrapm(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// lap_h u (times h^2) at cell (x, y, z): ((u[x-1] + u[x+1]) + (u[y-1] + u[y+1])) - 4 u  [+ ((u[z-1] + u[z+1]) - 2 u)]
__device__ __forceinline__ double lap_raw(const double* __restrict__ u, int x, int y, int z, int nx, int ny, int nz) {
  const int64_t row = (int64_t)nx, plane = (int64_t)nx * ny;
  const int64_t zc = z * plane, yc = y * row;
  const double c = u[zc + yc + x];
  const double sx = u[zc + yc + wrapm(x - 1, nx)] + u[zc + yc + wrapm(x + 1, nx)];
  const double sy = u[zc + wrapm(y - 1, ny) * row + x] + u[zc + wrapm(y + 1, ny) * row + x];
  double l = (sx + sy) - 4.0 * c;
  if (nz > 1) l = l + ((u[wrapm(z - 1, nz) * plane + yc + x] + u[wrapm(z + 1, nz) * plane + yc + x]) - 2.0 * c);
  return l;
}

__device__ __forceinline__ double hs(double u) { return ((u * u) * u) * ((6.0 * (u * u) - 15.0 * u) + 10.0); }
__device__ __forceinline__ double hsp(double u) { return (30.0 * (u * u)) * ((1.0 - u) * (1.0 - u)); }

// BM2 pass 1: mu = f_c - kappa_c inv_h2 lap_raw(c)
__global__ __launch_bounds__(256) void bm2_mu_kernel(const MfdParams p, const double* __restrict__ u,
                                                     double* __restrict__ mu) {
  const int64_t i = (int64_t)p.zlo * p.nx * p.ny + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)p.zhi * p.nx * p.ny) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3];
  const double c = u[i];
  double h = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) h = h + hs(u[(k + 1) * p.fs + i]);
  const double fc = (2.0 * r2) * (c - ca) * (1.0 - h) + (2.0 * r2) * (c - cb) * h;
  mu[i] = fc - (kc * p.inv_h2) * lap_raw(u, x, y, z, p.nx, p.ny, p.nz);
}

// BM2 pass 2: c+ = c + (dt M inv_h2) lap_raw(mu);  eta+ = eta - (dt L) (f_eta - kappa_eta inv_h2 lap_raw(eta))
__global__ __launch_bounds__(256) void bm2_update_kernel(const MfdParams p, const double* __restrict__ u,
                                                         const double* __restrict__ mu, double* __restrict__ un,
                                                         double dt) {
  const int64_t i = (int64_t)p.zlo * p.nx * p.ny + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)p.zhi * p.nx * p.ny) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], Mob = p.q[4], ke = p.q[5], w = p.q[6], al = p.q[7], L = p.q[8];
  const double c = u[i];
  un[i] = c + (dt * Mob * p.inv_h2) * lap_raw(mu, x, y, z, p.nx, p.ny, p.nz);
  double e[4], e2 = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] = u[(k + 1) * p.fs + i];
    e2 = e2 + e[k] * e[k];
  }
  const double dfab = r2 * ((c - cb) * (c - cb)) - r2 * ((c - ca) * (c - ca));  // f_beta - f_alpha
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double ek = e[k];
    const double well = (2.0 * ek * ((1.0 - ek) * (1.0 - ek)) - 2.0 * (ek * ek) * (1.0 - ek)) + (2.0 * al) * ek * (e2 - ek * ek);
    const double fe = dfab * hsp(ek) + w * well;
    const double lap = lap_raw(u + (k + 1) * p.fs, x, y, z, p.nx, p.ny, p.nz);
    un[(k + 1) * p.fs + i] = ek - (dt * L) * (fe - (ke * p.inv_h2) * lap);
  }
}

// BM3: phi_t = (1/tau) (W^2 inv_h2 lap_raw(phi) + dfdp);  phi+ = phi + dt phi_t;  U+ = U + dt (D inv_h2 lap_raw(U) + phi_t / 2)
__global__ __launch_bounds__(256) void bm3_update_kernel(const MfdParams p, const double* __restrict__ u,
                                                         double* __restrict__ un, double dt) {
  const int64_t i = (int64_t)p.zlo * p.nx * p.ny + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)p.zhi * p.nx * p.ny) return;
  const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
  const double lam = p.q[0], it = p.q[1], W2 = p.q[2], D = p.q[3];
  const double U = u[i], ph = u[p.fs + i];
  const double P = 1.0 - ph * ph;
  const double dfdp = (ph - (lam * U) * P) * P;
  const double pt = it * ((W2 * p.inv_h2) * lap_raw(u + p.fs, x, y, z, p.nx, p.ny, p.nz) + dfdp);
  un[p.fs + i] = ph + dt * pt;
  un[i] = U + dt * ((D * p.inv_h2) * lap_raw(u, x, y, z, p.nx, p.ny, p.nz) + 0.5 * pt);
}

// =====================================================================================================================
// Streaming ("2.5-D") forms of the three passes above for 3-D boxes whose x extent is a multiple of 128 (and y of the tile
// height): the design of the BM1 stencil kernel (ch_fd_kernels.hip) applied to the multi-field models.  A workgroup of 4
// waves owns a 128 x (4 RPT) column of cells and walks a z-chunk plane by plane:
//   * a lane owns 2 x-adjacent cells (16-byte global loads / stores: full 128-byte lines per quarter wave) in RPT rows;
//   * the z-neighbours of a cell are ITS OWN values in the planes before / after: three planes per stencilled field live
//     in registers and rotate, so every input plane is read from HBM once per workgroup;
//   * the x / y neighbours of plane z come from an LDS tile per stencilled field (own cells + a 1-cell halo: the halo
//     rows are one extra 16-byte load for two of the waves, the halo columns 2 scalars per row for a third) -- a halo
//     cell needs no z-pipeline of its own because every pass here is a single radius-1 stage (that is what keeps this
//     kernel so much simpler than the fused two-stage BM1 kernel; the price is BM2's separate mu pass);
//   * same arithmetic, same operation order as the one-thread-per-cell kernels above (bit-identical; they stay the path
//     for 2-D problems and for extents that do not tile).
// PASS 30: BM3 update (U, phi -> U, phi: 32 B/cell).  PASS 20: BM2 mu pass (c with stencil, 4 eta pointwise -> mu: 48 B).
// PASS 21: BM2 update (mu and 4 eta with stencil, c pointwise -> 5 fields: 88 B).
// output planes are never re-read inside the launch: non-temporal stores keep them from evicting the halo lines the
// neighbour tiles are about to read from the XCD's L2 (the BM1 kernel gained 2-8 % from the same); NT is a
// template switch for the in-process A/B (pfk_set_tuning key 10)
template <bool NT>
__device__ __forceinline__ void st_out(double* p, double2 v) {
  if (NT) {
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
  } else {
    *reinterpret_cast<double2*>(p) = v;
  }
}
constexpr int SX = 128, SPITCH = 132;   // tile width; LDS row pitch (own cells at 2..129, halo at 1 and 130)

template <int PASS>
struct PassTraits;
template <>
struct PassTraits<30> { static constexpr int NS = 2, NPW = 0, RPT = 4, MINW = 1, AHEAD = 2; };   // (AHEAD = 3: 0.6905 vs 0.6942 ms at 254 VGPRs + AGPRs -- not worth the registers)   // stencilled fields, pointwise inputs, rows per thread, waves per SIMD asked of the register allocator
template <>
struct PassTraits<20> { static constexpr int NS = 1, NPW = 4, RPT = 2, MINW = 1, AHEAD = 1; };
template <>
struct PassTraits<21> { static constexpr int NS = 5, NPW = 1, RPT = 1, MINW = 1, AHEAD = 1; };

template <int PASS, bool NT>
__global__ __launch_bounds__(256, PassTraits<PASS>::MINW) void mfd_stream_kernel(const MfdParams p, const double* __restrict__ u,
                                                         const double* __restrict__ mu_in, double* __restrict__ out,
                                                         double dt, int zchunk) {
  using T = PassTraits<PASS>;
  constexpr int NS = T::NS, NPW = T::NPW, RPT = T::RPT, TY = 4 * RPT, AH = T::AHEAD;   // AH: planes between request and use
  __shared__ __attribute__((aligned(16))) double tile[NS][TY + 2][SPITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware tile order (as in bm2_fused_kernel below): every XCD gets a contiguous run of tiles, so that the halo rows a
  // tile reads are the own rows of a tile on the same L2
  const int ntx = p.nx / SX, nty = p.ny / TY;
  const int nb = gridDim.x, per = (nb + 7) / 8, full = nb % 8;
  const int xq = blockIdx.x % 8, rq = blockIdx.x / 8;
  const int tile_id = full == 0 ? xq * per + rq : ((xq < full ? xq * per : full * per + (xq - full) * (per - 1)) + rq);
  const int x0 = (tile_id % ntx) * SX, y0 = ((tile_id / ntx) % nty) * TY;
  const int zb = p.zlo + (tile_id / (ntx * nty)) * zchunk, ze = zb + zchunk < p.zhi ? zb + zchunk : p.zhi;
  const int64_t row = p.nx, plane = (int64_t)p.nx * p.ny;
  const int xo = x0 + 2 * lane;
  // stencilled field s of this pass -> where it lives
  auto sfield = [&](int s) -> const double* {
    if (PASS == 30) return u + (int64_t)s * p.fs;            // U, phi
    if (PASS == 20) return u;                                   // c
    return s == 0 ? mu_in : u + (int64_t)s * p.fs;             // mu, eta1..eta4
  };
  auto pfield = [&](int k) -> const double* { return PASS == 20 ? u + (int64_t)(k + 1) * p.fs : u; };  // eta_k / c
  double2 zm[NS][RPT], zc[NS][RPT], zp[NS][RPT];   // own cells in planes z-1, z, z+1
  double2 zq[NS][RPT], zr[NS][RPT];                // AH >= 2: plane z+2 (AH == 3: and z+3), in flight
  const int64_t own = (int64_t)(y0 + wave * RPT) * row + xo;
#define MFD_LOAD_OWN(DST, Z)                                                                                     \
  {                                                                                                              \
    const int64_t zo_ = (int64_t)wrapm((Z), p.nz) * plane + own;                                                 \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) _Pragma("unroll") for (int r = 0; r < RPT; ++r)               \
        DST[s][r] = *reinterpret_cast<const double2*>(sfield(s) + zo_ + (int64_t)r * row);                       \
  }
  MFD_LOAD_OWN(zm, zb - 1)
  MFD_LOAD_OWN(zc, zb)
  // halo of a plane: rows y0 - 1 and y0 + TY (waves 0 and 1), columns x0 - 1 and x0 + 128 (wave 2), periodic wrap
  // (every wave forms both addresses -- clamped to something valid -- and loads under a predicate: arrays that are
  // written in one branch only end up in scratch memory).  The halo of plane z + 1 is requested TOGETHER with the own
  // cells of plane z + 1, one iteration before it is used: a halo row is an own row of the y-neighbour tile, which
  // requests it for ITS plane z + 1 at that moment, so the two requests meet in the XCD's L2.  Requested one plane later
  // (with the own cells of z + 2, the round-3 form) the line had already left the 4 MiB L2 -- one plane step of an XCD's
  // 128 tiles is 4 MB -- and came from HBM a second time: 1.52x the input bytes (profiles/r03/summary_bm3_fd_512c.json).
  const bool do_row = wave < 2, do_col = wave == 2 && lane < 2 * TY;
  const int yh = wrapm(wave == 0 ? y0 - 1 : y0 + TY, p.ny);
  const int yy = y0 + (do_col ? (lane >> 1) : 0), xh = wrapm((lane & 1) ? x0 + SX : x0 - 1, p.nx);
  const int64_t arow0 = (int64_t)yh * row + xo, acol0 = (int64_t)yy * row + xh;
  double2 hrow[NS], hrow_n[NS], hrow_q[NS], hrow_r[NS];
  double hcol[NS], hcol_n[NS], hcol_q[NS], hcol_r[NS];
#define MFD_LOAD_HALO(HR, HC, Z)                                                                                 \
  {                                                                                                              \
    const int64_t zo_ = (int64_t)wrapm((Z), p.nz) * plane;                                                       \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                             \
      HR[s] = do_row ? *reinterpret_cast<const double2*>(sfield(s) + zo_ + arow0) : make_double2(0.0, 0.0);      \
      HC[s] = do_col ? sfield(s)[zo_ + acol0] : 0.0;                                                             \
    }                                                                                                            \
  }
  MFD_LOAD_HALO(hrow, hcol, zb)
  if (AH >= 2) {
    MFD_LOAD_OWN(zp, zb + 1)
    MFD_LOAD_HALO(hrow_n, hcol_n, zb + 1)
  }
  if (AH == 3) {
    MFD_LOAD_OWN(zq, zb + 2)
    MFD_LOAD_HALO(hrow_q, hcol_q, zb + 2)
  }
  for (int z = zb; z < ze; ++z) {
    if (AH == 3) {
      MFD_LOAD_OWN(zr, z + 3)
      MFD_LOAD_HALO(hrow_r, hcol_r, z + 3)
    } else if (AH == 2) {   // one workgroup per CU (launch_stream): two planes in flight keep the memory pipe as full as four waves can
      MFD_LOAD_OWN(zq, z + 2)
      MFD_LOAD_HALO(hrow_q, hcol_q, z + 2)
    } else {
      MFD_LOAD_OWN(zp, z + 1)
      MFD_LOAD_HALO(hrow_n, hcol_n, z + 1)
    }
    double2 pw[NPW > 0 ? NPW : 1][RPT];
#pragma unroll
    for (int k = 0; k < NPW; ++k)
#pragma unroll
      for (int r = 0; r < RPT; ++r)
        pw[k][r] = *reinterpret_cast<const double2*>(pfield(k) + z * plane + (int64_t)(y0 + wave * RPT + r) * row + xo);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) *reinterpret_cast<double2*>(&tile[s][1 + wave * RPT + r][2 + 2 * lane]) = zc[s][r];
      if (do_row) *reinterpret_cast<double2*>(&tile[s][wave == 0 ? 0 : TY + 1][2 + 2 * lane]) = hrow[s];
      if (do_col) tile[s][1 + (lane >> 1)][(lane & 1) ? SX + 2 : 1] = hcol[s];
    }
    __syncthreads();
    // raw 7-point Laplacians (times h^2) of the stencilled fields at the two own cells of each row
    double2 lap[NS][RPT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int ty = 1 + wave * RPT + r, tx = 2 + 2 * lane;
        const double xl = tile[s][ty][tx - 1], xr = tile[s][ty][tx + 2];
        const double2 ym = *reinterpret_cast<const double2*>(&tile[s][ty - 1][tx]);
        const double2 yp = *reinterpret_cast<const double2*>(&tile[s][ty + 1][tx]);
        const double2 c = zc[s][r];
        double2 l;
        l.x = ((xl + c.y) + (ym.x + yp.x)) - 4.0 * c.x;
        l.y = ((c.x + xr) + (ym.y + yp.y)) - 4.0 * c.y;
        l.x = l.x + ((zm[s][r].x + zp[s][r].x) - 2.0 * c.x);
        l.y = l.y + ((zm[s][r].y + zp[s][r].y) - 2.0 * c.y);
        lap[s][r] = l;
      }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int64_t o = z * plane + (int64_t)(y0 + wave * RPT + r) * row + xo;
      if (PASS == 30) {
        const double lam = p.q[0], it = p.q[1], W2 = p.q[2], D = p.q[3];
        double2 nU, nP;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const double U = h2 ? zc[0][r].y : zc[0][r].x, ph = h2 ? zc[1][r].y : zc[1][r].x;
          const double lU = h2 ? lap[0][r].y : lap[0][r].x, lP = h2 ? lap[1][r].y : lap[1][r].x;
          const double P = 1.0 - ph * ph;
          const double dfdp = (ph - (lam * U) * P) * P;
          const double pt = it * ((W2 * p.inv_h2) * lP + dfdp);
          const double np_ = ph + dt * pt, nu_ = U + dt * ((D * p.inv_h2) * lU + 0.5 * pt);
          if (h2) {
            nP.y = np_;
            nU.y = nu_;
          } else {
            nP.x = np_;
            nU.x = nu_;
          }
        }
        st_out<NT>(out + o, nU);
        st_out<NT>(out + p.fs + o, nP);
      } else if (PASS == 20) {
        const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3];
        double2 m;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const double c = h2 ? zc[0][r].y : zc[0][r].x;
          double h = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) h = h + hs(h2 ? pw[k][r].y : pw[k][r].x);
          const double fc = (2.0 * r2) * (c - ca) * (1.0 - h) + (2.0 * r2) * (c - cb) * h;
          const double v = fc - (kc * p.inv_h2) * (h2 ? lap[0][r].y : lap[0][r].x);
          if (h2)
            m.y = v;
          else
            m.x = v;
        }
        *reinterpret_cast<double2*>(out + o) = m;
      } else {
        const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], Mob = p.q[4], ke = p.q[5], w = p.q[6], al = p.q[7], L = p.q[8];
        double2 res[5];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const double c = h2 ? pw[0][r].y : pw[0][r].x;
          const double nc = c + (dt * Mob * p.inv_h2) * (h2 ? lap[0][r].y : lap[0][r].x);
          double e[4], e2 = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            e[k] = h2 ? zc[k + 1][r].y : zc[k + 1][r].x;
            e2 = e2 + e[k] * e[k];
          }
          const double dfab = r2 * ((c - cb) * (c - cb)) - r2 * ((c - ca) * (c - ca));
          if (h2)
            res[0].y = nc;
          else
            res[0].x = nc;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double ek = e[k];
            const double well = (2.0 * ek * ((1.0 - ek) * (1.0 - ek)) - 2.0 * (ek * ek) * (1.0 - ek)) + (2.0 * al) * ek * (e2 - ek * ek);
            const double fe = dfab * hsp(ek) + w * well;
            const double ne = ek - (dt * L) * (fe - (ke * p.inv_h2) * (h2 ? lap[k + 1][r].y : lap[k + 1][r].x));
            if (h2)
              res[k + 1].y = ne;
            else
              res[k + 1].x = ne;
          }
        }
#pragma unroll
        for (int f = 0; f < 5; ++f) st_out<NT>(out + (int64_t)f * p.fs + o, res[f]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        zm[s][r] = zc[s][r];
        zc[s][r] = zp[s][r];
        if (AH >= 2) zp[s][r] = zq[s][r];
        if (AH == 3) zq[s][r] = zr[s][r];
      }
      hrow[s] = hrow_n[s];
      hcol[s] = hcol_n[s];
      if (AH >= 2) {
        hrow_n[s] = hrow_q[s];
        hcol_n[s] = hcol_q[s];
      }
      if (AH == 3) {
        hrow_q[s] = hrow_r[s];
        hcol_q[s] = hcol_r[s];
      }
    }
  }
#undef MFD_LOAD_OWN
#undef MFD_LOAD_HALO
}

// =====================================================================================================================
// BM2 in ONE pass: c and the four order parameters read once, written once -- 80 B per cell update, the algorithmic
// minimum (the two streaming passes above move 136 B: mu goes through HBM).  c is a two-stage update (mu = f_c - kappa
// lap c, then c += dt M lap mu), so mu of plane z is needed on the tile AND on a 1-cell ring around it; the ring cells get
// no z-pipeline in registers -- instead the last three c planes stay in an LDS ring (own cells + a 2-cell halo), from which
// any thread can form lap c at a ring cell, and mu lives in two LDS planes (z, z + 1).  The order parameters are single
// radius-1 stages: own z-neighbours in registers, x / y neighbours from one LDS tile per field, as in mfd_stream_kernel.
// Workgroup = 8 waves on a 128 x 8 tile (one row per wave, 2 x-adjacent cells per lane: 16-byte accesses), one workgroup
// per CU (99 KB of LDS), walking its z-chunk with the inputs of plane z + 2 / z + 3 already in flight.  Per plane:
//   P1  c(z+2) (own + halo, prefetched)            -> LDS ring                                            | barrier
//   P2  mu(z+1) on tile + ring from the ring planes z, z+1, z+2 and h(eta(z+1))  -> LDS mu plane; prefetch c(z+3), eta(z+2) | barrier
//   P3  c+(z) = c + dt M lap mu(z);  eta_k+(z) from the eta(z) tiles and the registers; 5 stores          | barrier
//   P4  eta(z+1) (own + halo) -> LDS tiles; rotate registers   (eta(z+2), requested in P2, is taken over in the next P2)
// Halo work by role: waves 0-3 load one halo row of c each, wave 4 its 4 halo columns, waves 5 / 6 the eta halo rows (and
// compute mu on the ring rows), wave 7 the eta halo columns (and mu on the ring columns).  Two warm-up iterations per
// z-chunk fill the pipeline (outputs suppressed).  Same arithmetic and operation order as bm2_mu_kernel /
// bm2_update_kernel: bit-identical results.
constexpr int B2TY = 8, B2T = 512;
struct Bm2Lds {
  double cc[3][B2TY + 4][SPITCH];
  double mu[2][B2TY + 2][SPITCH];
  double et[4][B2TY + 2][SPITCH];
};

template <bool NT>
__global__ __launch_bounds__(B2T) void bm2_fused_kernel(const MfdParams p, const double* __restrict__ u,
                                                       double* __restrict__ un, double dt, int zchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Bm2Lds& S = *reinterpret_cast<Bm2Lds*>(smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2); give every XCD a
  // contiguous run of tiles (x fastest, then y, then z-chunk) so that the halo rows a tile reads are the own rows of a
  // tile on the SAME L2 -- with the plain blockIdx order y-neighbours sit 4 apart, i.e. on different XCDs, and every halo
  // row comes from HBM a second time (measured: 1.3x the algorithmic traffic)
  const int ntx = p.nx / SX, nty = p.ny / B2TY;
  const int nb = gridDim.x, per = (nb + 7) / 8, full = nb % 8;
  const int xq = blockIdx.x % 8, rq = blockIdx.x / 8;
  const int tile = full == 0 ? xq * per + rq : ((xq < full ? xq * per : full * per + (xq - full) * (per - 1)) + rq);
  const int bx = tile % ntx, by = (tile / ntx) % nty, bz = tile / (ntx * nty);
  const int x0 = bx * SX, y0 = by * B2TY;
  const int zb = p.zlo + bz * zchunk, ze = zb + zchunk < p.zhi ? zb + zchunk : p.zhi;
  const int64_t row = p.nx, plane = (int64_t)p.nx * p.ny;
  const int xo = x0 + 2 * lane, tx = 2 + 2 * lane;
  const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3], Mob = p.q[4], ke = p.q[5], w = p.q[6], al = p.q[7], L = p.q[8];
  auto zw = [&](int z) { return z < 0 ? z + p.nz : (z >= p.nz ? z - p.nz : z); };  // z in [-nz, 2 nz): no integer division
  // ---- roles (every wave forms every address, clamped to something valid, and loads under a predicate) ----
  const bool c_row = wave < 4, c_col = wave == 4 && lane < 4 * (B2TY + 2), e_row = wave == 5 || wave == 6,
             e_col = wave == 7 && lane < 2 * B2TY;
  // c halo row of this wave: y0-2, y0-1, y0+TY, y0+TY+1 -> ring rows 0, 1, TY+2, TY+3
  const int chr = wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 2 ? B2TY + 2 : B2TY + 3));
  const int a_crow = (int)wrapm(y0 - 2 + chr, p.ny) * row + xo;
  // c halo columns: lane -> (ring row 1 .. TY+2, one of x0-2, x0-1, x0+128, x0+129)
  const int ccr = 1 + (c_col ? lane >> 2 : 0), cck = lane & 3;
  const int ccx = cck == 0 ? 0 : (cck == 1 ? 1 : (cck == 2 ? SX + 2 : SX + 3));   // column index in the ring row
  const int a_ccol = (int)wrapm(y0 - 2 + ccr, p.ny) * row + wrapm(x0 - 2 + ccx, p.nx);
  // eta halo row (wave 5: y0-1 -> tile row 0; wave 6: y0+TY -> tile row TY+1) and columns (wave 7)
  const int ehr = wave == 5 ? 0 : B2TY + 1;
  const int a_erow = (int)wrapm(y0 - 1 + ehr, p.ny) * row + xo;
  const int ecr = 1 + (e_col ? lane >> 1 : 0), ecx = (lane & 1) ? SX + 2 : 1;
  const int a_ecol = (int)wrapm(y0 - 1 + ecr, p.ny) * row + wrapm(x0 - 2 + ecx, p.nx);
  const int a_own = (int)(y0 + wave) * row + xo;
  const int oy = wave + 2;  // own row in the c ring; own row in the mu / eta tiles is wave + 1

  double2 e[4][3], eh[4], peh[4];          // eta own: planes z-1, z, z+1; halo of z+1 / prefetched z+2 (.x only for columns)
  double2 mu_m1, pch;                      // mu own of plane z-1 (planes z, z+1 are re-read from the LDS mu planes); c halo of the plane about to enter the ring
  // Prefetch registers: own c of the plane about to enter the ring and own eta, requested one plane step ahead.  The halos
  // of the same step are requested BEFORE the own cells: loads return in order, so waiting for a halo that was issued after
  // the own-cell requests would drain those too -- halo first, and the wait for it leaves the five own-cell requests in
  // flight.  Together with the mu planes re-read from LDS instead of kept in registers and 32-bit in-plane offsets (214 VGPRs,
  // no spill) this took the step from 2.19 to 2.00 ms (0.61 -> 0.67 of peak; round 4, the three changes were made together).  A second register stage (own cells two
  // steps ahead, loop unrolled by two) needs 256 VGPRs + spills, and a scratch reload waits for every older load: 2.37 ms.
  double2 pc, pe[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) e[k][j] = make_double2(0.0, 0.0);
  mu_m1 = make_double2(0.0, 0.0);

#define B2_LOAD_C_OWN(PC, Z) PC = *reinterpret_cast<const double2*>(u + (int64_t)zw(Z) * plane + a_own);
#define B2_LOAD_C_HALO(Z)                                                                                       \
  {                                                                                                             \
    const double* cz_ = u + (int64_t)zw(Z) * plane;                                                             \
    pch = c_row ? *reinterpret_cast<const double2*>(cz_ + a_crow) : make_double2(c_col ? cz_[a_ccol] : 0.0, 0.0); \
  }
#define B2_STORE_C(PC, Z)                                                                                       \
  {                                                                                                             \
    const int sl_ = ((Z) + 3) % 3;                                                                              \
    *reinterpret_cast<double2*>(&S.cc[sl_][oy][tx]) = PC;                                                       \
    if (c_row) *reinterpret_cast<double2*>(&S.cc[sl_][chr][tx]) = pch;                                          \
    if (c_col) S.cc[sl_][ccr][ccx] = pch.x;                                                                     \
  }
#define B2_LOAD_E_OWN(DST, Z)                                                                                   \
  {                                                                                                             \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                               \
        DST[k] = *reinterpret_cast<const double2*>(u + (int64_t)(k + 1) * p.fs + (int64_t)zw(Z) * plane + a_own); \
  }
#define B2_LOAD_E_HALO(DSTH, Z)                                                                                 \
  {                                                                                                             \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                             \
      const double* ez_ = u + (int64_t)(k + 1) * p.fs + (int64_t)zw(Z) * plane;                                \
      DSTH[k] = e_row ? *reinterpret_cast<const double2*>(ez_ + a_erow) : make_double2(e_col ? ez_[a_ecol] : 0.0, 0.0); \
    }                                                                                                           \
  }
  // mu at one cell from the ring (planes P-1, P, P+1 in slots sm, sc, sp), the cell's neighbours and h = sum hs(eta_k(P))
  auto mu_cell = [&](double c, double xl, double xr, double ym, double yp, double zm, double zp, double h) {
    double l = ((xl + xr) + (ym + yp)) - 4.0 * c;
    l = l + ((zm + zp) - 2.0 * c);
    const double fc = (2.0 * r2) * (c - ca) * (1.0 - h) + (2.0 * r2) * (c - cb) * h;
    return fc - (kc * p.inv_h2) * l;
  };
  auto mu_pair = [&](int sm, int sc, int sp, int ry, const double2 (&en)[4]) {
    const double2 c = *reinterpret_cast<const double2*>(&S.cc[sc][ry][tx]);
    const double xl = S.cc[sc][ry][tx - 1], xr = S.cc[sc][ry][tx + 2];
    const double2 ym = *reinterpret_cast<const double2*>(&S.cc[sc][ry - 1][tx]);
    const double2 yp = *reinterpret_cast<const double2*>(&S.cc[sc][ry + 1][tx]);
    const double2 zm = *reinterpret_cast<const double2*>(&S.cc[sm][ry][tx]);
    const double2 zp = *reinterpret_cast<const double2*>(&S.cc[sp][ry][tx]);
    double h0 = 0.0, h1 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h0 = h0 + hs(en[k].x);
      h1 = h1 + hs(en[k].y);
    }
    return make_double2(mu_cell(c.x, xl, c.y, ym.x, yp.x, zm.x, zp.x, h0), mu_cell(c.y, c.x, xr, ym.y, yp.y, zm.y, zp.y, h1));
  };

  // prologue: ring planes zb-2, zb-1; halos and own cells of c(zb), eta(zb-1)
  B2_LOAD_C_OWN(pc, zb - 2)
  B2_LOAD_C_HALO(zb - 2)
  B2_STORE_C(pc, zb - 2)
  B2_LOAD_C_OWN(pc, zb - 1)
  B2_LOAD_C_HALO(zb - 1)
  B2_STORE_C(pc, zb - 1)
  B2_LOAD_C_HALO(zb)
  B2_LOAD_E_HALO(peh, zb - 1)
  B2_LOAD_C_OWN(pc, zb)
  B2_LOAD_E_OWN(pe, zb - 1)
#pragma unroll
  for (int k = 0; k < 4; ++k) eh[k] = make_double2(0.0, 0.0);
  for (int z = zb - 2; z < ze; ++z) {
    // ---- P1 ----
    B2_STORE_C(pc, z + 2)
    __syncthreads();
    // ---- P2: mu(z+1) ----
    // eta(z+1) was requested one whole plane step ago (P2 of the previous step) and is taken over only now: consuming it in P4
    // of the step that requested it gave the loads one P3 to land and left the memory pipe idle for half of every plane
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e[k][2] = pe[k];
      eh[k] = peh[k];
    }
    {
      const int sm = (z + 3) % 3, sc = (z + 4) % 3, sp = (z + 5) % 3, ms = (z + 3) & 1;   // planes z, z+1, z+2; mu slot of z+1
      double2 en[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) en[k] = e[k][2];
      const double2 m = mu_pair(sm, sc, sp, oy, en);
      *reinterpret_cast<double2*>(&S.mu[ms][wave + 1][tx]) = m;
      if (e_row) {  // ring rows: eta halo values of this wave give h there
        const int ry = ehr + 1;  // ring row 1 (y0-1) or TY+2 (y0+TY)
        *reinterpret_cast<double2*>(&S.mu[ms][ehr][tx]) = mu_pair(sm, sc, sp, ry, eh);
      }
      if (e_col) {  // ring columns: one cell per lane
        const int ry = ecr + 1, cx = ecx;
        double h = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) h = h + hs(eh[k].x);
        S.mu[ms][ecr][cx] = mu_cell(S.cc[sc][ry][cx], S.cc[sc][ry][cx - 1], S.cc[sc][ry][cx + 1], S.cc[sc][ry - 1][cx],
                                    S.cc[sc][ry + 1][cx], S.cc[sm][ry][cx], S.cc[sp][ry][cx], h);
      }
    }
    B2_LOAD_C_HALO(z + 3)          // halos first, then the own cells (see above)
    B2_LOAD_E_HALO(peh, z + 2)
    B2_LOAD_C_OWN(pc, z + 3)
    B2_LOAD_E_OWN(pe, z + 2)
    __syncthreads();
    // ---- P3: outputs of plane z ----
    const double2 mu_z = *reinterpret_cast<const double2*>(&S.mu[(z + 2) & 1][wave + 1][tx]);   // own mu of plane z (garbage in the
                                                                                              // first warm-up step: never used)
    if (z >= zb) {
      const int sc = (z + 3) % 3, ms = (z + 2) & 1, ty = wave + 1;
      const double2 c = *reinterpret_cast<const double2*>(&S.cc[sc][oy][tx]);
      const int64_t o = (int64_t)z * plane + a_own;
      {
        const double xl = S.mu[ms][ty][tx - 1], xr = S.mu[ms][ty][tx + 2];
        const double2 ym = *reinterpret_cast<const double2*>(&S.mu[ms][ty - 1][tx]);
        const double2 yp = *reinterpret_cast<const double2*>(&S.mu[ms][ty + 1][tx]);
        const double2 m = mu_z;
        const double2 mp = *reinterpret_cast<const double2*>(&S.mu[ms ^ 1][ty][tx]);   // own mu of plane z + 1
        double lx = ((xl + m.y) + (ym.x + yp.x)) - 4.0 * m.x, ly = ((m.x + xr) + (ym.y + yp.y)) - 4.0 * m.y;
        lx = lx + ((mu_m1.x + mp.x) - 2.0 * m.x);
        ly = ly + ((mu_m1.y + mp.y) - 2.0 * m.y);
        st_out<NT>(un + o, make_double2(c.x + (dt * Mob * p.inv_h2) * lx, c.y + (dt * Mob * p.inv_h2) * ly));
      }
      double2 res[4];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const double cv = h2 ? c.y : c.x;
        double ev[4], e2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ev[k] = h2 ? e[k][1].y : e[k][1].x;
          e2 = e2 + ev[k] * ev[k];
        }
        const double dfab = r2 * ((cv - cb) * (cv - cb)) - r2 * ((cv - ca) * (cv - ca));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double ek = ev[k];
          const double xl = h2 ? e[k][1].x : S.et[k][ty][tx - 1], xr = h2 ? S.et[k][ty][tx + 2] : e[k][1].y;
          const double ym = S.et[k][ty - 1][tx + h2], yp = S.et[k][ty + 1][tx + h2];
          double l = ((xl + xr) + (ym + yp)) - 4.0 * ek;
          l = l + (((h2 ? e[k][0].y : e[k][0].x) + (h2 ? e[k][2].y : e[k][2].x)) - 2.0 * ek);
          const double well = (2.0 * ek * ((1.0 - ek) * (1.0 - ek)) - 2.0 * (ek * ek) * (1.0 - ek)) + (2.0 * al) * ek * (e2 - ek * ek);
          const double fe = dfab * hsp(ek) + w * well;
          const double ne = ek - (dt * L) * (fe - (ke * p.inv_h2) * l);
          if (h2)
            res[k].y = ne;
          else
            res[k].x = ne;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) st_out<NT>(un + (int64_t)(k + 1) * p.fs + o, res[k]);
    }
    __syncthreads();
    // ---- P4: eta(z+1) -> tiles; rotate ----
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      *reinterpret_cast<double2*>(&S.et[k][wave + 1][tx]) = e[k][2];
      if (e_row) *reinterpret_cast<double2*>(&S.et[k][ehr][tx]) = eh[k];
      if (e_col) S.et[k][ecr][ecx] = eh[k].x;
      e[k][0] = e[k][1];
      e[k][1] = e[k][2];
    }
    mu_m1 = mu_z;
  }
#undef B2_LOAD_C_OWN
#undef B2_LOAD_C_HALO
#undef B2_STORE_C
#undef B2_LOAD_E_OWN
#undef B2_LOAD_E_HALO
}

// diagnostics, raw sums per block: {sum second (BM2: c, BM3: (phi+1)/2), sum f_chem, sum of weighted squared forward
// differences (sum_f gradc_f |fwd diff u_f|^2), min over all fields, max over all fields}
__device__ __forceinline__ double fwd2(const double* __restrict__ u, int x, int y, int z, int nx, int ny, int nz) {
  const int64_t row = (int64_t)nx, plane = (int64_t)nx * ny;
  const int64_t zc = z * plane, yc = y * row;
  const double c = u[zc + yc + x];
  const double dx = u[zc + yc + wrapm(x + 1, nx)] - c, dy = u[zc + wrapm(y + 1, ny) * row + x] - c;
  double g = dx * dx + dy * dy;
  if (nz > 1) {
    const double dz = u[wrapm(z + 1, nz) * plane + yc + x] - c;
    g = g + dz * dz;
  }
  return g;
}

__global__ __launch_bounds__(256) void mfd_diag_kernel(const MfdParams p, const double* __restrict__ u,
                                                       double* __restrict__ partials) {
  __shared__ double sh[5][4];
  const int64_t cells = (int64_t)p.nx * p.ny * p.nz;
  const int64_t own0 = (int64_t)p.nx * p.ny * p.gz, own1 = cells - own0;   // owned cells (all of them unless slab mode)
  double v[5] = {0.0, 0.0, 0.0, INFINITY, -INFINITY};
  for (int64_t i = own0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < own1; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % p.nx), y = (int)((i / p.nx) % p.ny), z = (int)(i / ((int64_t)p.nx * p.ny));
    if (p.model == 2) {
      const double ca = p.q[0], cb = p.q[1], r2 = p.q[2], kc = p.q[3], ke = p.q[5], w = p.q[6], al = p.q[7];
      const double c = u[i];
      double e[4], h = 0.0, g = 0.0;
      for (int k = 0; k < 4; ++k) e[k] = u[(k + 1) * p.fs + i];
      for (int k = 0; k < 4; ++k) {
        h = h + hs(e[k]);
        g = g + (e[k] * e[k]) * ((1.0 - e[k]) * (1.0 - e[k]));
        for (int j = k + 1; j < 4; ++j) g = g + al * ((e[k] * e[k]) * (e[j] * e[j]));
      }
      const double fa = r2 * ((c - ca) * (c - ca)), fb = r2 * ((c - cb) * (c - cb));
      v[0] += c;
      v[1] += (fa * (1.0 - h) + fb * h) + w * g;
      double gr = kc * fwd2(u, x, y, z, p.nx, p.ny, p.nz);
      for (int k = 0; k < 4; ++k) gr = gr + ke * fwd2(u + (k + 1) * p.fs, x, y, z, p.nx, p.ny, p.nz);
      v[2] += gr;
      v[3] = fmin(v[3], c);
      v[4] = fmax(v[4], c);
      for (int k = 0; k < 4; ++k) {
        v[3] = fmin(v[3], e[k]);
        v[4] = fmax(v[4], e[k]);
      }
    } else {
      const double lam = p.q[0], W2 = p.q[2];
      const double U = u[i], ph = u[p.fs + i], p2 = ph * ph;
      v[0] += 0.5 * (ph + 1.0);
      v[1] += (-0.5 * p2 + 0.25 * (p2 * p2)) + (lam * U) * ph * ((1.0 - (2.0 / 3.0) * p2) + 0.2 * (p2 * p2));
      v[2] += W2 * fwd2(u + p.fs, x, y, z, p.nx, p.ny, p.nz);
      v[3] = fmin(v[3], fmin(U, ph));
      v[4] = fmax(v[4], fmax(U, ph));
    }
  }
  for (int k = 0; k < 5; ++k) {
    double a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double b = __shfl_down(a, o, 64);
      a = k < 3 ? a + b : (k == 3 ? fmin(a, b) : fmax(a, b));
    }
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    double a;
    if (k < 3)
      a = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
    else if (k == 3)
      a = fmin(fmin(sh[k][0], sh[k][1]), fmin(sh[k][2], sh[k][3]));
    else
      a = fmax(fmax(sh[k][0], sh[k][1]), fmax(sh[k][2], sh[k][3]));
    partials[(int64_t)blockIdx.x * 5 + k] = a;
  }
}

__global__ void mfd_diag_final_kernel(const double* __restrict__ partials, int nb, double* __restrict__ out5) {
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    double a = k < 3 ? 0.0 : (k == 3 ? INFINITY : -INFINITY);
    for (int b = 0; b < nb; ++b) {
      const double v = partials[(int64_t)b * 5 + k];
      a = k < 3 ? a + v : (k == 3 ? fmin(a, v) : fmax(a, v));
    }
    out5[k] = a;
  }
}

// initial conditions on the lattice (z-extruded); mnx / mny > 0: even extension of a no-flux domain with that many nodes
__global__ __launch_bounds__(256) void mfd_ic_kernel(const MfdParams p, double* __restrict__ u, double h, int mnx, int mny,
                                                     double a0, double a1, double a2, double a3, double a4) {
  const int64_t plane = (int64_t)p.nx * p.ny;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane) return;
  const int x = (int)(i % p.nx), y = (int)(i / p.nx);
  const int xe = (mnx > 0 && x >= mnx) ? 2 * (mnx - 1) - x : x;
  const int ye = (mny > 0 && y >= mny) ? 2 * (mny - 1) - y : y;
  const double X = xe * h, Y = ye * h;
  double val[5];
  int nf;
  if (p.model == 2) {  // pfbase.py:268-296; a0..a3 = c0, eps, eps_eta, psi
    nf = 5;
    const double t2 = cos(0.13 * X) * cos(0.087 * Y);
    val[0] = a0 + a1 * (cos(0.105 * X) * cos(0.11 * Y) + t2 * t2 + cos(0.025 * X - 0.15 * Y) * cos(0.07 * X - 0.02 * Y));
    for (int k = 0; k < 4; ++k) {
      const double ii = k + 1.0, i0 = (double)k;
      const double a = cos((0.01 * ii) * X - 4.0) * cos((0.007 + 0.01 * ii) * Y);
      const double b = cos((0.11 + 0.01 * ii) * X) * cos((0.11 + 0.01 * ii) * Y);
      const double cc = cos((0.046 + 0.001 * i0) * X - (0.0405 + 0.001 * i0) * Y) * cos((0.031 + 0.001 * i0) * X - (0.004 + 0.001 * i0) * Y);
      const double sum = (a + b) + a3 * (cc * cc);
      val[1 + k] = a2 * (sum * sum);
    }
  } else {  // pfbase.py:298-320; a0..a4 = Delta, r, w, vin, vout
    nf = 2;
    const double r = sqrt(X * X + Y * Y);
    val[0] = a0;
    if (r < a1 - 0.5 * a2)
      val[1] = a3;
    else if (r > a1 + 0.5 * a2)
      val[1] = a4;
    else
      val[1] = a4 + 0.5 * (a3 - a4) * (1.0 + cos(3.14159265358979323846 * (r - a1 + 0.5 * a2) / a2));
  }
  for (int f = 0; f < nf; ++f)
    for (int z = 0; z < p.nz; ++z) u[f * p.fs + z * plane + i] = val[f];
}

}  // namespace

struct MultiFD {
  MfdParams p;
  int64_t cells = 0;
  unsigned char* block = nullptr;   // the library's own two time levels (null: the caller's buffers)
  int ncu = 256;
  double h = 1.0;
  double* u[2] = {nullptr, nullptr};
  double* mu = nullptr;
  double *partials = nullptr, *out5 = nullptr, *out5_host = nullptr;
  int cur = 0;
  bool have_prev = false;
  bool own_u = true;
  hipStream_t stream = nullptr;
  bool use_stream = true;  // PFHIP_MFD_STREAM=0 (read at create): one-thread-per-cell kernels everywhere (A/B)
  bool bm2_fused = true;   // PFHIP_BM2_FUSED=0: BM2 by the two streaming passes (mu through HBM) instead of the one-pass kernel
  std::string err;
};

#define MF_HIP(expr)                                                     \
  do {                                                                   \
    hipError_t e_ = (expr);                                              \
    if (e_ != hipSuccess) {                                              \
      mf->err = std::string(#expr) + ": " + hipGetErrorString(e_);       \
      return -3;                                                         \
    }                                                                    \
  } while (0)

const char* multifd_error(const MultiFD* mf) { return mf->err.c_str(); }
int multifd_nfields(const MultiFD* mf) { return mf->p.nf; }

// model 2: mp = {c_alpha, c_beta, rho, kappa_c, M, kappa_eta, w, alpha, L}; model 3: mp = {W0, tau0, D, Delta}
int multifd_create(MultiFD** out, int model, int nx, int ny, int nz, int gz, double h, const double* mp, double* ext0,
                   double* ext1, hipStream_t stream, std::string* err) {
  MultiFD* mf = new MultiFD();
  *out = mf;
  MfdParams& p = mf->p;
  p.model = model;
  p.nf = model == 2 ? 5 : 2;
  p.nx = nx;
  p.ny = ny;
  p.nz = nz;
  p.gz = gz;
  p.inv_h2 = 1.0 / (h * h);
  if (const char* e = getenv("PFHIP_MFD_STREAM")) mf->use_stream = e[0] != '0';
  if (const char* e = getenv("PFHIP_BM2_FUSED")) mf->bm2_fused = e[0] != '0';
  for (double& q : p.q) q = 0.0;
  if (model == 2) {
    for (int i = 0; i < 9; ++i) p.q[i] = mp[i];
    p.q[2] = mp[2] * mp[2];  // rho^2
  } else {
    p.q[0] = mp[2] * mp[1] / (0.6267 * mp[0] * mp[0]);  // lam = D tau0 / (0.6267 W0^2)
    p.q[1] = 1.0 / mp[1];
    p.q[2] = mp[0] * mp[0];
    p.q[3] = mp[2];
  }
  mf->h = h;
  {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) mf->ncu = n;
  }
  mf->cells = (int64_t)nx * ny * nz;
  mf->stream = stream;
  auto body = [&]() -> int {
    size_t bytes = sizeof(double) * (size_t)mf->cells * p.nf;
    p.fs = mf->cells;
    if (ext0 && ext1) {
      mf->u[0] = ext0;
      mf->u[1] = ext1;
      mf->own_u = false;
    } else {
      // Both time levels from ONE block (csrc/device_alloc.hip: physically contiguous, so distances inside it are physical
      // distances).  At 512^3 a field is exactly 1 GiB, and the 2 nf streams of a step (every field read and written) would
      // all share their low 30 address bits: fields are spaced 68 KiB further apart and the second level starts 64 KiB
      // after a 512 KiB boundary (the distance that serves the BM1 kernel, pfhip_api.hip placed_offset_bytes).  Worth
      // 1.5 % for BM3 (0.787 -> 0.775 ms) and 0.5 % for BM2 on contiguous memory, 13 combinations scanned
      // (profiles/r04/mfd_placement_scan.log) -- on this memory system the big effect is WHERE a block lies, not how its
      // parts are spaced.  The caller's buffers (slab mode) keep their dense layout: fs = cells.
      const int64_t fpad = 68 * 1024, off = 64 * 1024;
      p.fs = mf->cells + fpad / (int64_t)sizeof(double);
      bytes = sizeof(double) * (size_t)p.fs * p.nf;
      const size_t lvl = (bytes + (512u << 10) - 1) / (512u << 10) * (512u << 10) + (size_t)off;
      MF_HIP(pf_malloc(&mf->block, lvl + bytes));
      mf->u[0] = reinterpret_cast<double*>(mf->block);
      mf->u[1] = reinterpret_cast<double*>(mf->block + lvl);
    }
    MF_HIP(hipMemsetAsync(mf->u[0], 0, bytes, stream));
    MF_HIP(hipMemsetAsync(mf->u[1], 0, bytes, stream));
    if (model == 2) MF_HIP(pf_malloc(&mf->mu, sizeof(double) * (size_t)mf->cells));
    if (model == 2) {
      MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bm2_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(Bm2Lds)));
      MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bm2_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(Bm2Lds)));
    }
    MF_HIP(pf_malloc(&mf->partials, sizeof(double) * 5 * 1024));
    MF_HIP(pf_malloc(&mf->out5, sizeof(double) * 8));
    MF_HIP(hipHostMalloc(&mf->out5_host, sizeof(double) * 8, hipHostMallocDefault));
    return 0;
  };
  int rc = body();
  if (rc && err) *err = mf->err;
  return rc;
}

void multifd_destroy(MultiFD* mf) {
  if (!mf) return;
  if (!mf->own_u) mf->u[0] = mf->u[1] = nullptr;
  for (void* q : {(void*)mf->block, (void*)mf->mu, (void*)mf->partials, (void*)mf->out5})
    if (q) (void)pf_free(q);
  if (mf->out5_host) (void)hipHostFree(mf->out5_host);
  delete mf;
}

// a: BM2 {c0, eps, eps_eta, psi, -}; BM3 {Delta, r, w, vin, vout}
int multifd_set_ic(MultiFD* mf, int mnx, int mny, const double* a) {
  const MfdParams& p = mf->p;
  const int64_t plane = (int64_t)p.nx * p.ny;
  hipLaunchKernelGGL(mfd_ic_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, mf->stream, p, mf->u[mf->cur],
                     mf->h, mnx, mny, a[0], a[1], a[2], a[3], a[4]);
  MF_HIP(hipGetLastError());
  mf->have_prev = false;
  return 0;
}

double* multifd_field_base(MultiFD* mf, int f) { return mf->u[mf->cur] + (int64_t)f * mf->p.fs; }
int multifd_cur_index(const MultiFD* mf) { return mf->cur; }
double* multifd_field_ptr(MultiFD* mf, int f) {   // the owned planes
  return mf->u[mf->cur] + (int64_t)f * mf->p.fs + (int64_t)mf->p.gz * mf->p.nx * mf->p.ny;
}
void multifd_touch(MultiFD* mf) { mf->have_prev = false; }

int g_mfd_nt = 1;  // non-temporal stores of the output planes of the streaming multi-field kernels (pfk_set_tuning key 10 = 0: plain stores, A/B:
                   // BM3 0.8416 -> 0.8145 ms per 512^3 step, BM2 2.275 -> 2.265 in one process, profiles/r04/mfd_nt_stores_ab.log)
void multifd_set_nt(int v) { g_mfd_nt = v; }
// which kernels multifd_step uses on this box: 1 = the streaming LDS-tiled forms, 0 = one thread per cell
int multifd_streaming(const MultiFD* mf) {
  const MfdParams& p = mf->p;
  return mf->use_stream && p.nz >= 4 && p.nx % SX == 0 && p.ny % 16 == 0;   // tile heights: 16 (BM3), 8 / 4 (BM2 passes), 8 (one-pass BM2)
}
namespace {
// z-chunks so that the grid holds ONE workgroup per CU; planes [zlo, zhi) of the box.  Measured at 512^3 (BM3,
// profiles/r04/mfd_bm3_grid_scan.log): 256 workgroups 0.743 ms, 512: 0.755, 768: 0.760, 1024 (rounds 3-4a): 0.775, 3072: 0.753,
// 128: 1.23 -- four waves per CU already keep enough bytes in flight (the next plane is requested a plane ahead), and every
// further workgroup is another set of streams for the DRAM pages and another z-chunk boundary whose halo planes are read twice.
// With one workgroup per CU the BM3 kernel keeps TWO planes in flight (PassTraits<30>::AHEAD): 0.742 -> 0.694 ms (frac 0.77).
template <int PASS>
void launch_stream(const MultiFD* mf, const double* u, const double* mu, double* out, double dt, int zlo, int zhi) {
  MfdParams p = mf->p;
  p.zlo = zlo;
  p.zhi = zhi;
  const int nzr = zhi - zlo;
  constexpr int TY = 4 * PassTraits<PASS>::RPT;
  const int tiles = (p.nx / SX) * (p.ny / TY);
  int nchunk = (mf->ncu + tiles - 1) / tiles;
  if (nchunk > nzr / 8) nchunk = nzr / 8 > 0 ? nzr / 8 : 1;
  const int zchunk = (nzr + nchunk - 1) / nchunk;
  nchunk = (nzr + zchunk - 1) / zchunk;
  if (g_mfd_nt)
    hipLaunchKernelGGL((mfd_stream_kernel<PASS, true>), dim3(tiles * nchunk), dim3(256), 0, mf->stream, p, u, mu, out, dt, zchunk);
  else
    hipLaunchKernelGGL((mfd_stream_kernel<PASS, false>), dim3(tiles * nchunk), dim3(256), 0, mf->stream, p, u, mu, out, dt, zchunk);
}

// one explicit step on planes [zlo, zhi) of the box: current level -> other level (no swap).  The two-pass BM2 forms
// need mu one plane beyond the range on each side (clamped to the box: a whole-box call wraps inside the kernels).
void launch_range(MultiFD* mf, double dt, int zlo, int zhi) {
  if (zhi <= zlo) return;
  MfdParams p = mf->p;
  p.zlo = zlo;
  p.zhi = zhi;
  const double* u = mf->u[mf->cur];
  double* un = mf->u[1 - mf->cur];
  const int64_t plane = (int64_t)p.nx * p.ny;
  const int mlo = zlo > 0 ? zlo - 1 : 0, mhi = zhi < p.nz ? zhi + 1 : p.nz;   // mu range of the two-pass BM2 forms
  if (multifd_streaming(mf) != 0) {
    if (p.model == 2 && mf->bm2_fused) {
      const int nzr = zhi - zlo;
      const int tiles = (p.nx / SX) * (p.ny / B2TY);
      int nchunk = (256 + tiles - 1) / tiles;   // one workgroup per CU
      if (nchunk > nzr / 8) nchunk = nzr / 8 > 0 ? nzr / 8 : 1;
      const int zchunk = (nzr + nchunk - 1) / nchunk;
      nchunk = (nzr + zchunk - 1) / zchunk;
      if (g_mfd_nt)
        hipLaunchKernelGGL(bm2_fused_kernel<true>, dim3(tiles * nchunk), dim3(B2T), sizeof(Bm2Lds), mf->stream, p, u, un, dt, zchunk);
      else
        hipLaunchKernelGGL(bm2_fused_kernel<false>, dim3(tiles * nchunk), dim3(B2T), sizeof(Bm2Lds), mf->stream, p, u, un, dt, zchunk);
    } else if (p.model == 2) {
      launch_stream<20>(mf, u, nullptr, mf->mu, dt, mlo, mhi);
      launch_stream<21>(mf, u, mf->mu, un, dt, zlo, zhi);
    } else {
      launch_stream<30>(mf, u, nullptr, un, dt, zlo, zhi);
    }
    return;
  }
  const unsigned nb = (unsigned)(((int64_t)(zhi - zlo) * plane + 255) / 256);
  if (p.model == 2) {
    MfdParams pm = p;
    pm.zlo = mlo;
    pm.zhi = mhi;
    hipLaunchKernelGGL(bm2_mu_kernel, dim3((unsigned)(((int64_t)(mhi - mlo) * plane + 255) / 256)), dim3(256), 0, mf->stream, pm, u,
                       mf->mu);
    hipLaunchKernelGGL(bm2_update_kernel, dim3(nb), dim3(256), 0, mf->stream, p, u, (const double*)mf->mu, un, dt);
  } else {
    hipLaunchKernelGGL(bm3_update_kernel, dim3(nb), dim3(256), 0, mf->stream, p, u, un, dt);
  }
}
}  // namespace

int multifd_step(MultiFD* mf, double dt, int nsteps) {
  for (int s = 0; s < nsteps; ++s) {
    launch_range(mf, dt, 0, mf->p.nz);
    mf->cur ^= 1;
    mf->have_prev = true;
  }
  MF_HIP(hipGetLastError());
  return 0;
}

// slab mode, one step in two parts (pf_step_begin / pf_step_finish of BM2 / BM3 handles): planes [zlo, zhi) of the ghosted
// local box, current level -> other level, no swap; then multifd_swap
int multifd_step_range(MultiFD* mf, double dt, int zlo, int zhi) {
  if (zlo < 0 || zhi > mf->p.nz) return -1;
  launch_range(mf, dt, zlo, zhi);
  MF_HIP(hipGetLastError());
  return 0;
}
void multifd_swap(MultiFD* mf) {
  mf->cur ^= 1;
  mf->have_prev = true;
}

int multifd_rollback(MultiFD* mf) {
  if (!mf->have_prev) return -4;
  mf->cur ^= 1;
  mf->have_prev = false;
  return 0;
}

// raw[5] = {sum second, sum f_chem, weighted sum of squared forward differences, min, max}; synchronises
int multifd_diag_raw(MultiFD* mf, double raw[5]) {
  const MfdParams& p = mf->p;
  int nb = (int)((mf->cells + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(mfd_diag_kernel, dim3(nb), dim3(256), 0, mf->stream, p, (const double*)mf->u[mf->cur], mf->partials);
  hipLaunchKernelGGL(mfd_diag_final_kernel, dim3(1), dim3(64), 0, mf->stream, (const double*)mf->partials, nb, mf->out5);
  MF_HIP(hipMemcpyAsync(mf->out5_host, mf->out5, 5 * sizeof(double), hipMemcpyDeviceToHost, mf->stream));
  MF_HIP(hipStreamSynchronize(mf->stream));
  for (int k = 0; k < 5; ++k) raw[k] = mf->out5_host[k];
  return 0;
}

}  // namespace pfhip
