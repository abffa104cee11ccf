// Internal declarations shared by the libpfhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "pfhip.h"

namespace pfhip {

// device_alloc.hip: every device allocation of the library.  Arrays of 32 MiB and more are physically contiguous
// (PFHIP_ALLOC), smaller ones are plain hipMallocs; pf_free takes either kind.
hipError_t pf_malloc_bytes(void** p, size_t bytes);
hipError_t pf_free(void* p);
template <class T>
inline hipError_t pf_malloc(T** p, size_t bytes) {
  return pf_malloc_bytes(reinterpret_cast<void**>(p), bytes);
}
std::string pf_alloc_describe();
void pf_alloc_set_fill(int byte);   // test hook: fill new allocations with a byte pattern (-1 = off)
// stream_util.hip: streams that do not share a hardware queue with the streams they are meant to run beside
bool pf_streams_overlap(hipStream_t a, hipStream_t b);
hipError_t pf_acquire_stream(const hipStream_t* avoid, int navoid, hipStream_t* out, bool* found);
int pf_acquire_side_streams(hipStream_t main, int want, hipStream_t* out);


// ---- arguments of the FD Cahn-Hilliard step launchers ---------------------------------------------------
struct FdArgs {
  const double* cin;
  double* cout;
  const double* phi;  // nullptr for BM1
  int nx, ny, nz;     // owned extents (nz = planes of the slab axis)
  int ghost;          // ghost planes per side present in the buffers
  int zwrap;          // 1: periodic inside the buffer; 0: read ghost planes
  int zlo, zhi;       // output plane range [zlo, zhi)
  int zlo2, zhi2;     // optional second output range of the same launch (fused kernel only; empty if zhi2 <= zlo2)
  double ca, cb, two_rho, kh2, amh2, kphi;
  double gq, cbar;    // BM6 with phi eliminated: cnew += gq (c - cbar); gq == 0: off
  // second range as SEVERAL chunks of (zhi2 - zlo2) planes, zstride2 apart (0: one chunk) -- the two boundary strips of
  // a slab -- dispatched after all chunks of the first range; and, for the single-launch slab step, the arrival flags
  // of the ghost planes: a second-range workgroup polls wait_lo / wait_hi (whichever its planes touch; null = no
  // neighbour on that side) until it holds wait_seq, bounded, before it reads anything
  int zstride2 = 0, nchunk2 = 1;
  const long long* wait_lo = nullptr;
  const long long* wait_hi = nullptr;
  long long wait_seq = 0;
  int* wait_timeout = nullptr;
};

// launchers (return hipError_t of the launch); all asynchronous on `stream`
hipError_t launch_ch_fd_fused(const FdArgs& a, hipStream_t stream);
// mu_scratch: (zhi - zlo + 2) planes of nx*ny doubles
hipError_t launch_ch_fd_twopass(const FdArgs& a, double* mu_scratch, hipStream_t stream);
bool ch_fd_fused_supported(const FdArgs& a);
// 2-D (nz == 1, single rank, BM1): K in {1, 2, 4} time steps per launch, cin -> cout
bool ch_fd2d_supported(const FdArgs& a);
hipError_t launch_ch_fd2d(const FdArgs& a, int K, hipStream_t stream);
void set_2d_rows(int k, int rows);

// diagnostics: raw sums {sum c, sum f_chem, sum |fwd diff|^2, sum c*phi, min c, max c} -> out6 (device, 6 doubles)
// partials: device scratch of diag_partials_elems() doubles
int diag_partials_elems();
void set_diag_tuning(int variant, int target_blocks);  // < 0 / <= 0: keep
// zends: bit 0 / bit 1 = local plane 0 / nz-1 is a no-flux wall of a z-line decomposition (trapezoid weights)
hipError_t launch_diag(const double* c, const double* phi, int nx, int ny, int nz, int ghost, int zwrap, int zends,
                       double rho, double ca, double cb, double* partials, double* out6, hipStream_t stream);
// fill the ghost planes outside a no-flux wall with mirror images of the owned planes (ends: same bits as zends)
void set_copy_tuning(int wgs_per_cu, int mode);  // <= 0 / < 0: keep
hipError_t launch_push_planes(const double* src, double* dst, int64_t n, long long* flag, long long seq,
                              unsigned* ticket, hipStream_t stream);
void set_push_wgs(int n);
hipError_t launch_signal_flag(long long* flag, long long seq, hipStream_t stream);
hipError_t launch_wait_flag(const long long* flag, long long seq, int* timeout, hipStream_t stream);
hipError_t run_grid_barrier_probe(int nblocks, int nthreads, int iters, double* ms_per_barrier, int* ok);
hipError_t launch_clock_probe(double* out, int nblocks, int spin_us, int busy, hipStream_t stream);
hipError_t run_xcd_barrier_probe(int nblocks, int nthreads, int iters, int handoff, double* us_per_round, int* participants,
                                 int* stale, int* ok);
hipError_t launch_stream_copy(const double* src, double* dst, int64_t n, hipStream_t stream);
hipError_t launch_reflect_ghosts(double* buf, int64_t plane, int nz, int ghost, int ends, hipStream_t stream);

// initial condition (pfbase.py:187-189 / :332-334), z-extruded; writes owned planes [0, nz) of a ghosted buffer
// mnx, mny > 0: lattice is the even extension of a no-flux domain with mnx x mny nodes (index reflection)
hipError_t launch_ic(double* c, int nx, int ny, int nz, int ghost, double h, double c0, double amp, double w0,
                     int mnx, int mny, hipStream_t stream);
void set_fused_variant(int v);
void set_fused_chunking(int target_wgs, int min_chunk);

// semi-implicit Fourier-spectral scheme (spectral.hip); functions return 0 or a negative pf_status, message in
// spectral_error()
struct Spectral;
int spectral_create(Spectral** out, int dim, int nx, int ny, int nz, double h, hipStream_t stream, std::string* err);
void spectral_destroy(Spectral* sp);
void spectral_invalidate(Spectral* sp);
void spectral_set_screening(Spectral* sp, double gq);  // BM6: gq = k^2 / eps; the step then treats -M gq (c - mean) implicitly  // call whenever the real-space field changed behind the scheme's back
int spectral_step(Spectral* sp, const double* c_in, double* c_out, double dt, double M, double kappa, double ca,
                  double cb, double two_rho, hipStream_t stream, bool store_field = true);
int spectral_grad_energy(Spectral* sp, const double* c, double* out_dev, hipStream_t stream);
const char* spectral_error(const Spectral* sp);
const char* spectral_path(const Spectral* sp);  // which transform kernels this handle runs

// rocFFT through its native API (fftplan.hip): the library fallback for sizes the hand-written passes do not cover
struct FftPlan;
int fftplan_real(FftPlan** out, int dim, const int* n, int batch, bool forward, hipStream_t stream, std::string* err);
int fftplan_c2c_strided(FftPlan** out, int n, int64_t stride, int64_t dist, int batch, hipStream_t stream, bool forward,
                        std::string* err);
int fftplan_exec(FftPlan* p, void* in, void* out_buf, std::string* err);   // out_buf = nullptr: in place
void fftplan_destroy(FftPlan* p);

// fused LDS-FFT spectral step for 2-D power-of-two grids (spectral2d_fused.hip); same spectrum layout as rocFFT D2Z
struct Fused2D;
bool fused2d_supported(int dim, int nx, int ny, int nz);
struct SpecLayout {  // a half-spectrum array of the hand-written passes: [z][nyp][pitch] complex elements (nyp = ny + pad rows)
  int pitch, nyp;
  int64_t rows;      // rows to allocate (pitch complex elements each)
};
SpecLayout fused_spectrum_layout(int dim, int nx, int ny, int nz);
int fused_spectrum_pitch(int dim, int nx, int ny, int nz);  // complex elements per k_x row of the arrays handed to fused2d_* / fused3d_poisson  // 2-D power-of-two grids, or the 512^3 cube
int fused2d_create(Fused2D** out, int nx, int ny, int nz, double h, hipStream_t stream, int want_mx = 0);  // want_mx: also the
                                                     // tables of the mixed-radix kernels (fused_poisson_dirichlet)
bool fused_dirichlet_supported(int dim, int nx, int ny, int nz);
int64_t fused_dirichlet_work_doubles(int npx, int npy, int npz);
int fused_poisson_dirichlet(Fused2D* f, const double* c, double* phi, double* S, int npx, int npy, int npz, double h,
                            double k_over_eps);
void fused2d_destroy(Fused2D* f);
void fused2d_invalidate(Fused2D* f);
// slab-decomposed transforms: local x / y passes with the all-to-all layout written / read directly, z pass on the
// transposed layout (spectral2d_fused.hip; used by slabfft.hip when every axis is a power of two in 128..1024)
bool fusedslab_supported(int nx, int ny, int nz, int P);
int fusedslab_forward_xy(Fused2D* f, const double* real_in, double2* tmp, double2* A, int nzl, int P, int use_fprime,
                         double ca, double cb, double two_rho);
int fusedslab_inverse_yx(Fused2D* f, double2* A, double2* tmp, double* real_out, int nzl, int P);
int fusedslab_z(Fused2D* f, double2* B, double2* chat, int mode, int nyl, int yoff, double dtM, double dtMkappa);
int fused3d_poisson(Fused2D* f, const double* c, double* phi, double2* W, double k_over_eps, double inv_h2);
int fused2d_spectrum(Fused2D* f, const double* c, double2* chat, double2* G);
const char* fused2d_describe(const Fused2D* f);  // one line: which kernels serve this box
int fused2d_persistent_steps(Fused2D* f, double2* chat, double2* G, double2* H, int nsteps, double dt, double M, double kappa,
                             double ca, double cb, double two_rho, double gam);  // 0 done, 1 not available, -3 error
int fused2d_step(Fused2D* f, const double* c_in, double* c_out, double2* chat, double2* G, double2* H, double dt,
                 double M, double kappa, double ca, double cb, double two_rho, double gam);

// Poisson solve of BM6 (poisson.hip): lap(phi) = -k c / eps on the lattice; npx, npy > 0 = the reference's
// Dirichlet-x / no-flux-y boundary conditions on the even extension of an npx x npy-node domain, 0 = periodic box
struct Poisson;
int poisson_create(Poisson** out, int dim, int nx, int ny, int nz, int npx, int npy, double h, double k, double eps,
                   hipStream_t stream, std::string* err);
void poisson_destroy(Poisson* po);
int poisson_solve(Poisson* po, const double* c, double* phi, hipStream_t stream);
const char* poisson_error(const Poisson* po);
const char* poisson_path(const Poisson* po);
int poisson_dirichlet_rhs_planes(const double* c, double* r, int nx, int ny, int nzl, int npx, int npy, double h, double k_over_eps,
                                 hipStream_t stream);   // slab mode: the reference's BCs on a slab of lattice planes
int poisson_dirichlet_fixup_planes(double* phi, int nx, int ny, int nzl, int npx, int npy, double h, hipStream_t stream);  // which transform kernels the solve runs

// slab-decomposed FFT building blocks (slabfft.hip)
struct SlabFFT;
int64_t slabfft_buffer_doubles(int nx, int ny, int nz, int P);
int slabfft_create(SlabFFT** out, int nx, int ny, int nz, int P, int rank, double h, bool with_spectral, double* extA,
                   double* extB, hipStream_t stream, std::string* err);
void slabfft_destroy(SlabFFT* sf);
int64_t slabfft_doubles_per_peer(const SlabFFT* sf);
double* slabfft_buf(const SlabFFT* sf, int which);  // 0 = A (pack side), 1 = B (transposed side)
int slabfft_forward_local(SlabFFT* sf, const double* real_in);
int slabfft_forward_local_dfdc(SlabFFT* sf, const double* c, double ca, double cb, double two_rho);
int slabfft_z(SlabFFT* sf, int inverse);
int slabfft_inverse_local(SlabFFT* sf, double* real_out);
int slabfft_poisson_on_T(SlabFFT* sf, double k_over_eps);
int slabfft_store_chat(SlabFFT* sf);
int slabfft_spectral_update_on_T(SlabFFT* sf, double dtM, double dtMkappa);
int slabfft_z_update(SlabFFT* sf, double dtM, double dtMkappa);
int slabfft_grad_energy_local(SlabFFT* sf, double* out_dev);
const char* slabfft_error(const SlabFFT* sf);
const char* slabfft_path(const SlabFFT* sf);   // which transform kernels run between the all-to-alls

// BE-parity mode (fem_be.hip): the reference's P1 crossed-mesh backward-Euler Newton solve on the GPU
struct FemBE;
int fembe_create(FemBE** out, int nodes_per_side, double h, int nf, double rho, double ca, double cb, double kappa,
                 double Mob, double k, double eps, hipStream_t stream, std::string* err, bool condensed = false);
void fembe_destroy(FemBE* fb);
int fembe_create_model(FemBE** out, int model, int nodes_per_side, double h, const double* mp, hipStream_t stream,
                       std::string* err);  // model 2 = BM2, 3 = BM3 (generic multi-field path)
int fembe_model(const FemBE* fb);           // 0: BM1 / BM6 kernels, 2 / 3: generic
int fembe_nfields(const FemBE* fb);
int fembe_set_ic_gen(FemBE* fb, const double* icp);
int fembe_set_field(FemBE* fb, int f, const double* host);
int fembe_nodes(const FemBE* fb);
int fembe_last_iters(const FemBE* fb);
int fembe_set_ic(FemBE* fb, double c0, double amp, double w0);
int fembe_set_c(FemBE* fb, const double* host);
int fembe_get(FemBE* fb, int field, double* host);
int fembe_step(FemBE* fb, double dt, int* converged, int* iters);
int fembe_rollback(FemBE* fb);
void fembe_set_pivot_always(FemBE* fb);      // PF_FLAG_FEM_ALWAYS_PIVOT
long long fembe_stat(const FemBE* fb, int key);
void fembe_set_max_newton(FemBE* fb, int n);  // n <= 0: keep the default (10, bench1.py:88)
int fembe_diagnostics(FemBE* fb, double out[3]);
const char* fembe_error(const FemBE* fb);
const char* fembe_describe(const FemBE* fb);  // which kernels factor the dense levels (says when the cooperative LU was switched off)

// explicit finite-difference schemes of the multi-field benchmarks BM2 / BM3 (multi_fd.hip)
struct MultiFD;
// gz > 0 (slab mode): nz INCLUDES gz ghost planes per side, refreshed by the caller before every step; ext0 / ext1: optional
// caller-owned buffers for the two time levels (nf x nx x ny x nz doubles each, field-major)
int multifd_create(MultiFD** out, int model, int nx, int ny, int nz, int gz, double h, const double* mp, double* ext0,
                   double* ext1, hipStream_t stream, std::string* err);
void multifd_destroy(MultiFD* mf);
int multifd_nfields(const MultiFD* mf);
int multifd_set_ic(MultiFD* mf, int mnx, int mny, const double* a);
double* multifd_field_base(MultiFD* mf, int f);  // ... including its ghost planes (slab mode)
int multifd_cur_index(const MultiFD* mf);
double* multifd_field_ptr(MultiFD* mf, int f);  // device pointer of field f of the current time level (lattice, no ghosts)
void multifd_touch(MultiFD* mf);                // a field was overwritten: no rollback state
int multifd_step(MultiFD* mf, double dt, int nsteps);
int multifd_streaming(const MultiFD* mf);
int multifd_step_range(MultiFD* mf, double dt, int zlo, int zhi);   // planes [zlo, zhi) of the (ghosted) box, no swap
void multifd_swap(MultiFD* mf);
void multifd_set_nt(int v);   // pfk_set_tuning key 10: non-temporal output stores of the streaming kernels (A/B)  // 1: the box tiles and multifd_step uses the streaming LDS-tiled kernels
int multifd_rollback(MultiFD* mf);
int multifd_diag_raw(MultiFD* mf, double raw[5]);
const char* multifd_error(const MultiFD* mf);

}  // namespace pfhip
