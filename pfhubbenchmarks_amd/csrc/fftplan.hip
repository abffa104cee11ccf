// Library transforms for the sizes the hand-written LDS-FFT passes do not cover (spectral2d_fused.hip: axes that factor
// into 2, 3, 5 up to 1024 points): rocFFT through its NATIVE API (rocfft_plan_create / rocfft_execute) -- no hipFFT
// (cuFFT-shaped) layer in between.  One FftPlan = plan + execution info (stream, work buffer).  Transforms are
// unnormalised in both directions, real transforms use rocFFT's default layouts: real input contiguous x-fastest,
// Hermitian output interleaved with nx/2 + 1 complex per x-row.
#include <rocfft/rocfft.h>

#include <mutex>

#include "pfhip_internal.h"

namespace pfhip {

struct FftPlan {
  rocfft_plan plan = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
};

namespace {

std::once_flag g_setup_once;

const char* rocfft_err(rocfft_status s) {
  switch (s) {
    case rocfft_status_success: return "rocfft_status_success";
    case rocfft_status_failure: return "rocfft_status_failure";
    case rocfft_status_invalid_arg_value: return "rocfft_status_invalid_arg_value";
    case rocfft_status_invalid_dimensions: return "rocfft_status_invalid_dimensions";
    case rocfft_status_invalid_array_type: return "rocfft_status_invalid_array_type";
    case rocfft_status_invalid_strides: return "rocfft_status_invalid_strides";
    case rocfft_status_invalid_distance: return "rocfft_status_invalid_distance";
    case rocfft_status_invalid_offset: return "rocfft_status_invalid_offset";
    case rocfft_status_invalid_work_buffer: return "rocfft_status_invalid_work_buffer";
    default: return "rocfft error";
  }
}

#define RF(expr)                                                    \
  do {                                                              \
    rocfft_status s_ = (expr);                                      \
    if (s_ != rocfft_status_success) {                              \
      if (err) *err = std::string(#expr) + ": " + rocfft_err(s_);   \
      return -3;                                                    \
    }                                                               \
  } while (0)

int finish(FftPlan* p, hipStream_t stream, std::string* err) {
  size_t wb = 0;
  RF(rocfft_plan_get_work_buffer_size(p->plan, &wb));
  RF(rocfft_execution_info_create(&p->info));
  if (wb) {
    if (pf_malloc(&p->work, wb) != hipSuccess) {
      if (err) *err = "hipMalloc of the rocFFT work buffer failed";
      return -5;
    }
    RF(rocfft_execution_info_set_work_buffer(p->info, p->work, wb));
  }
  RF(rocfft_execution_info_set_stream(p->info, stream));
  return 0;
}

}  // namespace

// batch real transforms of an n[0] x n[1] x n[2] box (x fastest; dim = 1..3), out of place.
//   forward: real -> Hermitian half spectrum;  inverse: half spectrum -> real (may overwrite its input)
int fftplan_real(FftPlan** out, int dim, const int* n, int batch, bool forward, hipStream_t stream, std::string* err) {
  std::call_once(g_setup_once, [] { (void)rocfft_setup(); });
  FftPlan* p = new FftPlan();
  *out = p;
  size_t len[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) len[d] = (size_t)n[d];
  RF(rocfft_plan_create(&p->plan, rocfft_placement_notinplace,
                        forward ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                        rocfft_precision_double, (size_t)dim, len, (size_t)batch, nullptr));
  return finish(p, stream, err);
}

// `batch` in-place complex 1-D transforms of n points with element stride `stride`, consecutive transforms `dist` apart
int fftplan_c2c_strided(FftPlan** out, int n, int64_t stride, int64_t dist, int batch, hipStream_t stream,
                        bool forward, std::string* err) {
  std::call_once(g_setup_once, [] { (void)rocfft_setup(); });
  FftPlan* p = new FftPlan();
  *out = p;
  rocfft_plan_description d = nullptr;
  RF(rocfft_plan_description_create(&d));
  const size_t st[1] = {(size_t)stride};
  rocfft_status s = rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved,
                                                            rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, st,
                                                            (size_t)dist, 1, st, (size_t)dist);
  if (s == rocfft_status_success) {
    const size_t len[1] = {(size_t)n};
    s = rocfft_plan_create(&p->plan, rocfft_placement_inplace,
                           forward ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                           rocfft_precision_double, 1, len, (size_t)batch, d);
  }
  (void)rocfft_plan_description_destroy(d);
  RF(s);
  return finish(p, stream, err);
}

int fftplan_exec(FftPlan* p, void* in, void* out_buf, std::string* err) {
  void* ib[1] = {in};
  void* ob[1] = {out_buf};
  RF(rocfft_execute(p->plan, ib, out_buf ? ob : nullptr, p->info));
  return 0;
}

void fftplan_destroy(FftPlan* p) {
  if (!p) return;
  if (p->info) (void)rocfft_execution_info_destroy(p->info);
  if (p->plan) (void)rocfft_plan_destroy(p->plan);
  if (p->work) (void)pf_free(p->work);
  delete p;
}

}  // namespace pfhip
