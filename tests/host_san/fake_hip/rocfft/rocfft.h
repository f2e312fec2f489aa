// rocFFT as far as pysdr_amd/csrc/api.hip uses it, doing nothing (tests/host_san: the transform itself
// is device work; the plan / buffer bookkeeping around it is what the sanitizers look at)
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
typedef int rocfft_status;
constexpr rocfft_status rocfft_status_success = 0;
struct rocfft_plan_t { size_t n; size_t batch; };
struct rocfft_execution_info_t { void* work; size_t work_bytes; hipStream_t st; };
typedef rocfft_plan_t* rocfft_plan;
typedef rocfft_execution_info_t* rocfft_execution_info;
enum rocfft_result_placement { rocfft_placement_inplace, rocfft_placement_notinplace };
enum rocfft_transform_type { rocfft_transform_type_complex_forward };
enum rocfft_precision { rocfft_precision_single };
inline rocfft_status rocfft_setup() { return 0; }
inline rocfft_status rocfft_cleanup() { return 0; }
inline rocfft_status rocfft_plan_create(rocfft_plan* p, rocfft_result_placement, rocfft_transform_type, rocfft_precision,
                                        size_t dims, const size_t* lengths, size_t batch, const void*) {
  *p = new rocfft_plan_t{dims ? lengths[0] : 0, batch};
  return 0;
}
inline rocfft_status rocfft_plan_destroy(rocfft_plan p) { delete p; return 0; }
inline rocfft_status rocfft_plan_get_work_buffer_size(rocfft_plan, size_t* n) { *n = 4096; return 0; }
inline rocfft_status rocfft_execution_info_create(rocfft_execution_info* i) { *i = new rocfft_execution_info_t{nullptr, 0, nullptr}; return 0; }
inline rocfft_status rocfft_execution_info_destroy(rocfft_execution_info i) { delete i; return 0; }
inline rocfft_status rocfft_execution_info_set_work_buffer(rocfft_execution_info i, void* w, size_t n) { i->work = w; i->work_bytes = n; return 0; }
inline rocfft_status rocfft_execution_info_set_stream(rocfft_execution_info i, void* s) { i->st = (hipStream_t)s; return 0; }
inline rocfft_status rocfft_execute(rocfft_plan p, void** in, void**, rocfft_execution_info) {
  // touch the whole in-place buffer the plan was made for: n x batch complex floats
  float2* b = static_cast<float2*>(in[0]);
  for (size_t i = 0; i < p->n * p->batch; ++i) b[i].x += 0.f;
  return 0;
}
