// A HIP runtime made of the host heap, for the sanitizer build of the HOST half of libpysdr_hip.so
// (tests/host_san): "device" memory is malloc'ed (so AddressSanitizer sees every size the host code
// passes to a copy or to a kernel), copies are memcpy, streams and events complete at once.  Only the
// calls pysdr_amd/csrc/api.hip makes exist.  Nothing here is part of the product.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "fake hip error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

struct fake_stream { int id; };
struct fake_event { std::chrono::steady_clock::time_point t; bool recorded; };
typedef fake_stream* hipStream_t;
typedef fake_event* hipEvent_t;
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; size_t totalGlobalMem; };
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  std::strcpy(p->name, "fake gfx950");
  std::strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}

template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = static_cast<T*>(std::malloc(n ? n : 1)); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }

inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new fake_stream{0}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 1; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new fake_event{{}, false}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  if (!a->recorded || !b->recorded) return hipErrorInvalidValue;
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
