// Can a latency-bound kernel of small workgroups (the WFM2 pilot loop: 2032 waves, ~40 VGPRs, no LDS, 0.47 ms of
// dependent VALU steps) run UNDERNEATH a persistent one-workgroup-per-CU kernel that owns the LDS and most of the
// registers (the IF decimator mixdec<1,16>: 1024 threads, 113 VGPRs, 159 KB LDS, ~0.8 ms)?  4 x 113 = 452 of a SIMD's
// 512 registers leave room for ONE 40-register wave per SIMD.  Stand-ins with the same footprints:
//   A  persistent, 256 workgroups x 1024 threads, dynamic LDS 159 KB, ~110 VGPRs held live, LDS reads + packed FMAs
//   B  nB workgroups x 64 threads, no LDS, a chain of dependent FMAs / v_sin / DPP adds
// measured: A alone, B alone, A then B on one stream, A || B on two streams with A launched first / B launched first
// (and the per-kernel spans from events).  Output: one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-vectorize -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -o scripts/diag/coresidency.bin scripts/diag/coresidency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void kernel_A(float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int i = tid; i < 159 * 128; i += 1024) lds[i] = make_float2((float)i * 1e-6f, 1.f);
  __syncthreads();
  v2f acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = (v2f){(float)k, 1.f};
  const float2* p = lds + (tid & 1023);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const float2* q = p + (it & 63) * 64;
#pragma unroll
    for (int k = 0; k < 28; ++k) {
      const float2 x = q[k * 96];
      const v2f xv = {x.x, x.y};
      acc[k] = __builtin_elementwise_fma(acc[k], (v2f){0.999f, 0.999f}, xv);
      if ((k & 7) == 7) asm volatile("" ::: "memory");     // keep at most 8 LDS reads in flight (registers)
    }
    if ((it & 63) == 63) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 28; ++k) s += acc[k].x + acc[k].y;
  if (s == 123.456f) sink[0] = s;
}

__device__ __forceinline__ float dpp_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}

__global__ __launch_bounds__(64) void kernel_B(float* sink, int steps, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  float th = (float)threadIdx.x * 0.01f + (float)blockIdx.x * 1e-4f, w = 0.001f;
  float keep[30];
#pragma unroll
  for (int k = 0; k < 30; ++k) keep[k] = th + (float)k;
  for (int i = 0; i < steps; ++i) {
    const float s = __builtin_amdgcn_sinf(th);
    float e = s * keep[i % 30];
    e += dpp_shr1(e);
    w = fmaf(1e-4f, e, w);
    th = th + fmaf(0.01f, e, w);
    th = th - floorf(th);
  }
  float s2 = th;
#pragma unroll
  for (int k = 0; k < 30; ++k) s2 += keep[k];
  if (s2 == 123.456f) sink[blockIdx.x] = s2;
}

int main(int argc, char** argv) {
  const int itersA = argc > 1 ? atoi(argv[1]) : 9000;
  const int stepsB = argc > 2 ? atoi(argv[2]) : 30000;
  const int nB = argc > 3 ? atoi(argv[3]) : 2032;
  const int bthreads = argc > 4 ? atoi(argv[4]) : 64;
  const int prio = argc > 5 ? atoi(argv[5]) : 0;
  float* sink;
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_A), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, a0, a1, b0, b1;
  for (hipEvent_t* e : {&e0, &e1, &a0, &a1, &b0, &b1}) CK(hipEventCreate(e));
  const size_t ldsA = 159 * 1024;
  auto A = [&](hipStream_t s) { hipLaunchKernelGGL(kernel_A, dim3(256), dim3(1024), ldsA, s, sink, itersA); };
  auto B = [&](hipStream_t s) { hipLaunchKernelGGL(kernel_B, dim3(nB * 64 / bthreads), dim3(bthreads), 0, s, sink, stepsB, prio); };
  auto ms = [&](hipEvent_t x, hipEvent_t y) { float t; CK(hipEventElapsedTime(&t, x, y)); return (double)t; };
  std::string js = "{\n";
  char buf[400];
  for (int rep = 0; rep < 3; ++rep) {
    // alone
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, sa)); A(sa); CK(hipEventRecord(a1, sa)); CK(hipStreamSynchronize(sa));
    const double tA = ms(a0, a1);
    CK(hipEventRecord(b0, sb)); B(sb); CK(hipEventRecord(b1, sb)); CK(hipStreamSynchronize(sb));
    const double tB = ms(b0, b1);
    // serial on one stream
    CK(hipEventRecord(e0, sa)); A(sa); B(sa); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
    const double tAB = ms(e0, e1);
    // two streams, A first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    CK(hipStreamWaitEvent(sb, e0, 0));
    CK(hipEventRecord(a0, sa)); A(sa); CK(hipEventRecord(a1, sa));
    CK(hipEventRecord(b0, sb)); B(sb); CK(hipEventRecord(b1, sb));
    CK(hipStreamWaitEvent(sa, b1, 0));
    CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
    const double tPar1 = ms(e0, e1), tPar1A = ms(a0, a1), tPar1B = ms(b0, b1);
    // two streams, B first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    CK(hipStreamWaitEvent(sb, e0, 0));
    CK(hipEventRecord(b0, sb)); B(sb); CK(hipEventRecord(b1, sb));
    CK(hipEventRecord(a0, sa)); A(sa); CK(hipEventRecord(a1, sa));
    CK(hipStreamWaitEvent(sa, b1, 0));
    CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
    const double tPar2 = ms(e0, e1), tPar2A = ms(a0, a1), tPar2B = ms(b0, b1);
    snprintf(buf, sizeof buf,
             " \"rep%d\": {\"A_alone_ms\": %.3f, \"B_alone_ms\": %.3f, \"A_then_B_ms\": %.3f, \"par_A_first_ms\": %.3f, \"par_A_first_A\": %.3f, "
             "\"par_A_first_B\": %.3f, \"par_B_first_ms\": %.3f, \"par_B_first_A\": %.3f, \"par_B_first_B\": %.3f},\n",
             rep, tA, tB, tAB, tPar1, tPar1A, tPar1B, tPar2, tPar2A, tPar2B);
    js += buf;
    fputs(buf, stderr);
  }
  snprintf(buf, sizeof buf, " \"itersA\": %d, \"stepsB\": %d, \"nB_waves\": %d, \"B_threads_per_wg\": %d, \"B_setprio\": %d\n}\n", itersA, stepsB, nB, bthreads, prio);
  js += buf;
  fputs(js.c_str(), stdout);
  return 0;
}
