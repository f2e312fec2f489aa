// What rate does a chain of v_mfma_f32_16x16x4_f32 reach on gfx950 in the shapes mixdec_mfma.hip can use?  (VERDICT r3:
// "measure first".)  One workgroup per CU, W waves, every wave runs STEPS x { [ds_read_b64] ; mfma(acc[i]) ; mfma(acc[j]) }
// REPS times; variants: operands from registers only / one 8-byte LDS read per two MFMAs through a ring of 8; 2 or 4
// accumulators; 1, 2 or 4 waves per SIMD; + V plain VALU FMAs per step (is vector work hidden beside an f32 MFMA?).
// Prints cycles per MFMA per SIMD (s_memtime) and the TFLOP/s of the whole chip by the wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate.bin mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) f2* lds_cf2;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

template <int NACC, bool LDS, int NV>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) float2 buf[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += blockDim.x) buf[i] = make_float2(1e-3f * (i & 63), 1e-3f);
  __syncthreads();
  constexpr int STEPS = 18;
  float B1[STEPS], B2[STEPS];
  for (int i = 0; i < STEPS; ++i) { B1[i] = 1e-3f * (lane + i); B2[i] = 1e-3f * (lane - i); }
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float vsum = 0.f;
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)buf + (unsigned)((lane & 15) * 2064 % 32768 + (lane >> 4) * 8);
  const lds_cf2 p = (lds_cf2)(size_t)base;
  f2 ring[8];
  for (int i = 0; i < 8; ++i) ring[i] = LDS ? p[4 * i] : (f2){1e-3f * lane, 2e-3f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int ls = 0; ls < STEPS; ++ls) {
      const f2 v = ring[ls % 8];
      acc[(2 * ls) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, B1[ls], acc[(2 * ls) % NACC], 0, 0, 0);
      acc[(2 * ls + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, B2[ls], acc[(2 * ls + 1) % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) vsum = fmaf(v.x, vsum, v.y);
      if (LDS) ring[ls % 8] = p[4 * ((ls + 8) % 24)];
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (LDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3] + vsum;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
}

template <int NACC, bool LDS, int NV>
int run(const char* name, int waves, int ncu, float* out, unsigned long long* cyc) {
  const int reps = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NACC, LDS, NV>), dim3(ncu), dim3(64 * waves), 0, 0, out, cyc, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, LDS, NV>), dim3(ncu), dim3(64 * waves), 0, 0, out, cyc, reps);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(ncu * waves);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double c = 0;
  for (auto v : h) c += (double)v;
  c /= h.size();
  const double nm = 36.0 * reps;                       // MFMAs per wave
  const double per_simd = c / (nm * (waves / 4.0));    // cycles per MFMA per SIMD (waves/4 waves share a SIMD)
  const double tf = nm * waves * ncu * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-44s waves/SIMD %d: %6.1f cycles per MFMA per SIMD, %6.1f TFLOP/s, clock %.2f GHz\n", name, waves / 4, per_simd, tf, c / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  int ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)ncu * 1024 * 4)); CK(hipMalloc(&cyc, (size_t)ncu * 16 * 8));
  for (int w : {4, 8, 16}) {
    run<2, false, 0>("registers, 2 accumulators", w, ncu, out, cyc);
    run<4, false, 0>("registers, 4 accumulators", w, ncu, out, cyc);
    run<2, true, 0>("LDS ring, 2 accumulators", w, ncu, out, cyc);
    run<4, true, 0>("LDS ring, 4 accumulators", w, ncu, out, cyc);
    run<2, true, 3>("LDS ring, 2 accumulators, 3 VALU per step", w, ncu, out, cyc);
    run<2, false, 3>("registers, 2 accumulators, 3 VALU per step", w, ncu, out, cyc);
    run<2, false, 8>("registers, 2 accumulators, 8 VALU per step", w, ncu, out, cyc);
  }
  return 0;
}
