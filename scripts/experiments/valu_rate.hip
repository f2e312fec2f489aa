// Microbenchmark: issue rate of v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_add_f32 / v_sin_f32 on gfx950
// as a function of the waves per SIMD (1, 2, 4, 8).  hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, int iters, float a, float b) {
  float x[16];
  v2f p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{x[i], x[i] + 0.5f}; }
  const v2f pa = {a, a + 1e-3f}, pb = {b, b - 1e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pa));
      if (OP == 3) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
      if (OP == 4) asm volatile("v_sin_f32 %0, %0" : "+v"(x[i]));
      if (OP == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pa));
      if (OP == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
      if (OP == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(pa), "v"(pb));
      if (OP == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(p[i]) : "v"(pa), "v"(pb));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
double run(int waves_per_simd, float* d) {
  const int iters = 100000;
  const int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU, waves_per_simd on each of the 4 SIMDs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256, threads>>>(d, iters, 1.0001f, 0.5f);      // warm-up: clocks
  float ms = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<OP><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (t < ms) ms = t;
  }
  // wave-instructions per SIMD = waves_per_simd * iters * 16
  return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 16);   // cycles (at 2.4 GHz) per wave-instruction per SIMD
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  const char* names[9] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_f32", "v_sin_f32", "v_pk_mul_f32",
                          "v_pk_fma_f32(op_sel swap src1)", "v_pk_fma_f32(op_sel_hi bcast src0)", "v_pk_fma_f32(neg)"};
  for (int w : {1, 2, 3, 4}) {
    double r[9] = {run<0>(w, d), run<1>(w, d), run<2>(w, d), run<3>(w, d), run<4>(w, d), run<5>(w, d),
                   run<6>(w, d), run<7>(w, d), run<8>(w, d)};
    printf("waves/SIMD %d:", w);
    for (int i = 0; i < 9; ++i) printf("  %s %.2f", names[i], r[i]);
    printf("   (cycles @2.4 GHz per wave-instruction per SIMD)\n");
  }
  return 0;
}
