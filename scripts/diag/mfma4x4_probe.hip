// v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout and issue rate (one wave per SIMD, 1 / 2 / 4 accumulators).
//   hipcc --offload-arch=gfx950 -O3 scripts/diag/mfma4x4_probe.hip -o scripts/diag/mfma4x4_probe.bin && scripts/diag/mfma4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {            // D = A x B with A = 1 + lane, B = 100 * (1 + lane): which lanes meet where
  const int lane = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(1000 * (1 + lane)), c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[v * 64 + lane] = c[v];
}

template <int NACC>
__global__ void rate(float* out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  f4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = (f4){0.f, 0.f, 0.f, 0.f};
  float x = 1.0f + lane * 1e-3f, y = 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[a], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run_rate(float* d_out, unsigned long long* d_cyc, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate<NACC><<<256, 256 * waves_per_simd>>>(d_out, 10, d_cyc);
  hipEventRecord(e0);
  rate<NACC><<<256, 256 * waves_per_simd>>>(d_out, iters, d_cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NACC;
  printf("acc %d, %d wave(s)/SIMD: %.2f ns per MFMA per wave (%.1f s_memtime ticks); %.1f TFLOP/s chip\n", NACC, waves_per_simd, ms * 1e6 / n,
         (double)cyc / n, 256.0 * 4 * waves_per_simd * n * 512 / (ms * 1e-3) / 1e12);
}

int main() {
  float* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_cyc, 8);
  layout<<<1, 64>>>(d_out);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d_out, 1024, hipMemcpyDeviceToHost);
  // D[v][lane] = a * b with a = 1 + la, b = 1000 * (1 + lb): decode (la, lb)
  for (int v = 0; v < 4; ++v) {
    printf("vgpr %d:", v);
    for (int lane = 0; lane < 64; ++lane) {
      const long p = (long)(h[v * 64 + lane] + 0.5f);
      int la = -1, lb = -1;
      for (int b = 1; b <= 64 && la < 0; ++b) if (p % (1000L * b) == 0 && p / (1000L * b) >= 1 && p / (1000L * b) <= 64) { /* ambiguous: pick consistent with block */ }
      // unambiguous decode: try the expected layout first
      const int blk = lane >> 2, j = lane & 3;
      const long want = (long)(1 + (blk * 4 + v)) * 1000L * (1 + (blk * 4 + j));
      printf(" %s", p == want ? "ok" : "??");
      (void)la; (void)lb;
    }
    printf("\n");
  }
  printf("(ok = D[vgpr i][lane 4*blk + j] = A[lane 4*blk + i] * B[lane 4*blk + j])\n");
  for (int w = 1; w <= 2; ++w) { run_rate<1>(d_out, d_cyc, w); run_rate<2>(d_out, d_cyc, w); run_rate<4>(d_out, d_cyc, w); }
  return 0;
}
