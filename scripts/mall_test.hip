// Microbenchmark: does a write -> read round trip of a buffer stay on-die (L2 / Infinity
// Cache) when it is small enough?  Alternates a streaming write kernel and a streaming read
// kernel over buffers of different sizes and prints the effective bandwidth of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void wr(float4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ void rd(const float4* p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  for (size_t mb : {8, 16, 32, 64, 128, 192, 256, 512, 2048}) {
    size_t bytes = mb << 20, n = bytes / 16;
    float4* p; hipMalloc(&p, bytes);
    float tw = 0, tr = 0; int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
      hipEventRecord(e0);
      wr<<<2048, 256>>>(p, n, (float)r);
      hipEventRecord(e1);
      rd<<<2048, 256>>>(p, n, out);
      hipEventRecord(e2);
      hipEventSynchronize(e2);
      float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
      if (r >= 3) { tw += a; tr += b; }
    }
    printf("%5zu MB: write %.0f GB/s, read-after-write %.0f GB/s\n", mb, bytes / (tw / reps * 1e-3) / 1e9, bytes / (tr / reps * 1e-3) / 1e9);
    hipFree(p);
  }
  return 0;
}
