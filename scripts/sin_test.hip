// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) against double
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const unsigned* ph, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float rev = (float)(int)ph[i] * (1.0f / 4294967296.0f);
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<unsigned> h(n);
  unsigned x = 12345;
  for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
  unsigned* d; float *s, *c;
  hipMalloc(&d, n * 4); hipMalloc(&s, n * 4); hipMalloc(&c, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, s, c, n);
  std::vector<float> hs(n), hc(n);
  hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0;
  for (int i = 0; i < n; ++i) {
    double rev = (double)(float)(int)h[i] / 4294967296.0;   // same rounded argument
    es = fmax(es, fabs(hs[i] - sin(2 * M_PI * rev)));
    ec = fmax(ec, fabs(hc[i] - cos(2 * M_PI * rev)));
  }
  printf("max abs err: sin %.3e cos %.3e\n", es, ec);
  return 0;
}
