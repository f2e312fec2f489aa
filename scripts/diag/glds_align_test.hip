// Does global_load_lds_dwordx4 (LDS-DMA) need a 16-byte aligned GLOBAL address?  The staging code of mixdec.hip /
// mixdec_mfma.hip assumes so (pairs of complex64 samples at even sample offsets).  This probe copies 64 x 16 bytes
// from src + off bytes for off = 0, 4, 8, 12 and compares.  Build: hipcc --offload-arch=gfx950 -O2 -o glds_align_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* src, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const int lane = threadIdx.x;
  const unsigned dst = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
  const char* g = src + lane * 16;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(g), "s"(dst) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int off = 0; off < 16; off += 4) {
    hipMemset(o, 0xff, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const char*)d + off, o);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (r[i] != (float)(i + off / 4)) ++bad;
    printf("offset %2d bytes: %s, %d of 256 words wrong (first words %g %g %g %g)\n", off, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
