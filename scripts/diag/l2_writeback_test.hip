// Does a global store that is overwritten while its line is still in the XCD's L2 reach the fabric
// once or every time?  (whether an L2-resident four-step intermediate could ever
// save the intermediate's WRITE traffic.)  Run under `rocprofv3 --pmc WRITE_SIZE` and, in a second
// pass, `--pmc FETCH_SIZE`: the kernels below are told apart by name.
//   rewrite_x1 / rewrite_x16   every workgroup writes its own 64 KB (16 MB in all = 2 MB per XCD) 1 / 16 times
//   interleave_4B              8 workgroups of one XCD (blockIdx b, b+8, ..) write the floats 8k+j of a shared region
//   contiguous_4B              the same bytes, every workgroup its own contiguous part
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/diag/l2_writeback_test.bin scripts/diag/l2_writeback_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R> __device__ __forceinline__ void rewrite_body(float4* buf) {
  float4* p = buf + (size_t)blockIdx.x * 4096;            // 64 KB per workgroup
  for (int r = 0; r < R; ++r) {
    const float v = (float)(r + 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i * 256 + threadIdx.x] = make_float4(v, v + i, v, v);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void rewrite_x1(float4* buf) { rewrite_body<1>(buf); }
__global__ __launch_bounds__(256) void rewrite_x16(float4* buf) { rewrite_body<16>(buf); }

// teams of 8 workgroups with equal blockIdx % 8 (one XCD under round-robin placement): team t = blocks
// 64 t' .. : block b -> xcd = b % 8, slot = b / 8; team = slot / 8, member j = slot % 8
__global__ __launch_bounds__(256) void interleave_4B(float* buf) {
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, team = slot >> 3, j = slot & 7;
  float* region = buf + ((size_t)(team * 8 + xcd)) * (8 * 16384);   // 8 members x 64 KB
  for (int i = 0; i < 64; ++i) {
    const int k = i * 256 + threadIdx.x;                   // 16384 floats per member
    region[8 * k + j] = (float)(k + j);
  }
}
__global__ __launch_bounds__(256) void contiguous_4B(float* buf) {
  float* p = buf + (size_t)blockIdx.x * 16384;
  for (int i = 0; i < 64; ++i) p[i * 256 + threadIdx.x] = (float)i;
}

int main() {
  float4* buf;
  CK(hipMalloc(&buf, 64u << 20));
  CK(hipMemset(buf, 0, 64u << 20));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto t = [&](const char* name, auto&& fn) {
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r) best = ms < best ? ms : best; }
    printf("%-16s %.2f us\n", name, best * 1e3);
  };
  t("rewrite_x1", [&] { rewrite_x1<<<256, 256>>>(buf); });
  t("rewrite_x16", [&] { rewrite_x16<<<256, 256>>>(buf); });
  t("interleave_4B", [&] { interleave_4B<<<512, 256>>>((float*)buf); });
  t("contiguous_4B", [&] { contiguous_4B<<<512, 256>>>((float*)buf); });
  return 0;
}
