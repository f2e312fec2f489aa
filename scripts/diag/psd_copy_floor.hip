// Copy-floor microbenchmark for the two-kernel 64k PSD: kernels that move exactly
// the bytes psd_cols / psd_rows move -- per frame 256 KB in + 512 KB out, then 512 KB in + 256 KB
// out, in groups of 448 frames over a 224 MB intermediate -- and compute nothing.  Two address
// patterns each:
//   P  the product's: 8-byte (float2) lanes, 128-byte row pieces 2 KB apart on the input, 512-byte
//      wave pieces on the intermediate, 4-byte non-temporal stores of the PSD in 128-byte pieces
//   C  16 bytes per lane, every wave access 1 KiB contiguous
// plus plain streams (read from HBM, read from a just-written 224 MB buffer = Infinity Cache,
// write) for the rates the fabric gives.  Output: one JSON object (profiles/r03_psd_copy_floor.json).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/diag/psd_copy_floor.bin scripts/diag/psd_copy_floor.hip
//   scripts/diag/psd_copy_floor.bin [nframes=10666] [group=448] [reps=5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define AS1 __attribute__((address_space(1)))
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kN = 65536, kM = 32768;

template <bool NT> __device__ __forceinline__ v2f ld2(const float2* p) {
  if (NT) return __builtin_nontemporal_load((const AS1 v2f*)p);
  return *(const AS1 v2f*)p;
}
template <bool NT> __device__ __forceinline__ v4f ld4(const float4* p) {
  if (NT) return __builtin_nontemporal_load((const AS1 v4f*)p);
  return *(const AS1 v4f*)p;
}
template <bool NT> __device__ __forceinline__ void st4(float4* p, v4f v) {
  if (NT) __builtin_nontemporal_store(v, (AS1 v4f*)p); else *(AS1 v4f*)p = v;
}
__device__ __forceinline__ void st2(float2* p, v2f v) { *(AS1 v2f*)p = v; }
template <bool NT> __device__ __forceinline__ void st1(float* p, float v) {
  if (NT) __builtin_nontemporal_store(v, (AS1 float*)p); else *(AS1 float*)p = v;
}

// ---- the product's address pattern (psdfft.hip cols_unit / rows_unit), arithmetic removed
__global__ __launch_bounds__(256) void cols_P(const float2* __restrict__ x, size_t hop, const float* __restrict__ win,
                                              float2* __restrict__ work) {
  const int f = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
  const int b = tid & 15, hi = tid >> 4, bb = cb * 16 + b;
  const float2* xf = x + (size_t)f * hop;
  v2f u[8];
#pragma unroll
  for (int a1 = 0; a1 < 8; ++a1) {
    const int n = 256 * (hi + 16 * a1) + bb;
    u[a1] = ld2<true>(xf + n) * win[n];
  }
  float2* o = work + (size_t)f * kN + (size_t)cb * 4096 + hi * 16 + b;
#pragma unroll
  for (int p0 = 0; p0 < 16; ++p0) st2(o + p0 * 256, u[p0 & 7]);
}
__global__ __launch_bounds__(512) void rows_P(const float2* __restrict__ work, float* __restrict__ out) {
  const int f = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
  const int c0 = tid & 15, pl = tid >> 4;
  const float2* src = work + (size_t)f * kN + (size_t)(rb * 32 + pl) * 16 + c0;
  v2f u[16];
#pragma unroll
  for (int c1 = 0; c1 < 16; ++c1) u[c1] = ld2<false>(src + 4096 * c1);
  const int pl2 = tid & 31, q1 = tid >> 5;
  const int kb = rb * 32 + pl2 + 256 * q1;
  float* of = out + (size_t)f * kN;
#pragma unroll
  for (int q0 = 0; q0 < 16; ++q0) {
    const int k = kb + 4096 * q0;
    st1<true>(of + ((k + kM) & (kN - 1)), u[q0].x + u[q0].y);
  }
}

// ---- 16 bytes per lane, 1 KiB per wave access.  One workgroup moves the same share of a frame
// as the product's (cols: 16 KB in -> 32 KB out; rows: 64 KB in -> 32 KB out).
template <bool NTIN, bool NTWORK>
__global__ __launch_bounds__(256) void cols_C(const float4* __restrict__ x, size_t hop4, float4* __restrict__ work) {
  const int f = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
  const float4* xf = x + (size_t)f * hop4 + cb * 1024;          // 16 KB = 1024 float4
  float4* o = work + (size_t)f * (kN / 2) + cb * 2048;          // 32 KB = 2048 float4
  v4f u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = ld4<NTIN>(xf + i * 256 + tid);
#pragma unroll
  for (int i = 0; i < 8; ++i) st4<NTWORK>(o + i * 256 + tid, u[i & 3] * (float)(i + 1));
}
template <bool NTWORK, bool NTOUT>
__global__ __launch_bounds__(512) void rows_C(const float4* __restrict__ work, float4* __restrict__ out) {
  const int f = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
  const float4* src = work + (size_t)f * (kN / 2) + rb * 4096;  // 64 KB = 4096 float4
  float4* o = out + (size_t)f * (kN / 4) + rb * 2048;           // 32 KB = 2048 float4
  v4f u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = ld4<NTWORK>(src + i * 512 + tid);
#pragma unroll
  for (int i = 0; i < 4; ++i) st4<NTOUT>(o + i * 512 + tid, u[i] + u[i + 4]);
}

// ---- mixed: which side of each kernel carries the difference between P and C?
__global__ __launch_bounds__(256) void cols_PinCout(const float2* __restrict__ x, size_t hop, const float* __restrict__ win,
                                                    float4* __restrict__ work) {
  const int f = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
  const int b = tid & 15, hi = tid >> 4, bb = cb * 16 + b;
  const float2* xf = x + (size_t)f * hop;
  v2f u[8];
#pragma unroll
  for (int a1 = 0; a1 < 8; ++a1) {
    const int n = 256 * (hi + 16 * a1) + bb;
    u[a1] = ld2<true>(xf + n) * win[n];
  }
  float4* o = work + (size_t)f * (kN / 2) + cb * 2048;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v4f w = {u[i].x, u[i].y, u[(i + 1) & 7].x, u[(i + 1) & 7].y}; st4<false>(o + i * 256 + tid, w); }
}
__global__ __launch_bounds__(256) void cols_CinPout(const float4* __restrict__ x, size_t hop4, float2* __restrict__ work) {
  const int f = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
  const float4* xf = x + (size_t)f * hop4 + cb * 1024;
  v4f u[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = ld4<true>(xf + i * 256 + tid);
  float2* o = work + (size_t)f * kN + (size_t)cb * 4096 + (tid >> 4) * 16 + (tid & 15);
#pragma unroll
  for (int p0 = 0; p0 < 16; ++p0) { v2f w = {u[p0 & 3].x, u[p0 & 3].y + (float)p0}; st2(o + p0 * 256, w); }
}
__global__ __launch_bounds__(512) void rows_PinCout(const float2* __restrict__ work, float4* __restrict__ out) {
  const int f = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
  const int c0 = tid & 15, pl = tid >> 4;
  const float2* src = work + (size_t)f * kN + (size_t)(rb * 32 + pl) * 16 + c0;
  v2f u[16];
#pragma unroll
  for (int c1 = 0; c1 < 16; ++c1) u[c1] = ld2<false>(src + 4096 * c1);
  float4* o = out + (size_t)f * (kN / 4) + rb * 2048;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v4f w = {u[4 * i].x, u[4 * i + 1].x, u[4 * i + 2].y, u[4 * i + 3].y}; st4<true>(o + i * 512 + tid, w); }
}
__global__ __launch_bounds__(512) void rows_CinPout(const float4* __restrict__ work, float* __restrict__ out) {
  const int f = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
  const float4* src = work + (size_t)f * (kN / 2) + rb * 4096;
  v4f u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = ld4<false>(src + i * 512 + tid);
  const int pl2 = tid & 31, q1 = tid >> 5;
  const int kb = rb * 32 + pl2 + 256 * q1;
  float* of = out + (size_t)f * kN;
#pragma unroll
  for (int q0 = 0; q0 < 16; ++q0) {
    const int k = kb + 4096 * q0;
    st1<true>(of + ((k + kM) & (kN - 1)), u[q0 & 7].x + u[q0 & 7].y + (float)q0);
  }
}

// ---- what in the columns kernel's LOAD side costs: the 4-byte window loads, or the 128-byte row pieces?
//   WIN: 0 = no window, 1 = eight 4-byte loads per thread (product), 2 = two 16-byte loads from a table laid out per thread
//   COLS: columns per workgroup (16 = 128-byte pieces, 32 = 256-byte pieces); threads = 16 * COLS
template <int WIN, int COLS>
__global__ __launch_bounds__(16 * COLS) void cols_V(const float2* __restrict__ x, size_t hop, const float* __restrict__ win,
                                                     float2* __restrict__ work) {
  const int f = blockIdx.y, cb = blockIdx.x, tid = threadIdx.x;
  const int b = tid % COLS, hi = tid / COLS, bb = cb * COLS + b;
  const float2* xf = x + (size_t)f * hop;
  float w[8];
  if (WIN == 2) {
    const v4f w0 = *(const AS1 v4f*)(win + ((size_t)cb * (16 * COLS) + tid) * 8), w1 = *(const AS1 v4f*)(win + ((size_t)cb * (16 * COLS) + tid) * 8 + 4);
    w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
  }
  v2f u[8];
#pragma unroll
  for (int a1 = 0; a1 < 8; ++a1) {
    const int n = 256 * (hi + 16 * a1) + bb;
    u[a1] = ld2<true>(xf + n);
    if (WIN == 1) u[a1] *= win[n];
    if (WIN == 2) u[a1] *= w[a1];
  }
  float2* o = work + (size_t)f * kN + (size_t)cb * (256 * COLS) + tid;
#pragma unroll
  for (int p0 = 0; p0 < 16; ++p0) st2(o + p0 * (16 * COLS), u[p0 & 7]);
}

// ---- plain streams
template <bool NT> __global__ __launch_bounds__(256) void rd_stream(const float4* __restrict__ p, size_t n, float* sink) {
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v4f a = ld4<NT>(p + i), b = ld4<NT>(p + i + stride), c = ld4<NT>(p + i + 2 * stride), d = ld4<NT>(p + i + 3 * stride);
    acc += a + b + c + d;
  }
  for (; i < n; i += stride) acc += ld4<NT>(p + i);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
template <bool NT> __global__ __launch_bounds__(256) void wr_stream(float4* __restrict__ p, size_t n, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  const v4f w = {v, v + 1.f, v + 2.f, v + 3.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st4<NT>(p + i, w);
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  void start() { CK(hipEventRecord(a)); }
  double stop_ms() { CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

int main(int argc, char** argv) {
  const int nframes = argc > 1 ? atoi(argv[1]) : 10666;
  const int group = argc > 2 ? atoi(argv[2]) : 448;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const size_t hop = kM;
  float2 *x, *work; float *out, *win, *sink;
  const size_t xbytes = (size_t)nframes * hop * 8, wbytes = (size_t)group * kN * 8, obytes = (size_t)nframes * kN * 4;
  CK(hipMalloc(&x, xbytes)); CK(hipMalloc(&work, wbytes)); CK(hipMalloc(&out, obytes));
  CK(hipMalloc(&win, kM * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(x, 0x3c, xbytes)); CK(hipMemset(work, 0, wbytes)); CK(hipMemset(out, 0, obytes)); CK(hipMemset(win, 0x3c, kM * 4));
  Timer t;
  std::string js = "{\n";
  char buf[512];
  auto best = [&](auto&& fn) { double m = 1e30; for (int r = 0; r < reps + 1; ++r) { t.start(); fn(); const double ms = t.stop_ms(); if (r) m = std::min(m, ms); } return m; };

  // plain streams
  {
    const size_t nx = xbytes / 16, nw = wbytes / 16;
    const double r_hbm = best([&] { rd_stream<false><<<4096, 256>>>((const float4*)x, nx, sink); });
    const double r_hbm_nt = best([&] { rd_stream<true><<<4096, 256>>>((const float4*)x, nx, sink); });
    const double w_hbm = best([&] { wr_stream<false><<<4096, 256>>>((float4*)out, obytes / 16, 1.f); });
    const double w_hbm_nt = best([&] { wr_stream<true><<<4096, 256>>>((float4*)out, obytes / 16, 1.f); });
    // write the 224 MB buffer, then read it back: the read comes from the Infinity Cache
    double w_mall = 1e30, r_mall = 1e30;
    for (int r = 0; r < reps + 1; ++r) {
      t.start(); wr_stream<false><<<4096, 256>>>((float4*)work, nw, (float)r); const double a = t.stop_ms();
      t.start(); rd_stream<false><<<4096, 256>>>((const float4*)work, nw, sink); const double b = t.stop_ms();
      if (r) { w_mall = std::min(w_mall, a); r_mall = std::min(r_mall, b); }
    }
    snprintf(buf, sizeof buf,
             " \"streams_TBps\": {\"read_hbm\": %.3f, \"read_hbm_nt\": %.3f, \"write_hbm\": %.3f, \"write_hbm_nt\": %.3f, "
             "\"write_224MB\": %.3f, \"read_224MB_after_write\": %.3f},\n",
             xbytes / r_hbm / 1e9, xbytes / r_hbm_nt / 1e9, obytes / w_hbm / 1e9, obytes / w_hbm_nt / 1e9, wbytes / w_mall / 1e9,
             wbytes / r_mall / 1e9);
    js += buf;
  }

  auto run_pair = [&](const char* name, auto&& lc, auto&& lr) {
    // whole job, grouped as the product groups it
    const double pair = best([&] {
      for (int f0 = 0; f0 < nframes; f0 += group) { const int nf = std::min(group, nframes - f0); lc(f0, nf); lr(f0, nf); }
    });
    // each kernel alone over the same groups (the intermediate is the same 224 MB every time)
    const double c = best([&] { for (int f0 = 0; f0 < nframes; f0 += group) lc(f0, std::min(group, nframes - f0)); });
    const double r = best([&] { for (int f0 = 0; f0 < nframes; f0 += group) lr(f0, std::min(group, nframes - f0)); });
    const double per = 1e6 / nframes;
    snprintf(buf, sizeof buf,
             " \"%s\": {\"pair_ns_per_frame\": %.1f, \"cols_ns_per_frame\": %.1f, \"rows_ns_per_frame\": %.1f, "
             "\"pair_TBps_of_1.5MB\": %.3f, \"cols_TBps_of_768KB\": %.3f, \"rows_TBps_of_768KB\": %.3f, \"frac_of_8TBps_on_512KB\": %.3f},\n",
             name, pair * per, c * per, r * per, 1.5 * 1048576 / (pair * per) / 1e3, 0.75 * 1048576 / (c * per) / 1e3,
             0.75 * 1048576 / (r * per) / 1e3, 0.5 * 1048576 / (pair * per) / 1e3 / 8.0);
    js += buf;
  };
  run_pair("P_product_pattern",
           [&](int f0, int nf) { cols_P<<<dim3(16, nf), 256>>>(x + (size_t)f0 * hop, hop, win, work); },
           [&](int f0, int nf) { rows_P<<<dim3(8, nf), 512>>>(work, out + (size_t)f0 * kN); });
  run_pair("C_16B_nt_streams",
           [&](int f0, int nf) { cols_C<true, false><<<dim3(16, nf), 256>>>((const float4*)(x + (size_t)f0 * hop), hop / 2, (float4*)work); },
           [&](int f0, int nf) { rows_C<false, true><<<dim3(8, nf), 512>>>((const float4*)work, (float4*)(out + (size_t)f0 * kN)); });
  run_pair("C_16B_no_hints",
           [&](int f0, int nf) { cols_C<false, false><<<dim3(16, nf), 256>>>((const float4*)(x + (size_t)f0 * hop), hop / 2, (float4*)work); },
           [&](int f0, int nf) { rows_C<false, false><<<dim3(8, nf), 512>>>((const float4*)work, (float4*)(out + (size_t)f0 * kN)); });
  run_pair("C_16B_nt_everywhere",
           [&](int f0, int nf) { cols_C<true, true><<<dim3(16, nf), 256>>>((const float4*)(x + (size_t)f0 * hop), hop / 2, (float4*)work); },
           [&](int f0, int nf) { rows_C<true, true><<<dim3(8, nf), 512>>>((const float4*)work, (float4*)(out + (size_t)f0 * kN)); });
  run_pair("mixed_colsPinCout_rowsPinCout",
           [&](int f0, int nf) { cols_PinCout<<<dim3(16, nf), 256>>>(x + (size_t)f0 * hop, hop, win, (float4*)work); },
           [&](int f0, int nf) { rows_PinCout<<<dim3(8, nf), 512>>>(work, (float4*)(out + (size_t)f0 * kN)); });
  run_pair("mixed_colsCinPout_rowsCinPout",
           [&](int f0, int nf) { cols_CinPout<<<dim3(16, nf), 256>>>((const float4*)(x + (size_t)f0 * hop), hop / 2, work); },
           [&](int f0, int nf) { rows_CinPout<<<dim3(8, nf), 512>>>((const float4*)work, out + (size_t)f0 * kN); });
  {
    // columns-kernel load side: per-frame ns of the cols kernel alone
    auto cols_only = [&](auto&& lc) { return best([&] { for (int f0 = 0; f0 < nframes; f0 += group) lc(f0, std::min(group, nframes - f0)); }) * 1e6 / nframes; };
    const double v00 = cols_only([&](int f0, int nf) { cols_V<0, 16><<<dim3(16, nf), 256>>>(x + (size_t)f0 * hop, hop, win, work); });
    const double v10 = cols_only([&](int f0, int nf) { cols_V<1, 16><<<dim3(16, nf), 256>>>(x + (size_t)f0 * hop, hop, win, work); });
    const double v20 = cols_only([&](int f0, int nf) { cols_V<2, 16><<<dim3(16, nf), 256>>>(x + (size_t)f0 * hop, hop, win, work); });
    const double v01 = cols_only([&](int f0, int nf) { cols_V<0, 32><<<dim3(8, nf), 512>>>(x + (size_t)f0 * hop, hop, win, work); });
    const double v11 = cols_only([&](int f0, int nf) { cols_V<1, 32><<<dim3(8, nf), 512>>>(x + (size_t)f0 * hop, hop, win, work); });
    const double v21 = cols_only([&](int f0, int nf) { cols_V<2, 32><<<dim3(8, nf), 512>>>(x + (size_t)f0 * hop, hop, win, work); });
    snprintf(buf, sizeof buf,
             " \"cols_load_side_ns_per_frame\": {\"16cols_no_window\": %.1f, \"16cols_window_8x4B\": %.1f, \"16cols_window_2x16B\": %.1f, "
             "\"32cols_no_window\": %.1f, \"32cols_window_8x4B\": %.1f, \"32cols_window_2x16B\": %.1f},\n", v00, v10, v20, v01, v11, v21);
    js += buf;
  }
  {
    // the same copies with the groups dealt over TWO streams, each with half a group of intermediate (what the
    // product does since round 3): columns of one half-group beside the rows of the other
    hipStream_t s2[2];
    hipEvent_t fork, join;
    CK(hipStreamCreateWithFlags(&s2[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2[1], hipStreamNonBlocking));
    CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
    const int half = group / 2;
    float2* wk[2] = {work, work + (size_t)half * kN};
    auto two = [&](bool cpat) {
      double m = 1e30;
      for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(t.a, s2[0]));
        CK(hipEventRecord(fork, s2[0])); CK(hipStreamWaitEvent(s2[1], fork, 0));
        int k = 0;
        for (int f0 = 0; f0 < nframes; f0 += half, ++k) {
          const int nf = std::min(half, nframes - f0), w = k & 1;
          if (cpat) {
            cols_C<true, false><<<dim3(16, nf), 256, 0, s2[w]>>>((const float4*)(x + (size_t)f0 * hop), hop / 2, (float4*)wk[w]);
            rows_C<false, true><<<dim3(8, nf), 512, 0, s2[w]>>>((const float4*)wk[w], (float4*)(out + (size_t)f0 * kN));
          } else {
            cols_P<<<dim3(16, nf), 256, 0, s2[w]>>>(x + (size_t)f0 * hop, hop, win, wk[w]);
            rows_P<<<dim3(8, nf), 512, 0, s2[w]>>>(wk[w], out + (size_t)f0 * kN);
          }
        }
        CK(hipEventRecord(join, s2[1])); CK(hipStreamWaitEvent(s2[0], join, 0));
        CK(hipEventRecord(t.b, s2[0])); CK(hipEventSynchronize(t.b));
        float ms; CK(hipEventElapsedTime(&ms, t.a, t.b));
        if (r) m = std::min(m, (double)ms);
      }
      return m * 1e6 / nframes;
    };
    const double p2 = two(false), c2 = two(true);
    snprintf(buf, sizeof buf, " \"two_streams_half_groups_pair_ns_per_frame\": {\"P_product_pattern\": %.1f, \"C_16B_nt_streams\": %.1f, "
             "\"frac_of_8TBps_on_512KB_P\": %.3f, \"frac_of_8TBps_on_512KB_C\": %.3f},\n", p2, c2,
             0.5 * 1048576 / p2 / 1e3 / 8.0, 0.5 * 1048576 / c2 / 1e3 / 8.0);
    js += buf;
  }
  snprintf(buf, sizeof buf, " \"nframes\": %d, \"group\": %d, \"reps\": %d, \"timing\": \"best of reps, hipEvents\"\n}\n", nframes, group, reps);
  js += buf;
  fputs(js.c_str(), stdout);
  return 0;
}
