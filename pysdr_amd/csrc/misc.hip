// Stand-alone NCO mixer (signal_generator.quad_mixer, receiver.py:552-553) and the
// element-wise halves of spectrum.periodogram (Plotting.py:462; formula pinned by
// rtty.py:839-841): window + zero-pad before the FFT, |X|^2 -> dB + fftshift after it.
#include "common.h"

namespace pysdr {

namespace {

// y[n] = x[n] * exp(j*2*pi*(phase0 + fword*n)/2^32); 2 samples (16 B) per lane.
__global__ __launch_bounds__(256) void quad_mixer_kernel(const float2* __restrict__ x,
                                                         float2* __restrict__ y, size_t n,
                                                         uint32_t phase0, uint32_t fword) {
  const size_t npairs = (n + 1) / 2;
  for (size_t pi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pi < npairs;
       pi += (size_t)gridDim.x * blockDim.x) {
    const size_t i = 2 * pi;
    const uint32_t ph0 = phase0 + fword * (uint32_t)i;
    const uint32_t ph1 = ph0 + fword;
    // v_sin_f32 / v_cos_f32 take revolutions: the 32-bit phase maps exactly (1.2e-7 abs error)
    const float r0 = (float)(int)ph0 * (1.0f / 4294967296.0f);
    const float r1 = (float)(int)ph1 * (1.0f / 4294967296.0f);
    const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
    const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
    if (i + 1 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      float4 o;
      o.x = v.x * c0 - v.y * s0;
      o.y = v.x * s0 + v.y * c0;
      o.z = v.z * c1 - v.w * s1;
      o.w = v.z * s1 + v.w * c1;
      *reinterpret_cast<float4*>(y + i) = o;
    } else {
      const float2 v = x[i];
      y[i] = make_float2(v.x * c0 - v.y * s0, v.x * s0 + v.y * c0);
    }
  }
}

// convolver.convolve_fast (receiver.py:207,216,862): y[i] = sum_k h[k] * xx[i + nt-1 - k],
// xx = [history (nt-1) | new samples].  Audio-rate data (1024 samples per call): one output
// per lane, taps through the scalar cache, input from L2.
__global__ __launch_bounds__(256) void fir_real_kernel(const float* __restrict__ xx,
                                                       const float* __restrict__ h, int nt,
                                                       float* __restrict__ y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = xx + i + (nt - 1);
  float acc = 0.f;
  for (int k = 0; k < nt; ++k) acc = fmaf(h[k], p[-k], acc);
  y[i] = acc;
}

// work[f][i] = x[f*hop + i]*win[i] (i < chunk), 0 (chunk <= i < nfft)
__global__ __launch_bounds__(256) void psd_pre_kernel(const float2* __restrict__ xc,
                                                      const float* __restrict__ xr, size_t hop,
                                                      int chunk, int nfft,
                                                      const float* __restrict__ win,
                                                      float2* __restrict__ work) {
  const int f = blockIdx.y;
  float2* w = work + (size_t)f * nfft;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nfft; i += gridDim.x * blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    if (i < chunk) {
      const float g = win[i];
      if (xc) {
        const float2 s = xc[(size_t)f * hop + i];
        v = make_float2(s.x * g, s.y * g);
      } else {
        v.x = xr[(size_t)f * hop + i] * g;
      }
    }
    w[i] = v;
  }
}

// out[f][j] = 10*log10(|X[(j + nfft/2) mod nfft]|^2 + floor)   (complex input, nout = nfft)
// out[f][j] = 10*log10(|X[j]|^2 + floor), j < nfft/2            (real input,    nout = nfft/2)
__global__ __launch_bounds__(256) void psd_post_kernel(const float2* __restrict__ work, int nfft,
                                                       int half, int db, float* __restrict__ out) {
  const int f = blockIdx.y;
  const int nout = half ? nfft / 2 : nfft;
  const float2* w = work + (size_t)f * nfft;
  float* o = out + (size_t)f * nout;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nout; j += gridDim.x * blockDim.x) {
    // np.fft.fftshift = roll by nfft/2: out[j] = X[(j + ceil(nfft/2)) mod nfft]; nfft need not be a
    // power of two (the GUI's clamp passes 65636, Plotting.py:374-375)
    int k = j;
    if (!half) { k = j + (nfft + 1) / 2; if (k >= nfft) k -= nfft; }
    const float2 v = w[k];
    float p = v.x * v.x + v.y * v.y;
    if (db) p = 10.f * log10f(p + 1.0e-30f);
    o[j] = p;
  }
}

}  // namespace

int launch_quad_mixer(const float2* x, float2* y, size_t n, uint32_t phase0, uint32_t fword,
                      hipStream_t st) {
  if (n == 0) return PYSDR_OK;
  size_t blocks = ((n + 1) / 2 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(quad_mixer_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, n, phase0,
                     fword);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_fir_real(const float* xx, const float* h, int nt, float* y, int n, hipStream_t st) {
  if (n <= 0) return PYSDR_OK;
  hipLaunchKernelGGL(fir_real_kernel, dim3((n + 255) / 256), dim3(256), 0, st, xx, h, nt, y, n);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_psd_pre(const float2* x, size_t hop, int nframes, int chunk, int nfft, const float* win,
                   float2* work, int is_complex, hipStream_t st) {
  int bx = (nfft + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(psd_pre_kernel, dim3(bx, nframes), dim3(256), 0, st,
                     is_complex ? x : nullptr,
                     is_complex ? nullptr : reinterpret_cast<const float*>(x), hop, chunk, nfft,
                     win, work);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_psd_post(const float2* work, int nframes, int nfft, int half, int db, float* out,
                    hipStream_t st) {
  int bx = (nfft + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(psd_post_kernel, dim3(bx, nframes), dim3(256), 0, st, work, nfft, half, db,
                     out);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
