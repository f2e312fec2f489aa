// EXPERIMENT (not part of libpysdr_hip.so): the 64k PSD of a zero-padded 32768-sample frame in ONE
// 1024-thread workgroup per CU, so that the four-step intermediate never crosses the fabric
// (index algebra checked in psd_one_model.py).
//   bins 4k + r (r = 0..3) = 16384-point DFT of  y_r[n] = W_N^(r n) (xw[n] + (-j)^r xw[n + 16384])
//   the windowed frame stays in 64 registers per thread (n = t + 1024 i), each y_r goes through
//   pass A (registers: DFT16 over i, twiddle) -> LDS -> pass B (DFT16 over t/64, twiddle) -> LDS ->
//   pass C (DFT16 over (t%64)/4, twiddle, DFT4 across the 4 lanes of a quad by DPP) -> |.|^2 -> dB ->
//   a per-workgroup scratch plane [kc][kb][ka][kd] (L2 / Infinity Cache), and after the four planes a
//   gather writes the interleaved, fft-shifted bins as contiguous 16-byte stores.
// RESULT (round 3): does not fit.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
//   -Rpass-analysis=kernel-resource-usage -c scripts/experiments/psd_one.hip  ->  VGPRs 128, 169 registers SPILLED
//   (652 bytes of scratch per lane): 1024 threads leave 128 registers, the frame pins 64 of them across the four
//   passes and a radix-16 pass wants 59-76 by itself.  The variants that do fit either re-read the frame for
//   three of the four passes or run radix-8 passes with twice the LDS exchanges; both move ~1.0 MB per frame
//   (fabric floor 150 ns) and are bound by vector issue near 200 ns.  Kept as the record of the
//   design; never run.
#include "../../pysdr_amd/csrc/psdfft.hip"

namespace pysdr {
namespace {

constexpr int kQ = 16384;
constexpr int kRS = 1092;                 // row stride (float2) of the [ka][t + 4 (t / 64)] image: 1088 + 4 spreads ka over the banks
constexpr int kOneLds = 16 * kRS;

#define PSD1_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))

__global__ __launch_bounds__(1024) void psd_one_kernel(const float2* __restrict__ x, size_t hop, int nframes,
                                                       const float* __restrict__ win, float* __restrict__ scratch,
                                                       float* __restrict__ out, int db) {
  extern __shared__ __attribute__((aligned(16))) float2 lds1[];
  float2* const tw = lds1 + kOneLds;
  const int t = threadIdx.x;
  tw256_build(tw, t);
  float* const sc = scratch + (size_t)blockIdx.x * 65536;
  const int lane = t & 63;
  // per-lane constants of the quad DFT4 (lane u of a quad ends up with bin kd = {0, 2, 1, 3}[u])
  const int u = lane & 3;
  const float g1 = (u & 2) ? -1.f : 1.f;                         // stage 1: partner + g1 * own
  const float k1 = (u == 0) ? 1.f : (u == 1) ? -1.f : 0.f;       // stage 2: re = A.x + k1 B.x + k2 B.y
  const float k2 = (u == 2) ? 1.f : (u == 3) ? -1.f : 0.f;       //          im = A.y - k2 B.x + k1 B.y
  const int kd = (u == 1) ? 2 : (u == 2) ? 1 : u;

  for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
    const float2* xf = x + (size_t)f * hop;
    float2 x0[16], x1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = t + 1024 * i;
      const float2 a = ldg2_stream(xf + n), b = ldg2_stream(xf + n + kQ);
      const float ga = ldg1(win + n), gb = ldg1(win + n + kQ);
      x0[i] = make_float2(a.x * ga, a.y * ga);
      x1[i] = make_float2(b.x * gb, b.y * gb);
    }
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      // ---- pass A: y_r[t + 1024 i] in registers, DFT16 over i, * W_Q^(t ka)
      {
        float2 v[16];
        const float sg = (r & 2) ? -1.f : 1.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float2 b = x1[i];
          const float2 rb = (r & 1) ? make_float2(b.y, -b.x) : b;       // (-j)^r x1 = sg * rb
          v[i] = make_float2(fmaf(sg, rb.x, x0[i].x), fmaf(sg, rb.y, x0[i].y));
        }
        if (r) twiddle_pow0(v, expmpi((float)r * (1.0f / 32.0f)), expmpi((float)(r * t) * (1.0f / 32768.0f)));
        dft16(v);
        twiddle_pow0(v, expmpi((float)t * (1.0f / 8192.0f)), make_float2(1.f, 0.f));
        float2* p = lds1 + t + 4 * (t >> 6);
#pragma unroll
        for (int ka = 0; ka < 16; ++ka) p[kRS * ka] = v[ka];
      }
      __syncthreads();
      // ---- pass B: thread (ka, tl): DFT16 over th (t = tl + 64 th), * W_1024^(tl kb), in place
      {
        const int ka = t >> 6, tl = t & 63;
        float2* p = lds1 + kRS * ka + tl;
        float2 v[16];
#pragma unroll
        for (int th = 0; th < 16; ++th) v[th] = p[68 * th];
        dft16(v);
        twiddle_pow0(v, expmpi((float)tl * (1.0f / 512.0f)), make_float2(1.f, 0.f));
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) p[68 * kb] = v[kb];
      }
      __syncthreads();
      // ---- pass C: thread (kb; ka, u): DFT16 over s (tl = u + 4 s), * W_64^(u kc), DFT4 over u across the quad
      {
        const int kb = t >> 6, ka = lane >> 2;
        const float2* p = lds1 + kRS * ka + 68 * kb + u;
        float2 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = p[4 * s];
        dft16(v);
        twiddle_tab(v, tw, 4 * u);
        float* o = sc + (size_t)r * kQ + 64 * kb + 4 * ka + kd;
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
          // stage 1: partner = lane u ^ 2
          const float px = PSD1_DPP(v[kc].x, 0x4E), py = PSD1_DPP(v[kc].y, 0x4E);          // quad_perm [2,3,0,1]
          const float sx = fmaf(g1, v[kc].x, px), sy = fmaf(g1, v[kc].y, py);
          // stage 2: partner = lane u ^ 1
          const float qx = PSD1_DPP(sx, 0xB1), qy = PSD1_DPP(sy, 0xB1);                    // quad_perm [1,0,3,2]
          const bool odd = (u & 1) != 0;
          const float ax = odd ? qx : sx, ay = odd ? qy : sy, bx = odd ? sx : qx, by = odd ? sy : qy;
          const float re = fmaf(k2, by, fmaf(k1, bx, ax));
          const float im = fmaf(k1, by, fmaf(-k2, bx, ay));
          float pw = re * re + im * im;
          if (db) pw = 3.0102999566398120f * __builtin_amdgcn_logf(pw + 1.0e-30f);
          stg1(o + 1024 * kc, pw);
        }
      }
      __syncthreads();
    }
    // ---- gather: plane[r] float4 #m holds X_r[m + 4096 kd], kd = 0..3; bins 4 k + r, fft-shifted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* of = out + (size_t)f * kN;
#pragma unroll 4
    for (int j = 0; j < 4; ++j) {
      const int m = t + 1024 * j;
      typedef float v4f_t __attribute__((ext_vector_type(4)));
      v4f_t pl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pl[r] = __builtin_nontemporal_load((const PYSDR_AS1 v4f_t*)(sc + (size_t)r * kQ) + m);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f_t w = {pl[0][q], pl[1][q], pl[2][q], pl[3][q]};
        const int k4 = (m + 4096 * q + 8192) & (kQ - 1);                 // float4 index of bins 4k .. 4k+3 after the fftshift
        __builtin_nontemporal_store(w, (PYSDR_AS1 v4f_t*)of + k4);
      }
    }
    __syncthreads();           // the scratch planes are rewritten by the next frame
  }
}

}  // namespace

// scratch: 256 KB per workgroup
int launch_psd64k_one(const float2* x, size_t hop, int nframes, const float* win, float* scratch, int nwg, float* out, int db,
                      hipStream_t st) {
  static bool attr = false;
  const size_t lds = (size_t)(kOneLds + 16 * kTwRow) * sizeof(float2);
  if (!attr) {
    PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_one_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(psd_one_kernel, dim3(nwg < nframes ? nwg : nframes), dim3(1024), lds, st, x, hop, nframes, win, scratch, out, db);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
