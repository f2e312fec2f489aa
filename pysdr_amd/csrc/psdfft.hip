// Fused PSD frame path for the RF waterfall (spectrum.periodogram, Plotting.py:462; sizes
// gui.py:611-616 after the 2^16 clamp of Plotting.py:370-375): window -> zero-pad
// 32768 -> 65536 -> FFT -> re^2+im^2 -> 10*log10 -> fftshift (formula pinned by
// rtty.py:839-841).  rocFFT needs a pad kernel, two transform kernels and a dB kernel
// (~3.5 MB of traffic per frame); here the 64k transform is a four-step 256 x 256
// decomposition in TWO kernels with the window/zero-pad fused into the first and the
// power/dB/fftshift into the second:
//
//   n = 256 a + b (a < 128 non-zero rows),  k = p + 256 q
//   psd_cols:  Y[p][b] = W_65536^(b p) * sum_a xw[256 a + b] W_256^(a p)     (256-pt over a)
//   psd_rows:  X[p + 256 q] = sum_b Y[p][b] W_256^(b q)                      (256-pt over b)
//
// HBM traffic per frame = 256 KB in + 256 KB out; the 512 KB intermediate Y of a group of
// frames is written and re-read immediately, so it lives in L2 / Infinity Cache.  Y is stored
// as [b / 16][p][b % 16]: a columns workgroup (16 columns) writes one contiguous 32 KB block, and
// every wave-level store / load of the intermediate covers 512 contiguous bytes (the natural
// [p][b] order makes 128-byte pieces 2 KB apart: 257 -> 251 ns per frame).  Each
// 256-point transform is radix 16 x 16 (one 16-point butterfly per thread per pass), data
// exchanged through LDS with conflict-free strides.
#include "common.h"

namespace pysdr {

namespace {

constexpr int kN = 65536, kM = 32768;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// exp(-j*pi*t): v_sin_f32 / v_cos_f32 take revolutions (1.2e-7 abs error on gfx950)
__device__ __forceinline__ float2 expmpi(float t) {
  const float rev = 0.5f * t;
  return make_float2(__builtin_amdgcn_cosf(rev), -__builtin_amdgcn_sinf(rev));
}

// Explicit global-address-space accesses: the unit bodies below are also compiled inside
// non-kernel functions (scripts/experiments), where a generic pointer would turn into
// flat_load / flat_store.
typedef float v2f_t __attribute__((ext_vector_type(2)));
#define PYSDR_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ float2 ldg2(const float2* p) {
  const v2f_t v = *(const PYSDR_AS1 v2f_t*)p;
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void stg2(float2* p, float2 v) {
  const v2f_t w = {v.x, v.y};
  *(PYSDR_AS1 v2f_t*)p = w;
}
// streaming accesses (input read once, PSD written once): non-temporal, so that they do not
// displace the four-step intermediate from the Infinity Cache (measured: 3.15 -> 3.02 ms per
// 10666 frames; the same hint on the intermediate itself makes things worse, 3.37 ms)
__device__ __forceinline__ float2 ldg2_stream(const float2* p) {
  const v2f_t v = __builtin_nontemporal_load((const PYSDR_AS1 v2f_t*)p);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ void stg1_stream(float* p, float v) {
  __builtin_nontemporal_store(v, (PYSDR_AS1 float*)p);
}
__device__ __forceinline__ float ldg1(const float* p) {
  return *(const PYSDR_AS1 float*)p;
}
__device__ __forceinline__ void stg1(float* p, float v) {
  *(PYSDR_AS1 float*)p = v;
}

// 4-point DFT in place (W4 = -j): (a,b,c,d) -> (X0,X1,X2,X3)
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
  const float2 s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
  a = cadd(s0, s2);
  c = csub(s0, s2);
  b = make_float2(s1.x + s3.y, s1.y - s3.x);   // s1 - j*s3
  d = make_float2(s1.x - s3.y, s1.y + s3.x);   // s1 + j*s3
}

// 16-point DFT in place, natural order in and out.  n = n0 + 4 n1, k = k1 + 4 k0:
// DFT4 over n1, twiddle W16^(n0 k1), DFT4 over n0.
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;   // cos, sin(pi/8)
  const float h = 0.70710678118654752f;
#pragma unroll
  for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
  // now v[n0 + 4*k1] = A[n0][k1]; twiddles W16^(n0*k1)
  v[5] = cmul(v[5], make_float2(c1, -s1));     // n0=1,k1=1 : W^1
  v[9] = cmul(v[9], make_float2(h, -h));       // n0=1,k1=2 : W^2
  v[13] = cmul(v[13], make_float2(s1, -c1));   // n0=1,k1=3 : W^3
  v[6] = cmul(v[6], make_float2(h, -h));       // n0=2,k1=1 : W^2
  v[10] = make_float2(v[10].y, -v[10].x);      // n0=2,k1=2 : W^4 = -j
  v[14] = cmul(v[14], make_float2(-h, -h));    // n0=2,k1=3 : W^6
  v[7] = cmul(v[7], make_float2(s1, -c1));     // n0=3,k1=1 : W^3
  v[11] = cmul(v[11], make_float2(-h, -h));    // n0=3,k1=2 : W^6
  v[15] = cmul(v[15], make_float2(-c1, s1));   // n0=3,k1=3 : W^9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
  // now v[4*k1 + k0] = X[k1 + 4*k0]: transpose the 4x4 index to natural order
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) {
      const float2 t = v[4 * i + j];
      v[4 * i + j] = v[4 * j + i];
      v[4 * j + i] = t;
    }
}

// v[k] *= w0 * w^k, k = 0..15 (powers built with depth <= 4 multiplications)
__device__ __forceinline__ void twiddle_pow0(float2 (&v)[16], float2 w, float2 w0) {
  const float2 w2 = cmul(w, w), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
  const float2 b0 = w0, b4 = cmul(w0, w4), b8 = cmul(w0, w8), b12 = cmul(b4, w8);
  const float2 w3 = cmul(w2, w);
  v[0] = cmul(v[0], b0); v[1] = cmul(v[1], cmul(b0, w)); v[2] = cmul(v[2], cmul(b0, w2)); v[3] = cmul(v[3], cmul(b0, w3));
  v[4] = cmul(v[4], b4); v[5] = cmul(v[5], cmul(b4, w)); v[6] = cmul(v[6], cmul(b4, w2)); v[7] = cmul(v[7], cmul(b4, w3));
  v[8] = cmul(v[8], b8); v[9] = cmul(v[9], cmul(b8, w)); v[10] = cmul(v[10], cmul(b8, w2)); v[11] = cmul(v[11], cmul(b8, w3));
  v[12] = cmul(v[12], b12); v[13] = cmul(v[13], cmul(b12, w)); v[14] = cmul(v[14], cmul(b12, w2)); v[15] = cmul(v[15], cmul(b12, w3));
}
// W_256^(i j), i, j < 16, as an LDS table of 16 rows of 17 float2; built by the first 256 threads of the workgroup,
// one v_sin/v_cos pair each.  Building the 15 powers per thread instead costs 14 complex
// multiplications (56 of the ~350 VALU instructions of a pass) and their rounding: the table is
// 2 % faster and three times closer to the exact transform (4.4e-7 vs 1.3e-6 of the peak).
// Row pitch: hipcc reads a row as ds_read2_b64 pairs, which the LDS serves 16 lanes at a time on 32 banks of 4 bytes.
// In psd_cols the 16 lanes of a group read the SAME row (broadcast); in psd_rows they read 16 DIFFERENT rows at the
// same column, row i at dword 2 * pitch * i: with the round-1 pitch of 18 that is 4 i mod 32 -- rows i and i + 8 on
// the same banks, the 795 k conflict cycles of 3.47 M LDS cycles in profiles/r03_c3_pmc_sq.json (23 %; the comments
// here claimed "conflict free") -- with 17 it is 2 i mod 32, two dwords per lane: all 32 banks once.
constexpr int kTwRow = 17;
__device__ __forceinline__ void tw256_build(float2* tw, int tid) {
  if (tid < 256) tw[kTwRow * (tid >> 4) + (tid & 15)] = expmpi((float)((tid >> 4) * (tid & 15)) * (1.0f / 128.0f));
}
// v[k] *= W_256^(i k)
__device__ __forceinline__ void twiddle_tab(float2 (&v)[16], const float2* tw, int i) {
  const float2* r = tw + kTwRow * i;
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], r[k]);
}

// ---- step 1: 16 columns of one frame per workgroup (grid = 16 x frames, 256 threads).
//   a = a0 + 16 a1 (a1 < 8: the rest is zero padding), p = p1 + 16 p0
//   pass 1 thread (b, a0): DFT16 over a1, * W_256^(a0 p1)      -> LDS[p1][a0][b]
//   pass 2 thread (b, p1): DFT16 over a0, * W_65536^(bb p)      -> Y[p][bb], bb = 16 cb + b
// LDS slot(p1, a0, b) = 272 p1 + 16 a0 + b: consecutive lanes write consecutive slots in
// pass 1, and the 16-slot pad per p1 spreads pass 2's two p1 per half-wave over all banks.
constexpr int kColsPerWg = 16;
constexpr int kColLds = 16 * 272;

// one unit = 16 columns [16 cb, 16 cb + 16) of frame xf, 256 threads (tid), LDS kColLds
__device__ __forceinline__ void cols_unit(const float2* __restrict__ xf, const float* __restrict__ win,
                                          float2* __restrict__ yf, int cb, int tid, float2* lds,
                                          const float2* tw) {
  const int b = tid & 15, hi = tid >> 4;
  const int bb = cb * kColsPerWg + b;
  {
    const int a0 = hi;
    float2 u[16];
#pragma unroll
    for (int a1 = 0; a1 < 8; ++a1) {
      const int n = 256 * (a0 + 16 * a1) + bb;
      const float2 s = ldg2_stream(xf + n);
      const float g = ldg1(win + n);
      u[a1] = make_float2(s.x * g, s.y * g);
    }
#pragma unroll
    for (int a1 = 8; a1 < 16; ++a1) u[a1] = make_float2(0.f, 0.f);
    dft16(u);
    __syncthreads();                                                   // the table (the loads above were in flight)
    twiddle_tab(u, tw, a0);                                            // W_256^(a0 p1)
    float2* p = lds + 16 * a0 + b;
#pragma unroll
    for (int p1 = 0; p1 < 16; ++p1) p[272 * p1] = u[p1];
  }
  __syncthreads();
  {
    const int p1 = hi;
    const float2* p = lds + 272 * p1 + b;
    float2 v[16];
#pragma unroll
    for (int a0 = 0; a0 < 16; ++a0) v[a0] = p[16 * a0];
    dft16(v);
    // four-step twiddle W_65536^(bb * (p1 + 16 p0)) = W^(bb p1) * (W^(16 bb))^p0
    twiddle_pow0(v, expmpi((float)(16 * bb) * (1.0f / 32768.0f)),
                 expmpi((float)(bb * p1) * (1.0f / 32768.0f)));
    // intermediate layout Y[cb][p][b]: the workgroup's 16 columns x 256 rows are one contiguous
    // 32 KB block, and a wave's store covers 512 contiguous bytes (4 p1 x 16 b)
    float2* o = yf + (size_t)cb * 4096 + p1 * 16 + b;
#pragma unroll
    for (int p0 = 0; p0 < 16; ++p0) stg2(o + p0 * 256, v[p0]);
  }
}

__global__ __launch_bounds__(256) void psd_cols_kernel(const float2* __restrict__ x, size_t hop,
                                                       const float* __restrict__ win,
                                                       float2* __restrict__ work) {
  __shared__ __attribute__((aligned(16))) float2 lds[kColLds];
  __shared__ __attribute__((aligned(16))) float2 tw[16 * kTwRow];
  const int f = blockIdx.y;
  tw256_build(tw, threadIdx.x);
  cols_unit(x + (size_t)f * hop, win, work + (size_t)f * kN, blockIdx.x, threadIdx.x, lds, tw);
}

// ---- step 2: 32 rows p of one frame per workgroup (grid = 8 x frames, 512 threads).
//   b = c0 + 16 c1, q = q1 + 16 q0
//   pass 1 thread (c0, pl): DFT16 over c1, * W_256^(c0 q1)     -> LDS[q1][pl][c0]
//   pass 2 thread (pl, q1): DFT16 over c0 -> X[p + 256 (q1 + 16 q0)] -> dB -> fftshift
// LDS slot(q1, pl, c0) = 544 q1 + 17 pl + c0: pass 1's 16-lane groups write 16 consecutive
// slots; pass 2's lanes (pl) are 17 slots apart = 34 dwords = 2 pl mod 32 for the 16 lanes a ds_read2_b64 access
// serves at a time, so the data passes are conflict free (the conflicts the counters showed came from the twiddle
// table, see kTwRow).  32 consecutive p per store = one full 128-B line.
constexpr int kRowsPerWg = 32;
constexpr int kRowLds = 16 * 544;

// how the rows pass reads the intermediate: plain loads (two-kernel path: the kernel boundary
// made it visible) or agent-scope loads that miss the CU's L1 (fused path: written by other
// CUs of the same XCD during this kernel, served by their common L2)
template <bool kBypassL1>
__device__ __forceinline__ float2 load_work(const float2* p) {
  if (kBypassL1) {
    const unsigned long long v = __hip_atomic_load(
        (const PYSDR_AS1 unsigned long long*)p,
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)(v & 0xffffffffull)), __uint_as_float((unsigned)(v >> 32)));
  }
  return ldg2(p);
}

// one unit = 32 rows [32 rb, 32 rb + 32) of frame yf, 512 threads, LDS kRowLds
template <bool kBypassL1>
__device__ __forceinline__ void rows_unit(const float2* yf, float* __restrict__ of, int rb, int db,
                                          int tid, float2* lds, const float2* tw) {
  {
    const int c0 = tid & 15, pl = tid >> 4;
    // Y[c1][p][c0]: a wave's load covers 512 contiguous bytes (4 rows x 16 columns)
    const float2* src = yf + (size_t)(rb * kRowsPerWg + pl) * 16 + c0;
    float2 u[16];
#pragma unroll
    for (int c1 = 0; c1 < 16; ++c1) u[c1] = load_work<kBypassL1>(src + 4096 * c1);
    dft16(u);
    __syncthreads();
    twiddle_tab(u, tw, c0);                                            // W_256^(c0 q1)
    float2* p = lds + 17 * pl + c0;
#pragma unroll
    for (int q1 = 0; q1 < 16; ++q1) p[544 * q1] = u[q1];
  }
  __syncthreads();
  {
    const int pl = tid & 31, q1 = tid >> 5;
    const float2* p = lds + 544 * q1 + 17 * pl;
    float2 v[16];
#pragma unroll
    for (int c0 = 0; c0 < 16; ++c0) v[c0] = p[c0];
    dft16(v);
    const int kb = rb * kRowsPerWg + pl + 256 * q1;
#pragma unroll
    for (int q0 = 0; q0 < 16; ++q0) {
      const int k = kb + 4096 * q0;
      float pw = v[q0].x * v[q0].x + v[q0].y * v[q0].y;
      // 10*log10(pw) = 10*log10(2) * log2(pw); pw + 1e-30 is a normal number, where v_log_f32 is
      // good to 1 ulp (log10f's expansion handles denormals and costs ~9 instructions per bin)
      if (db) pw = 3.0102999566398120f * __builtin_amdgcn_logf(pw + 1.0e-30f);
      stg1_stream(of + ((k + kM) & (kN - 1)), pw);
    }
  }
}

__global__ __launch_bounds__(512) void psd_rows_kernel(const float2* __restrict__ work,
                                                       float* __restrict__ out, int db) {
  __shared__ __attribute__((aligned(16))) float2 lds[kRowLds];
  __shared__ __attribute__((aligned(16))) float2 tw[16 * kTwRow];
  const int f = blockIdx.y;
  tw256_build(tw, threadIdx.x);
  rows_unit<false>(work + (size_t)f * kN, out + (size_t)f * kN, blockIdx.x, db, threadIdx.x, lds, tw);
}


// ---- the same pair with a 24-BIT intermediate (round 4).  The pair is bound by the bytes it moves across the XCD <->
// memory fabric (section 4.3 of DESIGN.md: 1.5 MB per frame, of which 1.0 MB is the intermediate written and read back),
// not by arithmetic, so the intermediate is stored as block-scaled 24-bit fixed point: 6 instead of 8 bytes per complex,
// 1.25 MB per frame.  Block = the 16 columns x 256 rows one columns workgroup produces; its scale maps the block's largest
// |component| to 8388600, so a stored value is off by at most 6e-8 of the block maximum -- the rows transform adds 256 of
// them incoherently (1e-6 of the block maximum) under a line that is at least ~16x the block's values: two orders inside the
// 5e-6 amplitude bar of psd_check.  The two planes keep every access a power of two wide: HI = the upper 16 bits of Re and Im
// in one dword per complex, LO = the low byte of each in one 16-bit word; same [cb][p][b] index order as the float2 form,
// so a wave's store / load covers 256 + 128 contiguous bytes.  v_perm_b32 packs and unpacks (6 vector instructions per
// complex on each side: +16 % on kernels that were never bound by them).
//   per frame (512 KB of the work buffer): [0, 256 K) HI dwords, [256 K, 384 K) LO shorts, [384 K, +256) the 16 x 4 block scales
constexpr int kPkLoOff = 65536 * 4, kPkScaleOff = 65536 * 6;

__device__ __forceinline__ unsigned pk_perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }

__device__ __forceinline__ void cols_unit_pk(const float2* __restrict__ xf, const float* __restrict__ win,
                                             char* __restrict__ wf, int cb, int tid, float2* lds, const float2* tw,
                                             float* red) {
  const int b = tid & 15, hi = tid >> 4;
  const int bb = cb * kColsPerWg + b;
  {
    const int a0 = hi;
    float2 u[16];
#pragma unroll
    for (int a1 = 0; a1 < 8; ++a1) {
      const int n = 256 * (a0 + 16 * a1) + bb;
      const float2 s = ldg2_stream(xf + n);
      const float g = ldg1(win + n);
      u[a1] = make_float2(s.x * g, s.y * g);
    }
#pragma unroll
    for (int a1 = 8; a1 < 16; ++a1) u[a1] = make_float2(0.f, 0.f);
    dft16(u);
    __syncthreads();
    twiddle_tab(u, tw, a0);
    float2* p = lds + 16 * a0 + b;
#pragma unroll
    for (int p1 = 0; p1 < 16; ++p1) p[272 * p1] = u[p1];
  }
  __syncthreads();
  {
    const int p1 = hi;
    const float2* p = lds + 272 * p1 + b;
    float2 v[16];
#pragma unroll
    for (int a0 = 0; a0 < 16; ++a0) v[a0] = p[16 * a0];
    dft16(v);
    twiddle_pow0(v, expmpi((float)(16 * bb) * (1.0f / 32768.0f)),
                 expmpi((float)(bb * p1) * (1.0f / 32768.0f)));
    // the block's largest |component|: thread, wave (DPP-free butterflies through shuffles), workgroup.
    // Measured (scripts/diag/psd_variants.sh, PSD alone, ms per 10666 frames): shipped 2.27; a fixed scale (ablation PSDX_NO_SCALE,
    // results wrong; scripts/experiments/ablation_switches.patch.txt) 2.18; a scale per WAVE (-DPSDX_WAVE_SCALE: no barrier, no LDS round trip in front of the stores; parity
    // green) 2.35-2.41 against 2.35-2.37 on the same box: the barrier is not what the reduction costs.  Without the LO plane
    // (ablation PSDX_NO_LO, results wrong) 2.00, without both 1.85: the pair moves its bytes at ~6 TB/s in every variant.
    float m = 0.f;
#pragma unroll
    for (int p0 = 0; p0 < 16; ++p0) m = fmaxf(fmaxf(m, fabsf(v[p0].x)), fabsf(v[p0].y));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
#ifndef PSDX_WAVE_SCALE
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
#endif
    // (a block whose largest component is below 1e-20 contributes powers below 1e-36 * 65536: far under the 1e-30 that is
    //  added before the logarithm -- it is stored as zeros, which also keeps 8388600 / m finite for denormal m)
    const bool live = m >= 1.0e-20f;
    const float sc = live ? __fdiv_rn(8388600.0f, m) : 0.0f;
    // four scale slots per block, one per wave (the same value four times unless PSDX_WAVE_SCALE)
    if ((tid & 63) == 0)
      *(PYSDR_AS1 float*)(wf + kPkScaleOff + 4 * (4 * cb + (tid >> 6))) = live ? __fdiv_rn(m, 8388600.0f) * (1.0f / 256.0f) : 0.f;
    PYSDR_AS1 unsigned* oh = (PYSDR_AS1 unsigned*)wf + (size_t)cb * 4096 + p1 * 16 + b;
    PYSDR_AS1 unsigned short* ol = (PYSDR_AS1 unsigned short*)(wf + kPkLoOff) + (size_t)cb * 4096 + p1 * 16 + b;
#pragma unroll
    for (int p0 = 0; p0 < 16; ++p0) {
      const unsigned a = (unsigned)__float2int_rn(v[p0].x * sc), c = (unsigned)__float2int_rn(v[p0].y * sc);
      oh[p0 * 256] = pk_perm(c, a, 0x06050201u);                       // [a.b1, a.b2, c.b1, c.b2]
      ol[p0 * 256] = (unsigned short)pk_perm(c, a, 0x0c0c0400u);       // [a.b0, c.b0]
    }
  }
}

__global__ __launch_bounds__(256) void psd_cols_pk_kernel(const float2* __restrict__ x, size_t hop,
                                                          const float* __restrict__ win, char* __restrict__ work) {
  __shared__ __attribute__((aligned(16))) float2 lds[kColLds];
  __shared__ __attribute__((aligned(16))) float2 tw[16 * kTwRow];
  __shared__ float red[4];
  const int f = blockIdx.y;
  tw256_build(tw, threadIdx.x);
  cols_unit_pk(x + (size_t)f * hop, win, work + (size_t)f * kN * sizeof(float2), blockIdx.x, threadIdx.x, lds, tw, red);
}

__device__ __forceinline__ void rows_unit_pk(const char* __restrict__ wf, float* __restrict__ of, int rb, int db,
                                             int tid, float2* lds, const float2* tw) {
  {
    const int c0 = tid & 15, pl = tid >> 4;
    const size_t idx = (size_t)(rb * kRowsPerWg + pl) * 16 + c0;
    const PYSDR_AS1 unsigned* sh = (const PYSDR_AS1 unsigned*)wf + idx;
    const PYSDR_AS1 unsigned short* sl = (const PYSDR_AS1 unsigned short*)(wf + kPkLoOff) + idx;
    const PYSDR_AS1 float* ss = (const PYSDR_AS1 float*)(wf + kPkScaleOff);
    unsigned h[16], l[16];
#pragma unroll
    for (int c1 = 0; c1 < 16; ++c1) { h[c1] = sh[4096 * c1]; l[c1] = sl[4096 * c1]; }
    float2 u[16];
#pragma unroll
    for (int c1 = 0; c1 < 16; ++c1) {
      const float inv = ss[4 * c1 + ((tid >> 6) & 3)];                   // block scale / 256 (wave-uniform: the wave's 4 rows p share p % 16 >> 2)
      const int xr = (int)pk_perm(l[c1], h[c1], 0x0100040cu);            // [0, l.b0, h.b0, h.b1] = Re * 256
      const int xi = (int)pk_perm(l[c1], h[c1], 0x0302050cu);            // [0, l.b1, h.b2, h.b3] = Im * 256
      u[c1] = make_float2((float)xr * inv, (float)xi * inv);
    }
    dft16(u);
    __syncthreads();
    twiddle_tab(u, tw, c0);
    float2* p = lds + 17 * pl + c0;
#pragma unroll
    for (int q1 = 0; q1 < 16; ++q1) p[544 * q1] = u[q1];
  }
  __syncthreads();
  {
    const int pl = tid & 31, q1 = tid >> 5;
    const float2* p = lds + 544 * q1 + 17 * pl;
    float2 v[16];
#pragma unroll
    for (int c0 = 0; c0 < 16; ++c0) v[c0] = p[c0];
    dft16(v);
    const int kb = rb * kRowsPerWg + pl + 256 * q1;
#pragma unroll
    for (int q0 = 0; q0 < 16; ++q0) {
      const int k = kb + 4096 * q0;
      float pw = v[q0].x * v[q0].x + v[q0].y * v[q0].y;
      if (db) pw = 3.0102999566398120f * __builtin_amdgcn_logf(pw + 1.0e-30f);
      stg1_stream(of + ((k + kM) & (kN - 1)), pw);
    }
  }
}

__global__ __launch_bounds__(512) void psd_rows_pk_kernel(const char* __restrict__ work, float* __restrict__ out, int db) {
  __shared__ __attribute__((aligned(16))) float2 lds[kRowLds];
  __shared__ __attribute__((aligned(16))) float2 tw[16 * kTwRow];
  const int f = blockIdx.y;
  tw256_build(tw, threadIdx.x);
  rows_unit_pk(work + (size_t)f * kN * sizeof(float2), out + (size_t)f * kN, blockIdx.x, db, threadIdx.x, lds, tw);
}

}  // namespace

// nframes frames; `work` holds nframes x 65536 complex of intermediate.
int launch_psd64k(const float2* x, size_t hop, int nframes, const float* win, float2* work,
                  float* out, int db, hipStream_t st, int packed) {
  if (packed) {
    hipLaunchKernelGGL(psd_cols_pk_kernel, dim3(256 / kColsPerWg, nframes), dim3(256), 0, st, x, hop, win, (char*)work);
    PYSDR_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(psd_rows_pk_kernel, dim3(256 / kRowsPerWg, nframes), dim3(512), 0, st, (const char*)work, out, db);
    PYSDR_HIP_CHECK(hipGetLastError());
    return PYSDR_OK;
  }
  hipLaunchKernelGGL(psd_cols_kernel, dim3(256 / kColsPerWg, nframes), dim3(256), 0, st, x, hop, win,
                     work);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(psd_rows_kernel, dim3(256 / kRowsPerWg, nframes), dim3(512), 0, st, work, out, db);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
