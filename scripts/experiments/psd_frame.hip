// Register-resident 64k PSD frame path (spectrum.periodogram at the RF-waterfall size, Plotting.py:462;
// sizes gui.py:611-616 after the 2^16 clamp of Plotting.py:370-375): window -> zero-pad 32768 -> 65536
// -> FFT -> re^2+im^2 -> 10*log10 -> fftshift (formula pinned by rtty.py:839-841).
//
// The zero-padded 2N-point transform of an N-point frame (N = 32768) is two N-point transforms:
//     X[2k]   = FFT_N( x w )[k]                       (even bins)
//     X[2k+1] = FFT_N( x w exp(-j pi n / N) )[k]      (odd bins)
// ONE 512-thread workgroup per frame (one per CU: 256 registers per thread) does both, one after
// the other, with the whole 256 KB transform living in its REGISTERS (64 complex per thread); nothing
// but the input (twice, the second time out of L2 / the Infinity Cache) and the finished PSD ever
// crosses the fabric -- the two-kernel four-step form (psdfft.hip) moved 1.5 MB per frame, this one
// 0.75 MB at most (0.5 MB algorithmic).
//
//   n = 512 a + t            t = 8 b + c in [0,512)      a, b in [0,64), c in [0,8)
//   k = kA + 64 kB + 4096 kC                             kA, kB in [0,64), kC in [0,8)
//   step 1  thread t            : DFT64 over a -> kA, * W_32768^(t kA)
//   exchange 1 (workgroup, LDS) : (kA | b, c) -> (b | kA, c)      two rounds of 128 KB
//   step 2  thread (c, kA)      : DFT64 over b -> kB, * W_512^(c kB)
//   exchange 2 (wave local)     : (kB | c) -> (c | kB)             8 x 8 blocks among 8 lanes
//   step 3                      : DFT8 over c -> kC;  |.|^2, dB; the even-bin pass keeps its 64 values
//                                 in registers, the odd-bin pass stores (even, odd) pairs
// Lanes = (c, kA_lo): 512-byte coalesced input rows, 64-byte output runs.
#include "common.h"

namespace pysdr {

namespace {

constexpr int kNh = 32768;            // transform length of one parity
// Ablation switches for scripts/experiments/psd_frame_test (never defined in the product build):
// PSD_ABL bit 0 no global loads, 1 no stores, 2 no exchange 1, 3 no exchange 2, 4 no twiddles,
// 5 no 64-point transforms
#ifndef PSD_ABL
#define PSD_ABL 0
#endif
// PSD_STAMP: s_memtime stamps of the phases of every wave of workgroup 0 (experiment builds only)
#ifdef PSD_STAMP
__device__ unsigned long long g_psd_stamps[2 * 16 * 8 * 4];
#define STAMP(i) do { if (stamp_on) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); \
    if ((tid & 63) == 0) g_psd_stamps[((E * 16 + (i)) * 8 + (tid >> 6))] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
constexpr int kRow = 264;             // exchange-1 row: 256 columns + 8 (reader lanes 16 banks apart)
constexpr int kLdsComplex = 64 * kRow;

typedef float v2f_t __attribute__((ext_vector_type(2)));
#define PYSDR_AS1 __attribute__((address_space(1)))

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mulmj(float2 a) { return make_float2(a.y, -a.x); }      // * (-j)
// exp(-2 pi j rev): v_sin_f32 / v_cos_f32 take revolutions (1.2e-7 abs error on gfx950)
__device__ __forceinline__ float2 expm2pi(float rev) {
  return make_float2(__builtin_amdgcn_cosf(rev), -__builtin_amdgcn_sinf(rev));
}

// W_64^m = (cos, -sin)(2 pi m / 64)
__device__ __forceinline__ float2 w64(int m) {
  constexpr float c[64] = {1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f, 0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f};
  constexpr float s[64] = {0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f, 0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f, 1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f};
  return make_float2(c[m & 63], s[m & 63]);
}
__device__ __forceinline__ float2 mul_w64(float2 v, int m) {       // m is a compile-time constant after unrolling
  m &= 63;
  if (m == 0) return v;
  if (m == 16) return mulmj(v);
  if (m == 32) return make_float2(-v.x, -v.y);
  if (m == 48) return make_float2(-v.y, v.x);
  return cmul(v, w64(m));
}

// 8-point DFT in place, natural order in and out (elements v[s*i], i = 0..7)
template <int S>
__device__ __forceinline__ void dft8(float2* v) {
  const float h = 0.70710678118654752f;
  float2 s0 = cadd(v[0], v[4 * S]), d0 = csub(v[0], v[4 * S]);
  float2 s1 = cadd(v[S], v[5 * S]), d1 = csub(v[S], v[5 * S]);
  float2 s2 = cadd(v[2 * S], v[6 * S]), d2 = csub(v[2 * S], v[6 * S]);
  float2 s3 = cadd(v[3 * S], v[7 * S]), d3 = csub(v[3 * S], v[7 * S]);
  d1 = make_float2((d1.x + d1.y) * h, (d1.y - d1.x) * h);          // * W8^1 = (1 - j)/sqrt2
  d2 = mulmj(d2);                                                  // * W8^2
  d3 = make_float2((d3.y - d3.x) * h, -(d3.x + d3.y) * h);         // * W8^3 = (-1 - j)/sqrt2
  // even outputs: DFT4 of s
  {
    const float2 t0 = cadd(s0, s2), t1 = csub(s0, s2), t2 = cadd(s1, s3), t3 = mulmj(csub(s1, s3));
    v[0] = cadd(t0, t2); v[4 * S] = csub(t0, t2); v[2 * S] = cadd(t1, t3); v[6 * S] = csub(t1, t3);
  }
  // odd outputs: DFT4 of d
  {
    const float2 t0 = cadd(d0, d2), t1 = csub(d0, d2), t2 = cadd(d1, d3), t3 = mulmj(csub(d1, d3));
    v[S] = cadd(t0, t2); v[5 * S] = csub(t0, t2); v[3 * S] = cadd(t1, t3); v[7 * S] = csub(t1, t3);
  }
}

// 64-point DFT in place: input v[a], a = 8 a1 + a0; output v[8 k0 + k1] = X[k0 + 8 k1]
__device__ __forceinline__ void dft64(float2 (&v)[64]) {
  // the scheduler is fenced after every 8-point transform: interleaving several of them for ILP
  // costs more registers than the two waves of a SIMD need to keep it busy
#pragma unroll
  for (int a0 = 0; a0 < 8; ++a0) {
    dft8<8>(&v[a0]);                                              // over a1 -> k0, result at [8 k0 + a0]
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k0 = 0; k0 < 8; ++k0) {
    if (k0 > 0) {
#pragma unroll
      for (int a0 = 1; a0 < 8; ++a0) v[8 * k0 + a0] = mul_w64(v[8 * k0 + a0], a0 * k0);
    }
    dft8<1>(&v[8 * k0]);                                          // over a0 -> k1
    __builtin_amdgcn_sched_barrier(0);
  }
}
// register r of a dft64 result holds frequency index fidx(r)
__device__ __forceinline__ constexpr int fidx(int r) { return (r >> 3) + 8 * (r & 7); }

// exchange 1, one parity class of waves: PAR = wave & 1.  Round j moves the elements with
// (kA_hi + b_hi) & 1 == j: every thread gives up 32 registers and takes 32 per round, so no more
// than 64 complex values are ever live in a thread.
template <int PAR>
__device__ __forceinline__ void exchange1(float2 (&v)[64], float2* lds, int lane, int wave) {
  float2 nv[64];
  float2* wp = lds + 64 * (wave >> 1) + lane;                              // + kA * kRow
  const float2* rp = lds + ((lane >> 3) + 8 * wave) * kRow + (lane & 7);   // + 64 (b_hi >> 1) + 8 b_lo
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int kpar = (j + PAR) & 1;                // kA_hi parity written / b_hi parity read this round
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0)
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const int k1 = 2 * kh + kpar;
        wp[(k0 + 8 * k1) * kRow] = v[8 * k0 + k1];
      }
    __syncthreads();
#pragma unroll
    for (int bh = 0; bh < 4; ++bh) {
      const int b_hi = 2 * bh + kpar;
#pragma unroll
      for (int b_lo = 0; b_lo < 8; ++b_lo) nv[b_lo + 8 * b_hi] = rp[64 * bh + 8 * b_lo];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = nv[i];
}

// One parity (E = 0: even bins, kept in p0; E = 1: odd bins, stored next to the even ones) of one
// frame.  Two instantiations back to back instead of a loop: a loop would carry p0 through a phi
// whose (undefined) initial values get spilled at entry.
template <int E>
__device__ __forceinline__ void psd_pass(const float2* __restrict__ xf, const float* __restrict__ win,
                                         const float2* __restrict__ cwin1, float2* __restrict__ of, int db,
                                         float2* lds, int tid, float (&p0)[64], bool stamp_on) {
  constexpr int e = E;
  float2 v[64];
  STAMP(0);
  // Everything a pass computes from the thread index (64 load and 64 store addresses, 126
  // twiddle factors) is invariant across the two passes; hoisted out of the parity loop it would
  // need 300 registers and spill.  An opaque copy of the index per pass keeps it inside.
  int tq = tid;
  asm volatile("" : "+v"(tq));
  const int lane = tq & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tq >> 6);
  // ---- load + window (+ half-bin shift for the odd bins): v[a] = x[512 a + t] * w.  All 64 sample
  // loads go out first (HBM / Infinity Cache latency, 128 registers); the window values (L2
  // resident, shared by every frame) follow in groups of 16 so that they never hold more than 32
  // registers next to the 64 the even-bin results occupy.
#pragma unroll
  for (int a = 0; a < 64; ++a) {
    if (PSD_ABL & 1) { v[a] = make_float2((float)(tq + a), (float)(tq ^ a)); continue; }
    const v2f_t s = *(const PYSDR_AS1 v2f_t*)(xf + 512 * a + tq);
    v[a] = make_float2(s.x, s.y);
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
#pragma unroll
  for (int g0 = 0; g0 < 64; g0 += 16) {
    if (PSD_ABL & 1) continue;
    if (E == 0) {
      float g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = *(const PYSDR_AS1 float*)(win + 512 * (g0 + i) + tq);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[g0 + i] = make_float2(v[g0 + i].x * g[i], v[g0 + i].y * g[i]);
    } else {
      v2f_t g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = *(const PYSDR_AS1 v2f_t*)(cwin1 + 512 * (g0 + i) + tq);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[g0 + i] = cmul(v[g0 + i], make_float2(g[i].x, g[i].y));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- steps 1 and 2 share one loop body (the kernel's code must stay inside the 64 KB instruction
  // cache: fully unrolled it was ~100 KB and the waves spent 40 % of their time waiting for
  // instructions): DFT64 over the register index, twiddle W_32768^(idx * k) with idx = t after
  // step 1 and idx = 64 c after step 2 (W_512^(c kB)), then the exchange that follows the step.
  STAMP(2);
#pragma unroll 1
  for (int step = 0; step < 2; ++step) {
    if (!(PSD_ABL & 32)) dft64(v);
    if (step == 0) STAMP(3); else STAMP(6);
    if (!(PSD_ABL & 16)) {
      // (the twiddles only depend on the thread index: without the opaque copy they are computed
      // ahead of the transform and held in 126 registers)
      int ti = (step == 0) ? tq : 64 * (lane & 7);
      asm volatile("" : "+v"(ti), "+v"(v[0].x));
      const float tr = (float)ti * (1.0f / 32768.0f);
#pragma unroll
      for (int r = 1; r < 64; ++r) {
        v[r] = cmul(v[r], expm2pi(tr * (float)fidx(r)));     // idx * k < 32768: exact
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (step == 0) STAMP(4); else STAMP(7);
    if (step == 0) {
      // ---- exchange 1 (workgroup)
      if (PSD_ABL & 4) continue;
      if (wave & 1) exchange1<1>(v, lds, lane, wave);
      else exchange1<0>(v, lds, lane, wave);
      STAMP(5);
    } else {
      if (PSD_ABL & 8) continue;
      // ---- exchange 2 (wave local, two alternating 4.5 KB areas per wave inside the exchange-1
      // region, which every wave has finished reading: the barrier that closed exchange 1)
      float2 u[64];
      const int c = lane & 7, k0 = lane >> 3;
      float2* area = lds + wave * (2 * 8 * 72);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float2* ar = area + (q & 1) * (8 * 72) + k0 * 72;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) ar[rr * 8 + (c ^ rr)] = v[8 * rr + q];   // kB = 8 q + rr
        asm volatile("" ::: "memory");       // LDS is in order per wave: no wait, only no compiler reordering
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) u[8 * q + cc] = ar[c * 8 + (cc ^ c)];   // this lane: kB = 8 q + c
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = u[i];
    }
  }
  float2 (&u)[64] = v;
  STAMP(8);
  // ---- step 3: DFT8 over c -> kC, power, dB
#pragma unroll
  for (int q = 0; q < 8; ++q) dft8<1>(&u[8 * q]);
  int t3 = lane;
  asm volatile("" : "+v"(t3), "+v"(u[0].x));
  const int kbase = (t3 >> 3) + 8 * wave + 64 * (t3 & 7);          // kA + 64 c'
  STAMP(9);
  float pw[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const float2 z = u[i];
    pw[i] = z.x * z.x + z.y * z.y;
  }
  if (db) {
#pragma unroll
    for (int i = 0; i < 64; ++i) pw[i] = 3.01029995663981195f * __log2f(pw[i] + 1.0e-30f);     // 10 log10
  }
  STAMP(10);
  if (E == 0) {
#pragma unroll
    for (int i = 0; i < 64; ++i) p0[i] = pw[i];
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const int k = kbase + 512 * q + 4096 * kc;
        const v2f_t o = {p0[8 * q + kc], pw[8 * q + kc]};
        __builtin_nontemporal_store(o, (PYSDR_AS1 v2f_t*)(of + ((k + (kNh >> 1)) & (kNh - 1))));
      }
  }
  STAMP(11);
  __syncthreads();          // exchange-2 areas are read: the next pass may overwrite the region
  STAMP(12);
}

__global__ __launch_bounds__(512) void psd_frame_kernel(const float2* __restrict__ x, size_t hop,
                                                        const float* __restrict__ win,
                                                        const float2* __restrict__ cwin1,
                                                        float* __restrict__ out, int db, int nframes) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int tid = threadIdx.x;
  // persistent workgroups (one per CU: all its registers and most of its LDS), frames dealt round robin
#pragma unroll 1
  for (int f = blockIdx.x; f < nframes; f += gridDim.x) {
    const float2* xf = x + (size_t)f * hop;
    float2* of = reinterpret_cast<float2*>(out + (size_t)f * 2 * kNh);
    float p0[64];
    const bool stamp_on = (blockIdx.x == 0) && (f == (int)(2 * gridDim.x));
    psd_pass<0>(xf, win, cwin1, of, db, lds, tid, p0, stamp_on);
    psd_pass<1>(xf, win, cwin1, of, db, lds, tid, p0, stamp_on);
  }
}

}  // namespace

#ifdef PSD_STAMP
int psd_read_stamps(unsigned long long* out) {
  PYSDR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psd_stamps), sizeof(unsigned long long) * 2 * 16 * 8));
  return PYSDR_OK;
}
#endif

size_t psd_frame_lds_bytes() { return (size_t)kLdsComplex * sizeof(float2); }

// nframes frames of 32768 complex samples every `hop` -> nframes x 65536 dB values, fftshifted.
// cwin1[n] = win[n] * exp(-j pi n / 32768)
int launch_psd64k_frames(const float2* x, size_t hop, int nframes, const float* win, const float2* cwin1,
                         float* out, int db, hipStream_t st) {
  static std::mutex attr_mu;
  static uint64_t attr_done = 0;
  {
    int dev = 0;
    PYSDR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(psd_frame_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)psd_frame_lds_bytes()));
      attr_done |= 1ull << (dev & 63);
    }
  }
  int ncu = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
  }
  const int grid = nframes < ncu ? nframes : ncu;
  hipLaunchKernelGGL(psd_frame_kernel, dim3(grid), dim3(512), psd_frame_lds_bytes(), st, x, hop, win, cwin1, out, db, nframes);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr

// ---- EXPERIMENT RECORD (round 2, MI355X) ---------------------------------------------------------
// Correct at the first run: max linear error 1.5e-6 of the peak against the two-kernel path, 2.4e-6 dB
// against a double-precision DFT on the strong bins, closer to the exact value than the two-kernel path
// on the weak ones (direct v_sin/v_cos twiddles instead of power chains).  Fabric traffic per frame
// 0.72 MB (2 x FETCH_SIZE + WRITE_SIZE) against 1.5 MB.  But 345 ns per frame against 261:
//   s_memtime stamps, one frame = 201 k cycles: x loads 15-31 k per pass (per-CU share of the HBM, all
//   CUs in the same phase), window 8-11 k (four L2 round trips), DFT64 6 k each, twiddles 1.5 k,
//   exchange 1 ~12 k (mostly the load skew of the 8 waves surfacing at the first barrier), exchange 2
//   12.5 k, DFT8 x 8 1-2 k, power + dB 4 k, the final stores 42 k (64-byte runs = half lines).
//   Ablations (ns per frame): no loads 235-253, no exchanges 297, compute only 185-201.
// What it would take: full-line stores through a third LDS transposition (-40 k), a cheaper exchange 2,
// the window in fewer round trips -- ~135 k cycles = 230 ns, no better than the two-kernel path; and the
// structural limit stays: one workgroup per CU (all 512 x 256 registers) cannot overlap its own loads
// with its own arithmetic, while the memory floor alone (0.75 MB per frame at the CU's 1/256 share of
// 6.3 TB/s) is 31 us = 75 k cycles per frame.  Lessons kept: -fno-slp-vectorize (packed
// f32 ops bring register-pair constraints and v_mov shuffles: 616 -> 268 spilled registers), opaque
// copies of the thread index right before each use (loop-invariant twiddles / addresses are otherwise
// hoisted and spilled), peeled passes instead of a loop (268 -> 20), one loop body for both DFT64 steps
// (code 100 KB -> 41 KB).
