// Rational resampler for ONE sub-receiver with a SHORT prototype and a small DOWN/UP ratio, without the raw-chunk
// peak: the fs1 -> FS_OUT audio stage of broadcast FM (250 kHz -> 48 kHz = 24/125, 64 taps per branch; DESIGN.md 3.10,
// the stage behind rx.demod's discriminator of Receiver.demod_data, receiver.py:235).  Same arithmetic contract as
// mixdec.hip:  y[m] = exp(j phi(n_m)) sum_k g[p_m][k] x[n_m-k],  n_m = floor((t0 + m*DOWN)/UP), p_m = (t0 + m*DOWN) mod UP.
//
// Why not mixdec_kernel: that kernel gives every output a DPP row of 16 lanes (built for >= 96 taps per branch and a
// stream that is read once at HBM speed).  With 64 taps an output's 16 lanes do 4 tap steps each and then pay the full
// fold, rotate and index arithmetic: ~100 instructions per 4 outputs, and with 24 branches the taps cannot stay in
// registers.  At the bench batch (10.9 M IF samples -> 2.1 M outputs) it took 117 us for 87 MB of input -- vector issue,
// not memory.  Here ONE thread owns one output: 64 x (two 8-byte LDS reads + 4 FMAs), no cross-lane step; a workgroup's
// 256 outputs span ~1400 consecutive inputs, staged once into LDS together with the (padded) tap table.
// The sum runs over k = 0 .. kpad-1 in order whatever the call, tile or thread: batch == chunk by chunk bit for bit.
#include <mutex>
#include "common.h"
#include "mixdec_geom.h"
#include "hist_roll.h"

namespace pysdr {

namespace {

constexpr int kRsThreads = 256;

__global__ __launch_bounds__(kRsThreads) void resamp_small_kernel(const MixDecArgs a, int span_cap, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) float2 rs_lds[];
  float2* const xs = rs_lds;                       // [span_cap] input span of one group of 256 outputs
  float2* const tl = rs_lds + span_cap;            // [up][kpad + 1] taps (one pad per row: rows kpad*8 bytes apart would share a bank)
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && a.hist_new != nullptr)    // the decimator's history roll rides in this launch (hist_roll.h)
    roll_history(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n, tid, kRsThreads);
  // the tap table once per workgroup (a workgroup walks every gridDim.x-th group of outputs)
  const int kp1 = a.kpad + 1;
  if (kRsThreads % a.kpad == 0) {
    const int k = tid % a.kpad, rows = kRsThreads / a.kpad;
    for (int p = tid / a.kpad; p < a.up; p += rows) tl[p * kp1 + k] = a.taps[p * a.kpad + k];
  } else {
    for (int j = tid; j < a.up * a.kpad; j += kRsThreads) {
      const int p = j / a.kpad, k = j - p * a.kpad;
      tl[p * kp1 + k] = a.taps[j];
    }
  }
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int i0 = grp * kRsThreads;
    const int n_here = (a.n_out - i0 < kRsThreads) ? a.n_out - i0 : kRsThreads;
    // input span [lo, hi] (relative to the call's first sample; negative = history)
    uint32_t q0, r0, q1, r1;
    divmod_magic(a.t0 + (uint32_t)i0 * (uint32_t)a.down, (uint32_t)a.up, a.magic, q0, r0);
    divmod_magic(a.t0 + (uint32_t)(i0 + n_here - 1) * (uint32_t)a.down, (uint32_t)a.up, a.magic, q1, r1);
    const int lo = (int)q0 - (a.kpad - 1), hi = (int)q1;
    __syncthreads();                               // the previous group's reads of xs are done (and the taps are in place)
    if (lo >= 0 && (uint32_t)hi < a.n_total) {
      for (int j = tid; j <= hi - lo; j += kRsThreads) xs[j] = a.x[lo + j];
    } else {
      for (int j = tid; j <= hi - lo; j += kRsThreads) {
        const int rel = lo + j;
        float2 v = make_float2(0.f, 0.f);
        if (rel >= 0) { if ((uint32_t)rel < a.n_total) v = a.x[rel]; }
        else if (rel >= -a.hist_len) v = a.hist[a.hist_len + rel];
        xs[j] = v;
      }
    }
    __syncthreads();
    if (tid < n_here) {
      const int i = i0 + tid;
      uint32_t q, p;
      divmod_magic(a.t0 + (uint32_t)i * (uint32_t)a.down, (uint32_t)a.up, a.magic, q, p);
      const float2* xp = xs + ((int)q - lo);       // x[n_m]; tap k reads xp[-k]
      const float2* tp = tl + (int)p * kp1;
      float sr = 0.f, si = 0.f;
#pragma unroll 8
      for (int k = 0; k < a.kpad; ++k) {
        const float2 g = tp[k], v = xp[-k];
        sr = fmaf(g.x, v.x, sr);
        sr = fmaf(-g.y, v.y, sr);
        si = fmaf(g.x, v.y, si);
        si = fmaf(g.y, v.x, si);
      }
      const uint32_t ph = a.phase0[0] + a.fword[0] * q;
      const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
      const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
      float2 o;
      o.x = sr * cs - si * sn;
      o.y = sr * sn + si * cs;
      a.y[0][i] = o;
    }
  }
}

// ---- the same stage, BRANCH-MAJOR (round 4).  The kernel above spends its time in the LDS: per tap every lane reads its own
// tap (24 branches across a wave) and its own sample, 5.2 samples from its neighbour's (2-3 lanes per bank pair): ~14 LDS cycles
// per wave and tap against 16 cycles of FMAs on ONE of the CU's four SIMDs -- the four together are LDS-bound at 48 of its
// 85 us.  Here a tile is UP x 32 consecutive outputs and a HALF-WAVE owns one polyphase branch: lane l of branch index b
// computes output i = l UP + b of the tile.  The 32 lanes of a half-wave then read ONE tap (a broadcast) and samples exactly
// DOWN apart -- an odd DOWN puts them on 32 different bank pairs -- so a tap costs ~8 LDS cycles and the FMAs bound the
// kernel.  The outputs go through LDS once more to leave as contiguous 8-byte-per-lane stores (lane l's own outputs are UP
// outputs apart).  Same arithmetic per output as resamp_small_kernel, in the same order: bit for bit its results.
constexpr int kRbL = 32;                           // outputs per branch and tile = lanes per half-wave
constexpr int kRbPre = 6;                          // pieces of the next tile's span a thread holds in registers

// (six waves per SIMD = two workgroups of 24 x 32 threads per CU: 84 registers)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void resamp_branch_kernel(const MixDecArgs a, int span_cap, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float2 rs_lds[];
  const int tile_out = a.up * kRbL;
  float2* const xs = rs_lds;                       // [span_cap] input span of one tile
  float2* const os = rs_lds + span_cap;            // [tile_out] the tile's outputs, in output order
  float2* const tl = os + tile_out;                // [up][kpad + 1] taps
  const int tid = threadIdx.x, nth = blockDim.x;   // nth = up * 32
  if (blockIdx.x == 0 && a.hist_new != nullptr)    // the decimator's history roll rides in this launch (hist_roll.h)
    roll_history(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n, tid, nth);
  const int kp1 = a.kpad + 1;
  for (int j = tid; j < a.up * a.kpad; j += nth) {
    const int p = j / a.kpad, k = j - p * a.kpad;
    tl[p * kp1 + k] = a.taps[j];
  }
  const int b = tid >> 5, l = tid & 31;            // branch index inside a group of UP outputs, position of that group in the tile
  const int iloc = l * a.up + b;
  // (t0 + (i0 + iloc) DOWN) mod UP does not depend on the tile: i0 is a multiple of UP
  uint32_t qb, pb;
  divmod_magic(a.t0 + (uint32_t)iloc * (uint32_t)a.down, (uint32_t)a.up, a.magic, qb, pb);
  const float2* const tp = tl + (int)pb * kp1;
  // The input span of the NEXT tile is loaded into registers (kRbPre pieces per thread) while this tile is multiplied out of
  // the LDS: a tile's ~2 us of load latency was as long as its arithmetic, and two workgroups per CU did not hide it.
  float2 pre[kRbPre];
  auto span_of = [&](int tile, int& lo, int& hi, int& n_here) {
    const int i0 = tile * tile_out;
    n_here = (a.n_out - i0 < tile_out) ? a.n_out - i0 : tile_out;
    uint32_t q0, r0, q1, r1;
    divmod_magic(a.t0 + (uint32_t)i0 * (uint32_t)a.down, (uint32_t)a.up, a.magic, q0, r0);
    divmod_magic(a.t0 + (uint32_t)(i0 + n_here - 1) * (uint32_t)a.down, (uint32_t)a.up, a.magic, q1, r1);
    lo = (int)q0 - (a.kpad - 1);
    hi = (int)q1;
  };
  auto fetch = [&](int lo, int hi) {
    if (lo >= 0 && (uint32_t)hi < a.n_total) {
#pragma unroll
      for (int u = 0; u < kRbPre; ++u) {
        const int j = tid + u * nth;
        pre[u] = (j <= hi - lo) ? a.x[lo + j] : make_float2(0.f, 0.f);   // (nontemporal: no difference, 47.8-48.9 against 48.2-49.3 us)
      }
    } else {
#pragma unroll
      for (int u = 0; u < kRbPre; ++u) {
        const int j = tid + u * nth, rel = lo + j;
        float2 v = make_float2(0.f, 0.f);
        if (j <= hi - lo) {
          if (rel >= 0) { if ((uint32_t)rel < a.n_total) v = a.x[rel]; }
          else if (rel >= -a.hist_len) v = a.hist[a.hist_len + rel];
        }
        pre[u] = v;
      }
    }
  };
  int lo = 0, hi = -1, n_here = 0;
  if ((int)blockIdx.x < ntiles) { span_of(blockIdx.x, lo, hi, n_here); fetch(lo, hi); }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int i0 = tile * tile_out;
    __syncthreads();                               // the previous tile's reads of xs / os are done (and the taps are in place)
#pragma unroll
    for (int u = 0; u < kRbPre; ++u) {
      const int j = tid + u * nth;
      if (j <= hi - lo) xs[j] = pre[u];
    }
    const int lo_t = lo, n_t = n_here;
    const int nxt = tile + gridDim.x;
    __syncthreads();
    // behind the barrier (hipcc waits for every outstanding load in front of one): in flight during the sums below
    if (nxt < ntiles) { span_of(nxt, lo, hi, n_here); fetch(lo, hi); }
    if (iloc < n_t) {
      // n of output i0 + iloc: i0 DOWN / UP is exact, so q = i0 DOWN / UP + qb
      const uint32_t q = (uint32_t)(i0 / a.up) * (uint32_t)a.down + qb;
      const float2* xp = xs + ((int)q - lo_t);     // x[n_m]; tap k reads xp[-k]
      float sr = 0.f, si = 0.f;
#pragma unroll 8
      for (int k = 0; k < a.kpad; ++k) {
        const float2 g = tp[k], v = xp[-k];
        sr = fmaf(g.x, v.x, sr);
        sr = fmaf(-g.y, v.y, sr);
        si = fmaf(g.x, v.y, si);
        si = fmaf(g.y, v.x, si);
      }
      const uint32_t ph = a.phase0[0] + a.fword[0] * q;
      const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
      const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
      os[iloc] = make_float2(sr * cs - si * sn, sr * sn + si * cs);
    }
    __syncthreads();
    if (tid < n_t) a.y[0][i0 + tid] = os[tid];
  }
}

// ---- the same stage with a WAVE per polyphase branch and the taps in SCALAR registers.  resamp_branch_kernel is still bound by
// the LDS: a broadcast read costs it what any 512-byte read costs, so a tap is 8 LDS cycles per wave (tap + sample) on a
// unit that serves four SIMDs -- 27 of its 47 us.  With ALL 64 lanes of a wave on one branch the tap is wave-uniform: it
// comes through the scalar cache (s_load from the constant address space) and enters the FMAs as an SGPR operand; the LDS
// serves the samples only.  A tile is then UP x 64 outputs (8000 input samples at 24/125: 64.5 KB + 12 KB of outputs,
// two workgroups per CU), a wave takes the branches w, w + nwaves, ...  Same sums in the same order: bit for bit the audio of
// the other two forms (test_wbfm_audio_resampler_forms_agree_bit_for_bit).
typedef float rs_cf2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) rs_cf2* rs_ctaps;

__global__ __launch_bounds__(1024) void resamp_wave_kernel(const MixDecArgs a, int span_cap, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float2 rs_lds[];
  const int tile_out = a.up * 64;
  float2* const xs = rs_lds;                       // [span_cap] input span of one tile
  float2* const os = rs_lds + span_cap;            // [tile_out] the tile's outputs, in output order
  const int tid = threadIdx.x, nth = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6, l = tid & 63;
  if (blockIdx.x == 0 && a.hist_new != nullptr)    // the decimator's history roll rides in this launch (hist_roll.h)
    roll_history(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n, tid, nth);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int i0 = tile * tile_out;
    const int n_here = (a.n_out - i0 < tile_out) ? a.n_out - i0 : tile_out;
    uint32_t q0, r0, q1, r1;
    divmod_magic(a.t0 + (uint32_t)i0 * (uint32_t)a.down, (uint32_t)a.up, a.magic, q0, r0);
    divmod_magic(a.t0 + (uint32_t)(i0 + n_here - 1) * (uint32_t)a.down, (uint32_t)a.up, a.magic, q1, r1);
    const int lo = (int)q0 - (a.kpad - 1), hi = (int)q1;
    __syncthreads();                               // the previous tile's reads of xs / os are done
    if (lo >= 0 && (uint32_t)hi < a.n_total) {
      for (int j = tid; j <= hi - lo; j += nth) xs[j] = a.x[lo + j];
    } else {
      for (int j = tid; j <= hi - lo; j += nth) {
        const int rel = lo + j;
        float2 v = make_float2(0.f, 0.f);
        if (rel >= 0) { if ((uint32_t)rel < a.n_total) v = a.x[rel]; }
        else if (rel >= -a.hist_len) v = a.hist[a.hist_len + rel];
        xs[j] = v;
      }
    }
    __syncthreads();
    for (int b = wave; b < a.up; b += nwaves) {    // this wave's branches: lane l owns output l UP + b of the tile
      const int iloc = l * a.up + b;
      // (t0 + (i0 + iloc) DOWN) mod UP is the same for every lane and every tile: i0 and l UP are multiples of UP
      uint32_t qb, pb;
      divmod_magic(a.t0 + (uint32_t)iloc * (uint32_t)a.down, (uint32_t)a.up, a.magic, qb, pb);
      const rs_ctaps tp = (rs_ctaps)(uintptr_t)a.taps + __builtin_amdgcn_readfirstlane((int)pb) * a.kpad;
      const uint32_t q = (uint32_t)(i0 / a.up) * (uint32_t)a.down + qb;
      const float2* xp = xs + ((int)q - lo);       // x[n_m]; tap k reads xp[-k]
      float sr = 0.f, si = 0.f;
      if (iloc < n_here) {
#pragma unroll 8
        for (int k = 0; k < a.kpad; ++k) {
          const rs_cf2 g = tp[k];
          const float2 v = xp[-k];
          sr = fmaf(g.x, v.x, sr);
          sr = fmaf(-g.y, v.y, sr);
          si = fmaf(g.x, v.y, si);
          si = fmaf(g.y, v.x, si);
        }
        const uint32_t ph = a.phase0[0] + a.fword[0] * q;
        const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
        const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
        os[iloc] = make_float2(sr * cs - si * sn, sr * sn + si * cs);
      }
    }
    __syncthreads();
    for (int j = tid; j < n_here; j += nth) a.y[0][i0 + j] = os[j];
  }
}

}  // namespace

// LDS of the wave-per-branch form: input span of UP x 64 outputs + the outputs; 0 = not eligible
static size_t resamp_wave_lds(int up, int down, int kpad, int* span_out) {
  if (up > 32 || up < 2) return 0;
  const long span = 64L * down + kpad + 4;
  const size_t bytes = ((size_t)span + (size_t)up * 64) * sizeof(float2);
  if (bytes > 78 * 1024) return 0;                 // two workgroups per CU
  *span_out = (int)span;
  return bytes;
}

// LDS of the branch-major form: input span of UP x 32 outputs + the outputs + the tap table; 0 = not eligible
static size_t resamp_branch_lds(int up, int down, int kpad, int* span_out) {
  if (up > 32 || up < 2) return 0;
  const long span = (long)kRbL * down + kpad + 4;
  const size_t bytes = ((size_t)span + (size_t)up * kRbL + (size_t)up * (kpad + 1)) * sizeof(float2);
  if (bytes > 64 * 1024 || span > (long)kRbPre * up * kRbL) return 0;
  *span_out = (int)span;
  return bytes;
}

// span of input samples 256 consecutive outputs need (+ the filter): what the launch reserves in LDS; 0 = not eligible
int resamp_small_span(int up, int down, int kpad) {
  const long span = ((long)kRsThreads * down + up - 1) / up + kpad + 4;
  const long bytes = (span + (long)up * (kpad + 1)) * (long)sizeof(float2);
  if (span > 4096 || bytes > 60 * 1024) return 0;
  return (int)span;
}

int launch_resamp_small(const MixDecArgs& a, int grid_cap, int plain, hipStream_t st) {
  const int span = resamp_small_span(a.up, a.down, a.kpad);
  if (span <= 0 || a.nrx != 1) {
    set_last_error("resamp_small: up %d down %d kpad %d nrx %d not eligible", a.up, a.down, a.kpad, a.nrx);
    return PYSDR_ERR_ARG;
  }
  if (a.n_out <= 0) return PYSDR_OK;
  // the branch-major form whenever its tile fits (a choice by the decimator's shape only: every call of a stream takes the
  // same path -- and both paths sum every output in the same order anyway)
  int bspan = 0;
  const size_t blds = resamp_branch_lds(a.up, a.down, a.kpad, &bspan);
  int wspan = 0;
  const size_t wlds = resamp_wave_lds(a.up, a.down, a.kpad, &wspan);
  if (wlds > 0 && plain == 0) {
    static std::mutex attr_mu;
    static uint64_t attr_done = 0;
    {
      int dev = 0;
      PYSDR_HIP_CHECK(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lk(attr_mu);
      if (!((attr_done >> (dev & 63)) & 1ull)) {
        PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resamp_wave_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done |= 1ull << (dev & 63);
      }
    }
    const int tile_out = a.up * 64;
    const int ntiles = (a.n_out + tile_out - 1) / tile_out;
    int wgrid = ntiles < 512 ? ntiles : 512;
    if (grid_cap > 0 && wgrid > grid_cap) wgrid = grid_cap;
    const int nw = (a.up + 1) / 2;                         // two branches per wave
    hipLaunchKernelGGL(resamp_wave_kernel, dim3(wgrid), dim3(64 * nw), wlds, st, a, wspan, ntiles);
    PYSDR_HIP_CHECK(hipGetLastError());
    return PYSDR_OK;
  }
  if (blds > 0 && plain != 1) {
    const int tile_out = a.up * kRbL;
    const int ntiles = (a.n_out + tile_out - 1) / tile_out;
    int bgrid = ntiles < 512 ? ntiles : 512;               // two workgroups per CU: one stages while the other multiplies
    if (grid_cap > 0 && bgrid > grid_cap) bgrid = grid_cap;   // tests: many tiles per workgroup in a small call
    hipLaunchKernelGGL(resamp_branch_kernel, dim3(bgrid), dim3(a.up * kRbL), blds, st, a, bspan, ntiles);
    PYSDR_HIP_CHECK(hipGetLastError());
    return PYSDR_OK;
  }
  const size_t lds = ((size_t)span + (size_t)a.up * (a.kpad + 1)) * sizeof(float2);
  const int ngroups = (a.n_out + kRsThreads - 1) / kRsThreads;
  int grid = ngroups < 2048 ? ngroups : 2048;              // ~8 workgroups per CU, each stages the tap table once
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  hipLaunchKernelGGL(resamp_small_kernel, dim3(grid), dim3(kRsThreads), lds, st, a, span, ngroups);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
