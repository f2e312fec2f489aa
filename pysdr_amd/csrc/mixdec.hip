// Fused complex-NCO mix + polyphase rational decimator for all sub-receivers of one
// wideband stream (gfx950).  Stands behind the first two stages of
// Receiver.demod_data (receiver.py:235): rx.lo mixer and rx.dec resampler.
//
//   y_r[m] = sum_k h_r[p_m + UP*k] * ( x[n_m-k] * exp(j*phi_r(n_m-k)) )
//          = exp(j*phi_r(n_m)) * sum_k g_r[p_m][k] * x[n_m-k],
//   g_r[p][k] = h_r[p + UP*k] * exp(-j*w_r*k)      (LO folded into the taps, host side)
//   n_m = floor(m*DOWN/UP), p_m = (m*DOWN) mod UP.
//
// So the NCO runs at the OUTPUT rate (48 kHz) only, and the input is touched once:
// HBM-bound streaming read of interleaved IQ, shared by every RX.
//
// Work decomposition
//   workgroup  = `tile_out` consecutive outputs = one contiguous input span
//                (tile_out*DOWN/UP + K samples) staged in LDS with 16-B/lane coalesced
//                loads; the raw-chunk peak |x|^2 (rx.auto_mute, receiver.py:239) is
//                reduced on the way in, so every input sample is read exactly once.
//   half-wave  = one output: 32 lanes split the K taps (ds_read_b64 of x is
//                conflict-free: 32 consecutive float2 = all 64 banks), each lane
//                accumulates all RX from one x read, then a DPP row reduction + one
//                row_bcast folds 32 lanes; lanes 16..16+nrx-1 of the half rotate by the
//                LO phase and store.
#include "common.h"

namespace pysdr {

namespace {

__device__ __forceinline__ float dpp_quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_mirror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}
// rows 1 and 3 receive lane 15 of the previous row, rows 0 and 2 receive 0
__device__ __forceinline__ float dpp_bcast15(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
}
// sum over each 32-lane half; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_quad_xor1(v);
  v += dpp_quad_xor2(v);
  v += dpp_half_mirror(v);
  v += dpp_mirror(v);
  v += dpp_bcast15(v);
  return v;
}

// one 16-byte LDS-DMA element: LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int R>
__global__ __launch_bounds__(1024) void mixdec_kernel(const MixDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* xs = lds;                       // [tile_cap]
  float2* tl = lds + a.tile_cap;          // [R][up][kpad]

  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), give
  // each XCD a contiguous run of tiles so neighbouring halos hit the same L2.
  int b = blockIdx.x;
  {
    const int nb = gridDim.x;
    const int per = nb >> 3;
    if (per > 0 && b < (per << 3)) b = (b & 7) * per + (b >> 3);
  }
  const int i_first = b * a.tile_out;
  int tile_n = a.n_out - i_first;
  if (tile_n > a.tile_out) tile_n = a.tile_out;
  if (tile_n < 0) tile_n = 0;
  const int i_last = i_first + tile_n - 1;
  const bool last_tile = (b == a.ntiles - 1);

  // input span needed by the outputs + the samples this tile "owns" for the peak scan
  int own_lo, own_hi, need_lo, need_hi;
  if (tile_n > 0) {
    need_hi = (int)((a.t0 + (uint32_t)i_last * (uint32_t)a.down) / (uint32_t)a.up);
    need_lo = (int)((a.t0 + (uint32_t)i_first * (uint32_t)a.down) / (uint32_t)a.up) - (a.kpad - 1);
    own_hi = need_hi;
  } else {
    need_hi = -1; need_lo = 0; own_hi = -1;
  }
  own_lo = (b == 0) ? 0
                    : (int)((a.t0 + (uint32_t)(i_first - 1) * (uint32_t)a.down) / (uint32_t)a.up) + 1;
  if (last_tile) own_hi = (int)a.n_total - 1;
  int lo = need_lo < own_lo ? need_lo : own_lo;
  if (tile_n == 0) lo = own_lo;
  lo &= ~1;
  int hi = need_hi > own_hi ? need_hi : own_hi;
  const int npairs = (hi - lo + 2) >> 1;

  const int wave = tid >> 6, nwaves = nthr >> 6;
  const int lane = tid & 63;

  if (a.dbg & 2) {
  } else if (a.aligned16) {
    // ---- LDS-DMA staging (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR
    // round trip, every piece of the tile in flight at once).  The LDS image is
    // lane-linear: piece q covers float4 slots [64q, 64q+64).
    {
      const int nt4 = (R * a.up * a.kpad) >> 1;                 // taps as 16-B slots
      const float4* src = reinterpret_cast<const float4*>(a.taps);
      float4* dst = reinterpret_cast<float4*>(tl);
      for (int q = wave; q * 64 < nt4; q += nwaves) {
        const int slot = q * 64 + lane;
        if (slot < nt4) glds16(src + slot, dst + q * 64);
      }
    }
    float4* dst = reinterpret_cast<float4*>(xs);
    for (int q = wave; q * 64 < npairs; q += nwaves) {
      const int pi = q * 64 + lane;
      const int rel = lo + 2 * pi;
      // a pair is DMA-able when both samples exist: history (rel < 0) or rel+1 < n_total
      const bool ok = pi < npairs && (rel < 0 || (uint32_t)rel + 1u < a.n_total);
      const float2* src = (rel >= 0) ? (a.x + rel) : (a.hist + (a.hist_len + rel));
      if (ok) glds16(src, dst + q * 64);
    }
    // the one pair that straddles the end of an odd-length call
    if (tid == 0 && (a.n_total & 1u)) {
      const int rel = (int)a.n_total - 1;
      if (rel >= lo && rel <= hi) {
        const float2 p0 = a.x[rel];
        *reinterpret_cast<float4*>(xs + (rel - lo)) = make_float4(p0.x, p0.y, 0.f, 0.f);
      }
    }
  } else {
    // ---- generic staging for inputs that are only 8-byte aligned (slow path)
    const int nt = R * a.up * a.kpad;
    for (int i = tid; i < nt; i += nthr) tl[i] = a.taps[i];
    for (int pi = tid; pi < npairs; pi += nthr) {
      const int rel = lo + 2 * pi;
      float2 p0 = make_float2(0.f, 0.f), p1 = make_float2(0.f, 0.f);
      if (rel < 0) { p0 = a.hist[a.hist_len + rel]; p1 = a.hist[a.hist_len + rel + 1]; }
      else {
        if ((uint32_t)rel < a.n_total) p0 = a.x[rel];
        if ((uint32_t)rel + 1u < a.n_total) p1 = a.x[rel + 1];
      }
      xs[2 * pi] = p0;
      xs[2 * pi + 1] = p1;
    }
  }
  __syncthreads();

  // ---- raw-chunk peak |x|^2 over the samples this tile owns (rx.auto_mute input).
  // A tile may straddle chunk boundaries: one wave-reduced scan + one atomic per wave
  // for every chunk it touches (same-address atomics are slow: never one per sample).
  if (own_hi >= own_lo && !(a.dbg & 4)) {
    const uint32_t c_lo = (uint32_t)own_lo / a.chunk_len;
    const uint32_t c_hi = (uint32_t)own_hi / a.chunk_len;
    for (uint32_t c = c_lo; c <= c_hi; ++c) {
      const long long cb = (long long)c * a.chunk_len;
      const int s_lo = own_lo > cb ? own_lo : (int)cb;
      const long long ce = cb + a.chunk_len - 1;
      const int s_hi = own_hi < ce ? own_hi : (int)ce;
      const int p_lo = (s_lo - lo) >> 1, p_hi = (s_hi - lo) >> 1;
      float pk = 0.f;
      for (int pi = p_lo + tid; pi <= p_hi; pi += nthr) {
        const float4 v = *reinterpret_cast<const float4*>(xs + 2 * pi);
        const int rel = lo + 2 * pi;
        const float e0 = (rel >= s_lo) ? v.x * v.x + v.y * v.y : 0.f;
        const float e1 = (rel + 1 <= s_hi) ? v.z * v.z + v.w * v.w : 0.f;
        pk = fmaxf(pk, fmaxf(e0, e1));
      }
      pk = wave_max(pk);
      if (lane == 0 && pk > 0.f) atomicMax(a.peak + c, __float_as_uint(pk));
    }
  }

  // ---- polyphase dot products: one output per 32-lane half
  const int half = lane >> 5, s = lane & 31;
  const int npq = (tile_n + 1) >> 1;
  for (int pq = wave; pq < ((a.dbg & 1) ? 0 : npq); pq += nwaves) {
    int i = i_first + 2 * pq + half;
    const bool valid = (i <= i_last);
    if (!valid) i = i_last;
    const uint32_t t = a.t0 + (uint32_t)i * (uint32_t)a.down;
    const uint32_t rel = t / (uint32_t)a.up;
    const uint32_t p = t - rel * (uint32_t)a.up;
    const float2* xp = xs + ((int)rel - lo - s);
    const float2* tp = tl + p * a.kpad + s;
    float ar[R], ai[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
    for (int j = 0; j < a.kpad; j += 32) {
      const float2 xv = xp[-j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float2 g = tp[r * a.up * a.kpad + j];
        ar[r] = fmaf(g.x, xv.x, ar[r]);
        ar[r] = fmaf(-g.y, xv.y, ar[r]);
        ai[r] = fmaf(g.x, xv.y, ai[r]);
        ai[r] = fmaf(g.y, xv.x, ai[r]);
      }
    }
    float sr = 0.f, si = 0.f;
    uint32_t p0 = 0u, fw = 0u;
    float2* yp = nullptr;
    const int myr = s - 16;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float tr = half_wave_sum(ar[r]);
      const float ti = half_wave_sum(ai[r]);
      if (myr == r) { sr = tr; si = ti; p0 = a.phase0[r]; fw = a.fword[r]; yp = a.y[r]; }
    }
    if (valid && myr >= 0 && myr < R) {
      const uint32_t ph = p0 + fw * rel;
      float sn, cs;
      sincospif((float)(int)ph * (1.0f / 2147483648.0f), &sn, &cs);
      float2 o;
      o.x = sr * cs - si * sn;
      o.y = sr * sn + si * cs;
      yp[i] = o;
    }
  }
}

template <int R>
int launch_r(const MixDecArgs& a, int threads, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mixdec_kernel<R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_last_error("hipFuncSetAttribute(mixdec<%d>): %s", R, hipGetErrorString(e));
      return PYSDR_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(mixdec_kernel<R>, dim3(a.ntiles), dim3(threads), lds, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace

size_t mixdec_lds_bytes(const MixDecArgs& a) {
  return ((size_t)a.tile_cap + (size_t)a.nrx * a.up * a.kpad) * sizeof(float2);
}

int launch_mixdec(const MixDecArgs& a, int threads, hipStream_t st) {
  const size_t lds = mixdec_lds_bytes(a);
  if (lds > 160 * 1024) {
    set_last_error("mixdec: LDS request %zu > 160 KiB", lds);
    return PYSDR_ERR_ARG;
  }
  switch (a.nrx) {
    case 1: return launch_r<1>(a, threads, lds, st);
    case 2: return launch_r<2>(a, threads, lds, st);
    case 3: return launch_r<3>(a, threads, lds, st);
    case 4: return launch_r<4>(a, threads, lds, st);
    case 5: return launch_r<5>(a, threads, lds, st);
    case 6: return launch_r<6>(a, threads, lds, st);
    case 7: return launch_r<7>(a, threads, lds, st);
    case 8: return launch_r<8>(a, threads, lds, st);
    default: set_last_error("mixdec: nrx=%d", a.nrx); return PYSDR_ERR_ARG;
  }
}

}  // namespace pysdr
