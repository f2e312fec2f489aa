// Stage 2 of Receiver.demod_data (receiver.py:235) at FS_OUT: per-mode detector
// (rx.demod), AF filter (rx.demod.filter_bank_real/cmpx, receiver.py:873-874), block
// AGC (rx.agc, watchdog.py:298-302) and the history roll that makes chunked ==
// one-shot (sigs/iir.py:83-125).  The data rate here is UP/DOWN (~1/167) of the input
// rate, so these kernels are latency-, not bandwidth-, critical.
#include "common.h"

namespace pysdr {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// chunk (AGC block) that output i belongs to: the chunk holding its newest input sample
__device__ __forceinline__ uint32_t block_of(const Stage2Args& a, int r, int i) {
  if (a.single_block[r]) return 0u;
  const uint32_t t = a.t0 + (uint32_t)i * (uint32_t)a.down;
  return (t / (uint32_t)a.up) / a.chunk_len;
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- AM-Synch carrier PLL (rx.demod.am_pll, receiver.py:649): theta feeds back through
// sin/cos and atan2, so the recursion itself is serial.  One wave walks a range in blocks of 64
// samples loaded coalesced and broadcast with v_readlane; every lane runs the same recursion,
// lane j keeps v = y*exp(-j*theta) of sample j for the detector stage, stored coalesced.
// Nothing but arithmetic sits on the critical path (sin/cos by v_sin/v_cos on theta/2pi).
// Parallelism comes from time: segments with warm-up + a patch-up pass (PllPlan, common.h).
template <bool EMIT>
__device__ __forceinline__ void am_pll_walk(const Stage2Args& a, const float2* __restrict__ y,
                                            float2* __restrict__ o, int i_begin, int i_end, float& th,
                                            float& w, int lane) {
  const float kp = a.pll_kp, ki = a.pll_ki;
  const float pi = 3.14159265358979323846f, twopi = 6.28318530717958647692f;
  const float inv2pi = 0.15915494309189533577f;
  float2 y_next = (i_begin + lane < i_end) ? y[i_begin + lane] : make_float2(0.f, 0.f);
  for (int i0 = i_begin; i0 < i_end; i0 += 64) {
    const float2 yv = y_next;
    const int nidx = i0 + 64 + lane;
    y_next = (nidx < i_end) ? y[nidx] : make_float2(0.f, 0.f);     // in flight during the steps below
    float2 mine = make_float2(0.f, 0.f);
    const int count = (i_end - i0 < 64) ? i_end - i0 : 64;
#pragma unroll 4
    for (int j = 0; j < count; ++j) {
      const float yr = lane_bcast(yv.x, j), yi = lane_bcast(yv.y, j);
      const float rev = th * inv2pi;
      const float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
      const float vr = yr * c + yi * s;
      const float vi = yi * c - yr * s;
      const float e = atan2f(vi, vr);
      w = w + ki * e;
      th = th + (w + kp * e);
      if (th >= pi) th -= twopi;
      else if (th < -pi) th += twopi;
      if (lane == j) mine = make_float2(vr, vi);
    }
    if (EMIT && lane < count) o[i0 + lane] = mine;
  }
}

// tolerances of the patch-up pass: the carrier loop's trajectories merge to identical floats
// (scripts/experiments/pll_warmup.py); 2e-6 rad of phase = 2e-6 of the detector output
__device__ __forceinline__ bool am_state_differs(float th_a, float w_a, float th_b, float w_b) {
  const float pi = 3.14159265358979323846f, twopi = 6.28318530717958647692f;
  float d = th_a - th_b;
  if (d >= pi) d -= twopi;
  else if (d < -pi) d += twopi;
  return !(fabsf(d) <= 2.0e-6f && fabsf(w_a - w_b) <= 2.0e-8f);
}

// grid (K, nrx): segment k of RX r
__global__ __launch_bounds__(64) void am_pll_seg_kernel(const Stage2Args a) {
  const int r = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  if (a.det[r] != kDetPll) return;
  const PllPlan& pl = a.pll;
  const RxDevState* st = a.state + r;
  const int n = a.n_out;
  const int s0 = k * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
  float th = st->pll_theta, w = st->pll_w;
  int wb = s0 - pl.W;
  if (k > 0 && wb > 0) {
    // guessed state W samples ahead of the segment: free-running at the call's initial rate
    th = 0.f;
  } else {
    wb = 0;                                  // the true state of the call: exact, however short
  }
  if (wb < s0) am_pll_walk<false>(a, a.y[r], a.ypll[r], wb, s0, th, w, lane);
  uint32_t* sg = pl.seg + ((size_t)r * pl.K + k) * 4;
  if (lane == 0) { sg[0] = __float_as_uint(th); sg[1] = __float_as_uint(w); }
  am_pll_walk<true>(a, a.y[r], a.ypll[r], s0, s1, th, w, lane);
  if (lane == 0) { sg[2] = __float_as_uint(th); sg[3] = __float_as_uint(w); }
}

// grid (nrx): walk the chain of segments, redo what does not join up
__global__ __launch_bounds__(64) void am_pll_patch_kernel(const Stage2Args a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (a.det[r] != kDetPll) return;
  const PllPlan& pl = a.pll;
  const int K = pl.K, n = a.n_out;
  const uint32_t* sg = pl.seg + (size_t)r * K * 4;
  float th_fin = __uint_as_float(sg[(size_t)(K - 1) * 4 + 2]), w_fin = __uint_as_float(sg[(size_t)(K - 1) * 4 + 3]);
  int patched = 0;
  int k = 1;
  while (k < K) {
    int bad = K;
    for (int base = k; base < K && bad == K; base += 64) {
      const int kk = base + lane;
      bool mm = false;
      if (kk < K)
        mm = am_state_differs(__uint_as_float(sg[(size_t)(kk - 1) * 4 + 2]), __uint_as_float(sg[(size_t)(kk - 1) * 4 + 3]),
                              __uint_as_float(sg[(size_t)kk * 4 + 0]), __uint_as_float(sg[(size_t)kk * 4 + 1]));
      const unsigned long long bal = __ballot(mm);
      if (bal) bad = base + __builtin_ctzll(bal);
    }
    if (bad >= K) break;
    float th = __uint_as_float(sg[(size_t)(bad - 1) * 4 + 2]), w = __uint_as_float(sg[(size_t)(bad - 1) * 4 + 3]);
    int j = bad;
    bool joined = false;
    while (j < K) {
      const int s0 = j * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
      am_pll_walk<true>(a, a.y[r], a.ypll[r], s0, s1, th, w, lane);
      ++patched;
      ++j;
      if (j < K && !am_state_differs(th, w, __uint_as_float(sg[(size_t)j * 4 + 0]), __uint_as_float(sg[(size_t)j * 4 + 1]))) {
        joined = true;                        // segment j was started from (nearly) this state: it stands
        break;
      }
    }
    if (!joined) { th_fin = th; w_fin = w; break; }
    k = j + 1;
  }
  if (lane == 0) {
    RxDevState* st = a.state + r;
    st->pll_theta = th_fin;
    st->pll_w = w_fin;
    st->pll_segments = K;
    st->pll_patched = patched;
  }
}

// ---- detector + AF FIR + block peak.  grid = (tiles, nrx), 2048 outputs per workgroup,
// EIGHT consecutive outputs per thread.  The detector output d is staged in LDS as eight
// de-interleaved sub-arrays D_m[q] = d[8q+m]: thread t owns outputs 8t..8t+7, and at tap k
// needs d[8t+j-k] -- a window that slides by ONE element per tap, so each tap costs one
// conflict-free LDS read (lanes read consecutive q of one sub-array) for eight outputs.
// The (<= 256) taps live in VGPRs, one per lane, and are broadcast with v_readlane.
// Arithmetic is specialised by what the mode needs (kFir* below): AM/NFM/AM-Synch have a
// real detector and real taps (1 FMA per tap), the SSB family needs only Re(c*d) (2 FMA).
constexpr int kW = 8;                  // outputs per thread = window length
constexpr int kFirThreads = 256;
constexpr int kFirOut = kW * kFirThreads;   // outputs per workgroup
constexpr int kFirRealReal = 0;        // d real, c real    -> real
constexpr int kFirRePart = 1;          // d cplx, c cplx    -> Re(c*d)
constexpr int kFirCplx = 2;            // d cplx, c cplx    -> cplx (IQ)

__device__ __forceinline__ float2 detect(const Stage2Args& a, int r, int det, const float2* y, int i) {
  float2 d = make_float2(0.f, 0.f);
  if (i >= a.n_out) return d;
  const float2 yc = y[i];
  if (det == kDetAbs) {
    d.x = sqrtf(yc.x * yc.x + yc.y * yc.y);
  } else if (det == kDetFm) {
    // sigs/nfm.m:124-127: fm = Re(y1)*Im(d) - Im(y1)*Re(d), d = y[n+1]-y[n-1]
    const float2 y1 = y[i - 1], ya = y[i - 2];
    const float dr = yc.x - ya.x, di = yc.y - ya.y;
    const float fm = y1.x * di - y1.y * dr;
    const float den = 2.f * (y1.x * y1.x + y1.y * y1.y) + 1e-20f;
    d.x = (fm / den) * a.fm_scale;
  } else if (det == kDetBfo) {
    const uint32_t ph = a.bfo_fword[r] * (a.m0_lo + (uint32_t)i);
    const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
    d = cmul(yc, make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)));
  } else if (det == kDetPll) {
    d.x = yc.x;
  } else {
    d = yc;
  }
  return d;
}

// One tap for the kW outputs of a lane (window slots rotate instead of moving registers).
template <int KIND>
__device__ __forceinline__ void fir_tap(float2 c, int u, const float (&wr)[kW], const float (&wi)[kW],
                                        float2 (&acc)[kW]) {
#pragma unroll
  for (int j = 0; j < kW; ++j) {
    const int sl = (j - u) & (kW - 1);
    acc[j].x = fmaf(c.x, wr[sl], acc[j].x);
    if (KIND != kFirRealReal) {
      acc[j].x = fmaf(-c.y, wi[sl], acc[j].x);
      if (KIND == kFirCplx) {
        acc[j].y = fmaf(c.x, wi[sl], acc[j].y);
        acc[j].y = fmaf(c.y, wr[sl], acc[j].y);
      }
    }
  }
}

// TAPS_IN_LANES: the (<= 256) taps sit in four VGPR pairs, lane l holding taps l, 64+l, ...;
// each tap is broadcast with v_readlane -- no scalar-cache load (an s_load in flight forces
// every LDS wait to drain lgkmcnt completely) and no extra LDS read.
template <int KIND, bool TAPS_IN_LANES>
__device__ __forceinline__ void fir_window(int hq, const float2* __restrict__ taps, int ngroups,
                                           const float* dre, const float* dim, int sub, int tid,
                                           float2 (&acc)[kW]) {
  // window w[j] = d[e0 + j - k], e0 = kW*(hq + tid); element e lives at D[e % kW][e / kW]
  float wr[kW], wi[kW];
#pragma unroll
  for (int j = 0; j < kW; ++j) {
    wr[j] = dre[j * sub + hq + tid];
    wi[j] = (KIND == kFirRealReal) ? 0.f : dim[j * sub + hq + tid];
    acc[j] = make_float2(0.f, 0.f);
  }
  // after tap k = kW*g+u the window moves down one element: the slot that held d[e0+kW-1-k]
  // is refilled with d[e0-k-1] = D[kW-1-u][hq + tid - g - 1]
  if (TAPS_IN_LANES) {
    const int lane = tid & 63;
    constexpr int kGroupsPerReg = 64 / kW;
    float2 tc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      tc[jb] = (64 * jb + lane < kW * ngroups) ? taps[64 * jb + lane] : make_float2(0.f, 0.f);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      const int left = ngroups - kGroupsPerReg * jb;
      const int g_end = left < kGroupsPerReg ? left : kGroupsPerReg;
      for (int gg = 0; gg < g_end; ++gg) {
        const int q = hq + tid - (kGroupsPerReg * jb + gg) - 1;
#pragma unroll
        for (int u = 0; u < kW; ++u) {
          const float2 c = make_float2(lane_bcast(tc[jb].x, kW * gg + u),
                                       KIND == kFirRealReal ? 0.f : lane_bcast(tc[jb].y, kW * gg + u));
          fir_tap<KIND>(c, u, wr, wi, acc);
          const int m = (kW - 1 - u) & (kW - 1);
          wr[m] = dre[m * sub + q];
          if (KIND != kFirRealReal) wi[m] = dim[m * sub + q];
        }
      }
    }
  } else {
    for (int g = 0; g < ngroups; ++g) {
      const int q = hq + tid - g - 1;
#pragma unroll
      for (int u = 0; u < kW; ++u) {
        fir_tap<KIND>(taps[kW * g + u], u, wr, wi, acc);      // wave-uniform: scalar load
        const int m = (kW - 1 - u) & (kW - 1);
        wr[m] = dre[m * sub + q];
        if (KIND != kFirRealReal) wi[m] = dim[m * sub + q];
      }
    }
  }
}

__global__ __launch_bounds__(kFirThreads) void demod_fir_kernel(const Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int r = blockIdx.y;
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * kFirOut;
  const int det = a.det[r];
  const float2* y = (det == kDetPll) ? a.ypll[r] : a.y[r];
  const int ngroups = (a.ntaps + kW - 1) / kW;           // taps are zero padded to kW*ngroups
  const int hq = ngroups;                                // history groups in front of the tile
  const int sub = kFirOut / kW + hq;                     // length of one sub-array
  float* dre = lds_f;                                    // [kW][sub]
  float* dim = lds_f + kW * sub;                         // [kW][sub]
  const bool real_det = (det == kDetAbs || det == kDetFm || det == kDetPll);

  const float2* taps = a.aftaps[r];
  const int kind = a.out_complex[r] ? kFirCplx : (real_det && a.taps_real[r] ? kFirRealReal : kFirRePart);
  // stage d[i0 - kW hq + e], e = 0 .. kW*sub-1
  for (int e = tid; e < kW * sub; e += kFirThreads) {
    const float2 d = detect(a, r, det, y, i0 - kW * hq + e);
    dre[(e & (kW - 1)) * sub + (e / kW)] = d.x;
    if (kind != kFirRealReal) dim[(e & (kW - 1)) * sub + (e / kW)] = d.y;
  }
  __syncthreads();

  float2 acc[kW];
  if (ngroups * kW <= 256) {
    if (kind == kFirRealReal) fir_window<kFirRealReal, true>(hq, taps, ngroups, dre, dim, sub, tid, acc);
    else if (kind == kFirRePart) fir_window<kFirRePart, true>(hq, taps, ngroups, dre, dim, sub, tid, acc);
    else fir_window<kFirCplx, true>(hq, taps, ngroups, dre, dim, sub, tid, acc);
  } else {
    if (kind == kFirRealReal) fir_window<kFirRealReal, false>(hq, taps, ngroups, dre, dim, sub, tid, acc);
    else if (kind == kFirRePart) fir_window<kFirRePart, false>(hq, taps, ngroups, dre, dim, sub, tid, acc);
    else fir_window<kFirCplx, false>(hq, taps, ngroups, dre, dim, sub, tid, acc);
  }

  const int ib = i0 + kW * tid;
  float mag = 0.f;
  uint32_t blk_lo = 0xFFFFFFFFu, blk_hi = 0xFFFFFFFFu;
  if (ib < a.n_out) {
    const int last = (ib + kW - 1 < a.n_out) ? ib + kW - 1 : a.n_out - 1;
    blk_lo = block_of(a, r, ib);
    blk_hi = block_of(a, r, last);
#pragma unroll
    for (int j = 0; j < kW; ++j)
      if (ib + j < a.n_out) {
        a.a[r][ib + j] = acc[j];
        const float m = a.out_complex[r] ? sqrtf(acc[j].x * acc[j].x + acc[j].y * acc[j].y) : fabsf(acc[j].x);
        if (blk_lo == blk_hi) mag = fmaxf(mag, m);
        else atomicMax(a.blkpeak + (size_t)r * a.nchunks + block_of(a, r, ib + j), __float_as_uint(m));
      }
  }
  // NFM noise squelch (sigs/squelch.m:92-145): block sum of |2nd difference| of the
  // detector output -- out-of-band noise rises when the carrier goes away
  const bool squelch = (a.sq_thresh[r] > 0.f) && (det == kDetFm);
  float nz = 0.f;
  unsigned cnt = 0u;
  if (squelch && ib < a.n_out) {
#pragma unroll
    for (int j = 0; j < kW; ++j)
      if (ib + j < a.n_out) {
        const int e = kW * (hq + tid) + j;                // element of output ib+j
        const float d0 = dre[(e & (kW - 1)) * sub + (e / kW)];
        const float d1 = dre[((e - 1) & (kW - 1)) * sub + ((e - 1) / kW)];
        const float d2 = dre[((e - 2) & (kW - 1)) * sub + ((e - 2) / kW)];
        const float hp = fabsf(d0 - 2.f * d1 + d2);
        if (blk_lo == blk_hi) { nz += hp; cnt += 1u; }
        else {
          const uint32_t bj = block_of(a, r, ib + j);
          atomicAdd(a.blknoise + (size_t)r * a.nchunks + bj, hp);
          atomicAdd(a.blkcnt + (size_t)r * a.nchunks + bj, 1u);
        }
      }
  }
  // block peak: one atomic per wave when the whole wave sits inside one block
  const uint32_t b0 = __shfl(blk_lo, 0);
  const bool uniform = __all((blk_lo == b0 && blk_hi == b0) || blk_lo == 0xFFFFFFFFu);
  if (uniform) {
    float m = mag;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0 && b0 != 0xFFFFFFFFu)
      atomicMax(a.blkpeak + (size_t)r * a.nchunks + b0, __float_as_uint(m));
    if (squelch) {
      float sn = nz;
      unsigned sc = cnt;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { sn += __shfl_xor(sn, o); sc += __shfl_xor(sc, o); }
      if ((tid & 63) == 0 && b0 != 0xFFFFFFFFu) {
        atomicAdd(a.blknoise + (size_t)r * a.nchunks + b0, sn);
        atomicAdd(a.blkcnt + (size_t)r * a.nchunks + b0, sc);
      }
    }
  } else if (blk_lo != 0xFFFFFFFFu && blk_lo == blk_hi) {
    atomicMax(a.blkpeak + (size_t)r * a.nchunks + blk_lo, __float_as_uint(mag));
    if (squelch) {
      atomicAdd(a.blknoise + (size_t)r * a.nchunks + blk_lo, nz);
      atomicAdd(a.blkcnt + (size_t)r * a.nchunks + blk_lo, cnt);
    }
  }
}

// ---- AGC recursion over the blocks of this call (sigs/agc.m:6-12 loop filter on decay,
// immediate attack).  One workgroup per RX.  Only the envelope recursion is serial
// (lane 0, out of LDS, ~5 dependent VALU ops per block); the gains -- one IEEE division
// each -- are then computed by all lanes in parallel.
__global__ __launch_bounds__(256) void agc_scan_kernel(const Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) float agc_lds[];
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  float* pk = agc_lds;                 // [nchunks] block peaks, then envelopes
  for (int c = tid; c < a.nchunks; c += 256)
    pk[c] = __uint_as_float(a.blkpeak[(size_t)r * a.nchunks + c]);
  __syncthreads();
  const RxDevState st = a.state[r];
  float last_peak = st.maxbuf;
  if (tid == 0) {
    const float beta = 0.1f;
    float env = st.env;
    for (int c = 0; c < a.nchunks; ++c) {
      const float peak = pk[c];
      const float dec = __fadd_rn(env, __fmul_rn(beta, __fsub_rn(peak, env)));
      env = (peak > env) ? peak : dec;
      pk[c] = env;
      last_peak = peak;
    }
  }
  __syncthreads();
  for (int c = tid; c < a.nchunks; c += 256) {
    const float g = st.agc_enable ? fminf(__fdiv_rn(st.ref, fmaxf(pk[c], 1e-12f)), 1.0e4f) : 1.f;
    a.gain[(size_t)r * a.nchunks + c] = g;
    // the raw block peaks are consumed: leave them zeroed for the next call
    a.blkpeak[(size_t)r * a.nchunks + c] = 0u;
  }
  if (a.sq_thresh[r] > 0.f) {
    __syncthreads();
    if (tid == 0) {
      // serial one-pole smoothing of the block noise, gate the gain (0 = squelched)
      float lvl = st.sq_level;
      int open = st.sq_open;
      for (int c = 0; c < a.nchunks; ++c) {
        const size_t k = (size_t)r * a.nchunks + c;
        const unsigned n = a.blkcnt[k];
        if (n > 0u) {
          const float noise = __fdiv_rn(a.blknoise[k], (float)n);
          lvl = __fadd_rn(lvl, __fmul_rn(0.64f, __fsub_rn(noise, lvl)));
          open = (lvl <= a.sq_thresh[r]) ? 1 : 0;
          if (!open) a.gain[k] = 0.f;
        }
        a.blknoise[k] = 0.f;
        a.blkcnt[k] = 0u;
      }
      a.state[r].sq_level = lvl;
      a.state[r].sq_open = open;
    }
  }
  if (tid == 0 && a.nchunks > 0) {
    const float env = pk[a.nchunks - 1];
    const float g = st.agc_enable ? fminf(__fdiv_rn(st.ref, fmaxf(env, 1e-12f)), 1.0e4f) : 1.f;
    // field-wise: the squelch block above owns sq_level / sq_open
    a.state[r].env = env;
    a.state[r].gain = g;
    a.state[r].maxbuf = last_peak;
    a.state[r].err = __fsub_rn(st.ref, __fmul_rn(g, last_peak));
  }
}

// ---- apply the block gain, emit rx.am (real, or complex in IQ mode)
__global__ __launch_bounds__(256) void apply_kernel(const Stage2Args a) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_out) return;
  const float g = a.gain[(size_t)r * a.nchunks + block_of(a, r, i)];
  const float2 v = a.a[r][i];
  if (a.matrix[r]) {
    reinterpret_cast<float2*>(a.am[r])[i] = make_float2((v.x + v.y) * g, (v.x - v.y) * g);
  } else if (a.out_complex[r]) {
    reinterpret_cast<float2*>(a.am[r])[i] = make_float2(v.x * g, v.y * g);
  } else {
    a.am[r][i] = v.x * g;
  }
}

// ---- history roll of the FS_OUT-rate buffers: prefix <- last hy outputs.  One workgroup
// per buffer so overlapping source/destination ranges are safe.
__global__ __launch_bounds__(256) void epilogue_kernel(const EpilogueArgs a) {
  const int job = blockIdx.x;
  const int tid = threadIdx.x;
  float2* base = (job < a.nrx) ? a.ybase[job] : a.ypllbase[job - a.nrx];
  if (base == nullptr) return;
  // element j of the new prefix = old element n_out + j  (prefix occupies [0,hy))
  __shared__ float2 sh[4096];
  for (int j = tid; j < a.hy; j += 256) sh[j] = base[a.n_out + j];
  __syncthreads();
  for (int j = tid; j < a.hy; j += 256) base[j] = sh[j];
}

// ---- raw-sample history of a decimator: new = last hist_len samples of [old | x]
__global__ __launch_bounds__(256) void hist_roll_kernel(const float2* __restrict__ x,
                                                        const float2* __restrict__ hist_old,
                                                        float2* __restrict__ hist_new, int hist_len,
                                                        uint32_t n_total) {
  for (int j = threadIdx.x; j < hist_len; j += 256) {
    const long long rel = (long long)n_total - hist_len + j;
    hist_new[j] = (rel >= 0) ? x[rel] : hist_old[hist_len + rel];
  }
}

// ---- broadcast FM at the IF rate: polar discriminator (all lanes), then the 19 kHz pilot
// PLL of WFM2 -- inherently serial, one lane per RX, on a 32-bit phase accumulator.
__global__ __launch_bounds__(256) void wfm_disc_kernel(const WfmArgs a) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n1) return;
  const float2 yb = a.y1[r][i], ya = a.y1[r][i - 1];
  const float re = yb.x * ya.x + yb.y * ya.y;
  const float im = yb.y * ya.x - yb.x * ya.y;
  a.w[r][i] = make_float2(atan2f(im, re) * a.scale, 0.f);
}

// 19 kHz pilot PLL of the stereo decoder: an inherently serial recursion (the phase feeds
// back through cos).  One wave per RX walks the IF samples in blocks of 64: the block's mpx
// values are loaded coalesced (lane i = sample i) and broadcast with v_readlane, every lane
// runs the same recursion (ten dependent VALU operations per sample, cos by v_cos_f32 on the
// exact 32-bit phase), lane i keeps the phase of sample i, and the 38 kHz carrier
// sin(2*theta) and the output are then computed by all lanes at once.  No memory access sits
// on the recursion's critical path (the first version loaded and stored per sample and called
// cospif/sinpif: 250 ns per sample; now ~30 ns).
__device__ __forceinline__ void wfm_pll_step(float mj, uint32_t& ph, float& w, const WfmArgs& a) {
  const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
  const float c = __builtin_amdgcn_cosf(rev);
  const float e = __fmul_rn(__fmul_rn(mj, c), a.norm);
  w = __fadd_rn(w, __fmul_rn(a.ki, e));
  const int corr = __float2int_rn(__fmul_rn(__fadd_rn(w, __fmul_rn(a.kp, e)), a.rad2word));
  ph = ph + a.fword0 + (uint32_t)corr;
}

template <bool EMIT>
__device__ __forceinline__ void wfm_pll_walk(const WfmArgs& a, float2* __restrict__ o, int i_begin, int i_end,
                                             uint32_t& ph, float& w, int lane) {
  float m_next = (i_begin + lane < i_end) ? o[i_begin + lane].x : 0.f;
  for (int i0 = i_begin; i0 < i_end; i0 += 64) {
    const float m = m_next;
    const int nidx = i0 + 64 + lane;
    m_next = (nidx < i_end) ? o[nidx].x : 0.f;             // in flight during the 64 steps below
    uint32_t myph = 0u;
    const int count = (i_end - i0 < 64) ? i_end - i0 : 64;
    if (count == 64) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const float mj = lane_bcast(m, j);
        myph = (lane == j) ? ph : myph;
        wfm_pll_step(mj, ph, w, a);
      }
    } else {
      for (int j = 0; j < count; ++j) {
        const float mj = lane_bcast(m, j);
        myph = (lane == j) ? ph : myph;
        wfm_pll_step(mj, ph, w, a);
      }
    }
    if (EMIT && lane < count) {
      const float rev = (float)(int)myph * (1.0f / 4294967296.0f);
      const float s2 = __builtin_amdgcn_sinf(2.f * rev);
      o[i0 + lane] = make_float2(m, __fmul_rn(m, __fmul_rn(2.f, s2)));
    }
  }
}

// 512 words of 2^32 = 7.5e-7 rad of pilot phase (1.5e-6 of the 38 kHz carrier); the integrator
// within 1e-9 rad/sample (x 1/(zeta*wn) = 1900 samples of memory = 2e-6 rad)
__device__ __forceinline__ bool wfm_state_differs(uint32_t ph_a, float w_a, uint32_t ph_b, float w_b) {
  const int d = (int)(ph_a - ph_b);
  return !(d <= 512 && d >= -512 && fabsf(w_a - w_b) <= 1.0e-9f);
}

// grid (K, nrx): segment k of RX r.  The warm-up re-reads mpx values other segments are
// rewriting in place as (mpx, carrier): the .x they read is the same bits before and after.
__global__ __launch_bounds__(64) void wfm_pll_seg_kernel(const WfmArgs a) {
  const int r = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  if (!a.stereo[r]) return;
  const PllPlan& pl = a.pll;
  const RxDevState* st = a.state + r;
  const int n = a.n1;
  const int s0 = k * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
  uint32_t ph = st->wfm_phase;
  float w = st->wfm_w;
  int wb = s0 - pl.W;
  if (k > 0 && wb > 0) {
    // guess: the call's initial state free-running at its own rate up to the warm-up start
    const int corr = __float2int_rn(__fmul_rn(w, a.rad2word));
    ph = ph + (uint32_t)wb * (a.fword0 + (uint32_t)corr);
  } else {
    wb = 0;
  }
  if (wb < s0) wfm_pll_walk<false>(a, a.w[r], wb, s0, ph, w, lane);
  uint32_t* sg = pl.seg + ((size_t)r * pl.K + k) * 4;
  if (lane == 0) { sg[0] = ph; sg[1] = __float_as_uint(w); }
  wfm_pll_walk<true>(a, a.w[r], s0, s1, ph, w, lane);
  if (lane == 0) { sg[2] = ph; sg[3] = __float_as_uint(w); }
}

__global__ __launch_bounds__(64) void wfm_pll_patch_kernel(const WfmArgs a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  // roll the 1-sample IF history for the next call (the discriminator kernel is done: same stream)
  if (lane == 0 && a.n1 > 0) a.y1base[r][1] = a.y1[r][a.n1 - 1];
  if (!a.stereo[r] || a.n1 <= 0) return;
  const PllPlan& pl = a.pll;
  const int K = pl.K, n = a.n1;
  const uint32_t* sg = pl.seg + (size_t)r * K * 4;
  uint32_t ph_fin = sg[(size_t)(K - 1) * 4 + 2];
  float w_fin = __uint_as_float(sg[(size_t)(K - 1) * 4 + 3]);
  int patched = 0;
  int k = 1;
  while (k < K) {
    int bad = K;
    for (int base = k; base < K && bad == K; base += 64) {
      const int kk = base + lane;
      bool mm = false;
      if (kk < K)
        mm = wfm_state_differs(sg[(size_t)(kk - 1) * 4 + 2], __uint_as_float(sg[(size_t)(kk - 1) * 4 + 3]),
                               sg[(size_t)kk * 4 + 0], __uint_as_float(sg[(size_t)kk * 4 + 1]));
      const unsigned long long bal = __ballot(mm);
      if (bal) bad = base + __builtin_ctzll(bal);
    }
    if (bad >= K) break;
    uint32_t ph = sg[(size_t)(bad - 1) * 4 + 2];
    float w = __uint_as_float(sg[(size_t)(bad - 1) * 4 + 3]);
    int j = bad;
    bool joined = false;
    while (j < K) {
      const int s0 = j * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
      wfm_pll_walk<true>(a, a.w[r], s0, s1, ph, w, lane);
      ++patched;
      ++j;
      if (j < K && !wfm_state_differs(ph, w, sg[(size_t)j * 4 + 0], __uint_as_float(sg[(size_t)j * 4 + 1]))) {
        joined = true;
        break;
      }
    }
    if (!joined) { ph_fin = ph; w_fin = w; break; }
    k = j + 1;
  }
  if (lane == 0) {
    RxDevState* st = a.state + r;
    st->wfm_phase = ph_fin;
    st->wfm_w = w_fin;
    st->pll_segments = K;
    st->pll_patched = patched;
  }
}

}  // namespace

int launch_pll(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(am_pll_seg_kernel, dim3(a.pll.K, a.nrx), dim3(64), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(am_pll_patch_kernel, dim3(a.nrx), dim3(64), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_demod_fir(const Stage2Args& a, hipStream_t st) {
  if (a.n_out <= 0) return PYSDR_OK;
  const int ngroups = (a.ntaps + kW - 1) / kW;
  const size_t lds = (size_t)2 * kW * (kFirOut / kW + ngroups) * sizeof(float);
  dim3 grid((a.n_out + kFirOut - 1) / kFirOut, a.nrx);
  hipLaunchKernelGGL(demod_fir_kernel, grid, dim3(kFirThreads), lds, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_agc_scan(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(agc_scan_kernel, dim3(a.nrx), dim3(256), (size_t)a.nchunks * sizeof(float), st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_apply(const Stage2Args& a, hipStream_t st) {
  if (a.n_out <= 0) return PYSDR_OK;
  dim3 grid((a.n_out + 255) / 256, a.nrx);
  hipLaunchKernelGGL(apply_kernel, grid, dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_epilogue(const EpilogueArgs& a, hipStream_t st) {
  if (a.hy > 4096) {
    set_last_error("epilogue: history %d too long", a.hy);
    return PYSDR_ERR_ARG;
  }
  hipLaunchKernelGGL(epilogue_kernel, dim3(2 * a.nrx), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_hist_roll(const float2* x, const float2* hist_old, float2* hist_new, int hist_len,
                     uint32_t n_total, hipStream_t st) {
  hipLaunchKernelGGL(hist_roll_kernel, dim3(1), dim3(256), 0, st, x, hist_old, hist_new, hist_len, n_total);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_wfm(const WfmArgs& a, hipStream_t st) {
  if (a.n1 > 0) {
    hipLaunchKernelGGL(wfm_disc_kernel, dim3((a.n1 + 255) / 256, a.nrx), dim3(256), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
  }
  bool any_stereo = false;
  for (int r = 0; r < a.nrx; ++r) any_stereo |= (a.stereo[r] != 0);
  if (any_stereo && a.n1 > 0) {
    hipLaunchKernelGGL(wfm_pll_seg_kernel, dim3(a.pll.K, a.nrx), dim3(64), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(wfm_pll_patch_kernel, dim3(a.nrx), dim3(64), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
