// Stage 2 of Receiver.demod_data (receiver.py:235) at FS_OUT: per-mode detector
// (rx.demod), AF filter (rx.demod.filter_bank_real/cmpx, receiver.py:873-874), block
// AGC (rx.agc, watchdog.py:298-302) and the history roll that makes chunked ==
// one-shot (sigs/iir.py:83-125).  The data rate here is UP/DOWN (~1/167) of the input
// rate, so these kernels are latency-, not bandwidth-, critical.
#include "common.h"

namespace pysdr {

namespace {

constexpr int kFirTile = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// chunk (AGC block) that output i belongs to: the chunk holding its newest input sample
__device__ __forceinline__ uint32_t block_of(const Stage2Args& a, int i) {
  const uint32_t t = a.t0 + (uint32_t)i * (uint32_t)a.down;
  return (t / (uint32_t)a.up) / a.chunk_len;
}

// ---- AM-Synch carrier PLL (rx.demod.am_pll, receiver.py:649): inherently serial,
// one lane per RX; writes v = y*exp(-j*theta) for the detector stage.
__global__ void pll_kernel(const Stage2Args a) {
  const int r = blockIdx.x;
  if (threadIdx.x != 0 || a.det[r] != kDetPll) return;
  RxDevState* st = a.state + r;
  float th = st->pll_theta, w = st->pll_w;
  const float kp = a.pll_kp, ki = a.pll_ki;
  const float pi = 3.14159265358979323846f, twopi = 6.28318530717958647692f;
  const float2* y = a.y[r];
  float2* o = a.ypll[r];
  for (int i = 0; i < a.n_out; ++i) {
    float s, c;
    sincosf(th, &s, &c);
    const float2 yy = y[i];
    const float vr = yy.x * c + yy.y * s;
    const float vi = yy.y * c - yy.x * s;
    const float e = atan2f(vi, vr);
    w = w + ki * e;
    th = th + (w + kp * e);
    if (th >= pi) th -= twopi;
    else if (th < -pi) th += twopi;
    o[i] = make_float2(vr, vi);
  }
  st->pll_theta = th;
  st->pll_w = w;
}

// ---- detector + AF FIR + block peak.  grid = (tiles, nrx), 256 outputs per workgroup.
__global__ __launch_bounds__(kFirTile) void demod_fir_kernel(const Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const int r = blockIdx.y;
  const int nt = a.ntaps;
  float2* ds = lds;                    // [kFirTile + nt - 1]   detector output d[i0-(nt-1) ..]
  float2* cs = lds + kFirTile + nt - 1;// [nt]
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * kFirTile;
  const int det = a.det[r];
  const float2* y = (det == kDetPll) ? a.ypll[r] : a.y[r];

  for (int k = tid; k < nt; k += kFirTile) cs[k] = a.aftaps[r][k];
  const int nd = kFirTile + nt - 1;
  for (int j = tid; j < nd; j += kFirTile) {
    const int i = i0 - (nt - 1) + j;          // may be negative: history prefix
    float2 d = make_float2(0.f, 0.f);
    if (i < a.n_out) {
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
    }
    ds[j] = d;
  }
  __syncthreads();

  const int i = i0 + tid;
  float2 acc = make_float2(0.f, 0.f);
  const float2* dp = ds + (nt - 1) + tid;
  for (int k = 0; k < nt; ++k) {
    const float2 c = cs[k];
    const float2 d = dp[-k];
    acc.x = fmaf(c.x, d.x, acc.x);
    acc.x = fmaf(-c.y, d.y, acc.x);
    acc.y = fmaf(c.x, d.y, acc.y);
    acc.y = fmaf(c.y, d.x, acc.y);
  }
  float mag = 0.f;
  uint32_t blk = 0xFFFFFFFFu;
  if (i < a.n_out) {
    a.a[r][i] = acc;
    mag = a.out_complex[r] ? sqrtf(acc.x * acc.x + acc.y * acc.y) : fabsf(acc.x);
    blk = block_of(a, i);
  }
  // block peak: one atomic per wave when the wave sits inside one block
  const uint32_t b0 = __shfl(blk, 0);
  const bool uniform = __all(blk == b0 || blk == 0xFFFFFFFFu);
  if (uniform) {
    float m = mag;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0 && b0 != 0xFFFFFFFFu)
      atomicMax(a.blkpeak + (size_t)r * a.nchunks + b0, __float_as_uint(m));
  } else if (blk != 0xFFFFFFFFu) {
    atomicMax(a.blkpeak + (size_t)r * a.nchunks + blk, __float_as_uint(mag));
  }
}

// ---- AGC recursion over the blocks of this call (sigs/agc.m:6-12 loop filter on decay,
// immediate attack).  One workgroup per RX: the block peaks are staged in LDS by all
// lanes, lane 0 runs the serial recursion out of LDS (the only dependent chain is
// cmp/sub/mul/add/select on env), gains go back through LDS and are stored in parallel.
__global__ __launch_bounds__(256) void agc_scan_kernel(const Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) float agc_lds[];
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  float* pk = agc_lds;                 // [nchunks]
  float* gn = agc_lds + a.nchunks;     // [nchunks]
  for (int c = tid; c < a.nchunks; c += 256)
    pk[c] = __uint_as_float(a.blkpeak[(size_t)r * a.nchunks + c]);
  __syncthreads();
  if (tid == 0) {
    RxDevState st = a.state[r];
    const float beta = 0.1f;
    float env = st.env, g = st.gain, peak = st.maxbuf;
    for (int c = 0; c < a.nchunks; ++c) {
      peak = pk[c];
      const float dec = __fadd_rn(env, __fmul_rn(beta, __fsub_rn(peak, env)));
      env = (peak > env) ? peak : dec;
      g = st.agc_enable ? fminf(__fdiv_rn(st.ref, fmaxf(env, 1e-12f)), 1.0e4f) : 1.f;
      gn[c] = g;
    }
    if (a.nchunks > 0) {
      st.env = env; st.gain = g; st.maxbuf = peak;
      st.err = __fsub_rn(st.ref, __fmul_rn(g, peak));
      a.state[r] = st;
    }
  }
  __syncthreads();
  for (int c = tid; c < a.nchunks; c += 256) a.gain[(size_t)r * a.nchunks + c] = gn[c];
}

// ---- apply the block gain, emit rx.am (real, or complex in IQ mode)
__global__ __launch_bounds__(256) void apply_kernel(const Stage2Args a) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_out) return;
  const float g = a.gain[(size_t)r * a.nchunks + block_of(a, i)];
  const float2 v = a.a[r][i];
  if (a.out_complex[r]) {
    reinterpret_cast<float2*>(a.am[r])[i] = make_float2(v.x * g, v.y * g);
  } else {
    a.am[r][i] = v.x * g;
  }
}

// ---- history roll: y prefix <- last hy outputs, x history <- last hist_len samples.
// One workgroup per job so overlapping source/destination ranges are safe.
__global__ __launch_bounds__(256) void epilogue_kernel(const EpilogueArgs a) {
  const int job = blockIdx.x;
  const int tid = threadIdx.x;
  if (job < 2 * a.nrx) {
    float2* base = (job < a.nrx) ? a.ybase[job] : a.ypllbase[job - a.nrx];
    if (base == nullptr) return;
    // element j of the new prefix = old element n_out + j  (prefix occupies [0,hy))
    __shared__ float2 sh[4096];
    for (int j = tid; j < a.hy; j += 256) sh[j] = base[a.n_out + j];
    __syncthreads();
    for (int j = tid; j < a.hy; j += 256) base[j] = sh[j];
  } else {
    for (int j = tid; j < a.hist_len; j += 256) {
      const long long rel = (long long)a.n_total - a.hist_len + j;
      a.hist_new[j] = (rel >= 0) ? a.x[rel] : a.hist_old[a.hist_len + rel];
    }
  }
}

}  // namespace

int launch_pll(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(pll_kernel, dim3(a.nrx), dim3(64), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_demod_fir(const Stage2Args& a, hipStream_t st) {
  if (a.n_out <= 0) return PYSDR_OK;
  const size_t lds = (size_t)(kFirTile + 2 * a.ntaps - 1) * sizeof(float2);
  dim3 grid((a.n_out + kFirTile - 1) / kFirTile, a.nrx);
  hipLaunchKernelGGL(demod_fir_kernel, grid, dim3(kFirTile), lds, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_agc_scan(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(agc_scan_kernel, dim3(a.nrx), dim3(256), (size_t)a.nchunks * 2 * sizeof(float), st, a);
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
  hipLaunchKernelGGL(epilogue_kernel, dim3(2 * a.nrx + 1), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
