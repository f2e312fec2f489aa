// Stage 2 of Receiver.demod_data (receiver.py:235) at FS_OUT: per-mode detector
// (rx.demod), AF filter (rx.demod.filter_bank_real/cmpx, receiver.py:873-874), block
// AGC (rx.agc, watchdog.py:298-302) and the history roll that makes chunked ==
// one-shot (sigs/iir.py:83-125).  The data rate here is UP/DOWN (~1/167) of the input
// rate, so these kernels are latency-, not bandwidth-, critical.
#include "common.h"
#include "hist_roll.h"

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

// first output index behind block b (the outputs of a block are contiguous; every block of a call is whole)
__device__ __forceinline__ int block_end(const Stage2Args& a, int r, uint32_t b) {
  if (a.single_block[r] || b + 1u >= (uint32_t)a.nchunks) return a.n_out;
  const unsigned long long need = (unsigned long long)(b + 1u) * a.chunk_len * (unsigned)a.up;
  if (need <= a.t0) return 0;
  const unsigned long long i = (need - a.t0 + (unsigned)a.down - 1ull) / (unsigned)a.down;
  return i < (unsigned long long)a.n_out ? (int)i : a.n_out;
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// The PLL walks are one dependent chain per wave and a few thousand instructions long: what they cost is latency, not
// issue slots.  When the calls overlap (pysdr_set_overlap) they share their SIMDs with the waves of the NEXT call's mix +
// decimate kernel, and at equal priority the arbiter gave them a slot every ~30 cycles: 105 -> 373 us for the carrier
// loop, 2 x 185 -> 850 us for the pilot loop, LONGER than the front end they were meant to hide behind
// (scripts/diag/overlap_trace.sh).  At a raised priority they keep their own pace and take ~2 % of the slots.
#ifndef PLLX_PRIO
#define PLLX_PRIO 3
#endif
__device__ __forceinline__ void pll_wave_priority() { __builtin_amdgcn_s_setprio(PLLX_PRIO); }

// Inclusive scans over the 64 lanes of a wave (DPP: row_shr 1, 2, 4, 8 inside the rows of 16, then
// row_bcast15 / row_bcast31 carry the row totals up).
__device__ __forceinline__ float wave_scan_add(float v) {
#define PYSDR_DPP_F(ctrl, rm) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rm, 0xF, true))
  v += PYSDR_DPP_F(0x111, 0xF);
  v += PYSDR_DPP_F(0x112, 0xF);
  v += PYSDR_DPP_F(0x114, 0xF);
  v += PYSDR_DPP_F(0x118, 0xF);
  // the two steps across the rows of 16 add in place: rows outside the mask keep v.  Written through update_dpp, hipcc
  // cannot fold "v + (row masked off ? 0 : bcast)" into one DPP add (x + 0 is not x for x = -0) and issues v_mov_b32_dpp +
  // v_add_f32 + a v_mov that zeroes the old value: 6 instructions where these are 2, on the pilot loop's critical chain
  // (2 wait states between a VALU write and its DPP read)
  asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
#undef PYSDR_DPP_F
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
#define PYSDR_DPP_U(ctrl, rm, bc) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xF, bc)
  v += PYSDR_DPP_U(0x111, 0xF, true);
  v += PYSDR_DPP_U(0x112, 0xF, true);
  v += PYSDR_DPP_U(0x114, 0xF, true);
  v += PYSDR_DPP_U(0x118, 0xF, true);
  v += PYSDR_DPP_U(0x142, 0xA, false);
  v += PYSDR_DPP_U(0x143, 0xC, false);
#undef PYSDR_DPP_U
  return v;
}

// ---- AM-Synch carrier PLL (rx.demod.am_pll, receiver.py:649; oracle/sdr_oracle.py CarrierPLL):
//     v = y exp(-j theta), e = atan2(Im v, Re v), w += ki e, theta += w + kp e (wrapped), output Re v.
// e is wrap(arg y - theta): the loop is LINEAR in the phase domain, and that is what this form uses.
//  (i)   phi = arg y is taken once per sample, in parallel, off the chain (am_phase_kernel), as a word of 2^32 per
//        revolution; theta lives on the same 32-bit accumulator, so both wraps are the integer overflow and the sum
//        of phase increments is exact in any order;
//  (ii)  a block of 64 samples (one per lane) is solved by FIXED-POINT SWEEPS like the pilot loop's (wfm_pll_walk):
//        from a guess of the 64 phases every lane takes its own e_j; the integrator after sample j is w0 + ki *
//        (inclusive wave scan of e), the increment rint((w_j + kp e_j) 2^32/2pi), the phase in front of sample j
//        theta0 + (exclusive scan of the increments).  Sample 0 is exact from the start, sweep k makes samples 0..k
//        exact, and the walk stops when a sweep reproduces its input bit for bit -- on the recursion's own
//        trajectory for ANY input (64 sweeps at worst); 64 kp = 0.59, so a locked loop takes 8-10 sweeps of ~20
//        vector instructions where the sample-by-sample walk took 64 x ~100 through sin, cos and atan2
//        (scripts/experiments/am_pll_sweeps.py: the model of these sweeps in NumPy);
//  (iii) parallelism across a call comes from time: segments with a warm-up + a patch-up pass (PllPlan, common.h).
// Against the oracle's float32 walk: theta within 2e-6 rad (the float32 walk rounds theta to 2.4e-7 rad every step and
// wanders by that much around the exact recursion), Re v within 2e-7 of full scale (it depends on theta through
// sin(e) ~ 0).  History: one wave walking sample by sample 193 ns per sample; one LANE per segment (round 3) the same
// chain 64 segments at a time; this form ~10 ns per sample and segment.
constexpr float kRad2Word = 683565275.57643158f;          // 2^32 / 2pi
constexpr float kWord2Rad = 1.4629180792671596e-9f;       // 2pi / 2^32

// arg(x + jy) as a word of 2^32 per revolution: atan of min/max on [0, 1] by an odd polynomial (degree 15, fitted for
// the maximum error: 1.4e-7 rad in float32 arithmetic = the resolution of a float32 angle near pi) already scaled to
// words, the octant put back by exact integer arithmetic.  ~25 instructions where atan2f and the scaling took ~75
// (the kernel below was bound by them: 17 us for 4.2 M samples).
// The word's LSB is a FLAG, "this sample has no amplitude" (x = y = 0: zero-filled replay gaps, a muted or silent input longer
// than the filters): the spec's detector is e = atan2(Im v, Re v) with v = y exp(-j theta) = 0, and atan2(0, 0) = 0 -- the loop
// COASTS on its integrator (oracle CarrierPLL) -- where wrap(arg y - theta) with arg 0 := 0 would pull theta to 0 and leave
// another loop state behind the gap.  A live sample's word has the bit cleared: 2^-31 revolutions of resolution.
__device__ __forceinline__ uint32_t phase_word(float x, float y) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float q = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
  const float t = q * q;
  float p = -0.004054564982652664f * kRad2Word;
  p = __fmaf_rn(p, t, 0.021862950176000595f * kRad2Word);
  p = __fmaf_rn(p, t, -0.0559123158454895f * kRad2Word);
  p = __fmaf_rn(p, t, 0.09642196446657181f * kRad2Word);
  p = __fmaf_rn(p, t, -0.1390862911939621f * kRad2Word);
  p = __fmaf_rn(p, t, 0.19946566224098206f * kRad2Word);
  p = __fmaf_rn(p, t, -0.33329859375953674f * kRad2Word);
  p = __fmaf_rn(p, t, 0.9999993443489075f * kRad2Word);
  int a = __float2int_rn(p * q);                            // [0, 2^29]
  if (ay > ax) a = (1 << 30) - a;
  if (x < 0.f) a = (int)(0x80000000u - (uint32_t)a);
  return ((uint32_t)(y < 0.f ? -a : a) & ~1u) | (mx > 0.f ? 0u : 1u);
}
// the detector on phase words: wrap(phi - theta) (the wrap is the integer overflow), 0 for a sample without amplitude
__device__ __forceinline__ float am_detector(uint32_t phi, uint32_t theta) {
  return (phi & 1u) ? 0.f : (float)(int)(phi - theta);
}

// grid (ceil(n / 2048), nrx): the phase word of every new sample, into the .y of the PLL buffer (its .x gets Re v).
// Eight samples per thread, 256 apart, all eight loads in flight before the first is used (one sample per thread: 16 k
// workgroups of one load each, 17 us for 4.2 M samples = 3 TB/s of 12 bytes per sample).
constexpr int kPhasePer = 8;
__global__ __launch_bounds__(256) void am_phase_kernel(const Stage2Args a) {
  const int r = blockIdx.y, i0 = blockIdx.x * (256 * kPhasePer) + threadIdx.x;
  if (a.det[r] != kDetPll) return;
  float2 v[kPhasePer];
#pragma unroll
  for (int j = 0; j < kPhasePer; ++j) {
    const int i = i0 + 256 * j;
    v[j] = i < a.n_out ? a.y[r][i] : make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < kPhasePer; ++j) {
    const int i = i0 + 256 * j;
    if (i < a.n_out) reinterpret_cast<uint32_t*>(a.ypll[r] + i)[1] = phase_word(v[j].x, v[j].y);
  }
}

struct M2 { double a, b, c, d; };
__device__ __forceinline__ M2 m2_mul(const M2& x, const M2& y) {
  return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

// ---- A block's first guess by a DIRECT LINEAR SOLVE (round 5, late).  Around the line g[j] = theta0 + j G (G = the
// integrator's rate rounded to words) with u[j] = wrap(phi[j] - g[j]), eps = theta - g, V = integrator - G, the 64 samples
// of a block obey x[j+1] = A x[j] + b[j] on x = (eps, V), A = [[1 - kp - ki, 1], [-ki, 1]], b = ((kp + ki) u, ki u) while
// the detector does not wrap: an inclusive wave scan with CONSTANT matrices -- four row_shr steps with A, A^2, A^4, A^8,
// the two row broadcasts with per-lane powers A^((j & 15) + 1), A^((j & 31) + 1) -- yields all 64 phases at once, in float32
// (a guess: within a word or two of the recursion's own rounding).  ~45 instructions; the sweeps that follow then only
// CONFIRM it (2-3 until one reproduces its input, where the free-running guess needed 8-10), and they still repair it
// sample by sample where the detector did wrap, so the walk ends on the recursion's own trajectory for any input as before.
struct AmBlk {
  float a1[4], a2[4], a4[4], a8[4];          // A^1, A^2, A^4, A^8 (uniform)
  float m1[4], m2[4];                        // A^((lane & 15) + 1), A^((lane & 31) + 1)
  float c0;                                  // (A^(lane + 1))[0][1]: what V in front of the block adds to eps behind sample `lane`
  float kpi;
};
__device__ __forceinline__ void m2_to_f(const M2& m, float (&f)[4]) { f[0] = (float)m.a; f[1] = (float)m.b; f[2] = (float)m.c; f[3] = (float)m.d; }
__device__ __forceinline__ M2 m2_pow(const M2& A, int e) {      // e < 128
  M2 sq = A, p = M2{1.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int b = 0; b < 7; ++b) {
    if ((e >> b) & 1) p = m2_mul(p, sq);
    sq = m2_mul(sq, sq);
  }
  return p;
}
__device__ __forceinline__ AmBlk am_block_consts(const Stage2Args& a, int lane) {
  const double kp = (double)a.pll_kp, ki = (double)a.pll_ki;
  const M2 A = M2{1.0 - kp - ki, 1.0, -ki, 1.0};
  AmBlk k;
  const M2 A2 = m2_mul(A, A), A4 = m2_mul(A2, A2), A8 = m2_mul(A4, A4);
  m2_to_f(A, k.a1); m2_to_f(A2, k.a2); m2_to_f(A4, k.a4); m2_to_f(A8, k.a8);
  m2_to_f(m2_pow(A, (lane & 15) + 1), k.m1);
  m2_to_f(m2_pow(A, (lane & 31) + 1), k.m2);
  k.c0 = (float)m2_pow(A, lane + 1).b;
  k.kpi = (float)(kp + ki);
  return k;
}
template <int CTRL, int ROWS, bool BC>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, BC));
}

// The loop over [i_begin, i_end) from the state (ph0, w0) in front of sample i_begin; 64 samples per block.
// max_it <= 8: a COARSE walk for the early part of a warm-up -- that many sweeps per block, no questions asked (what
// they leave behind, ~0.6^s / s! of the first guess's error, is forgotten by the exact tail of the warm-up);
// otherwise sweeps until one reproduces its input.  Lanes past the end of a last, partial block compute on zeros:
// the scans only carry sums upwards, so nothing of theirs reaches a live lane.
#ifndef AMX_DEPTH
#define AMX_DEPTH 4
#endif
constexpr int kAmDepth = AMX_DEPTH;                           // blocks per group of loads
template <bool EMIT>
__device__ __forceinline__ void am_pll_walk(const Stage2Args& a, int r, int i_begin, int i_end, uint32_t& ph0,
                                            float& w0, int lane, const AmBlk& kc, int max_it = 66) {
  typedef float pl_v2f __attribute__((ext_vector_type(2)));
  const float2* __restrict__ y = a.y[r];
  float2* __restrict__ o = a.ypll[r];
  const float kp = a.pll_kp, ki = a.pll_ki;
  // Loads run a GROUP of kAmDepth blocks ahead: at the top of a group the phase words (and, where outputs are due, the
  // samples) of the whole next group are requested, and they are first touched by the copy at the group's end, which is
  // where hipcc then puts its s_waitcnt -- kAmDepth blocks of sweeps after the loads were issued.  History: one block
  // ahead through inline asm with a hand-placed wait (a load issued at the top of block i and used at the top of block
  // i + 1 made hipcc wait for it in front of the first sweep of block i: it does not count across the back edge); that was
  // enough while a block took 8-10 sweeps; with the direct solve a block is ~130 instructions, shorter than the latency
  // of a load (74 us per call: 32 blocks x ~2 us).  Several asm loads in flight over an unrolled loop with exits did NOT
  // survive register allocation: copies of registers still in flight on the exit edges, found by the control-flow check
  // of tests/test_isa_checks.py after the GPU had faulted.  Past the end of the range the index is clamped to the last
  // sample (lanes past the end of a last, partial block thus compute on a repeated sample: the scans only carry sums
  // upwards, so nothing of theirs reaches a live lane).
  constexpr int D = kAmDepth;
  if (i_begin >= i_end) return;
  uint32_t cf[D], nf[D];
  pl_v2f cy[D], ny[D];
  const uint32_t* __restrict__ fo = reinterpret_cast<const uint32_t*>(o);
  const pl_v2f* __restrict__ yl = reinterpret_cast<const pl_v2f*>(y);
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    int idx = i_begin + 64 * sl + lane;
    idx = idx < i_end ? idx : i_end - 1;
    cf[sl] = fo[2 * (size_t)idx + 1];
    cy[sl] = EMIT ? yl[idx] : (pl_v2f){0.f, 0.f};
  }
  for (int ib = i_begin; ib < i_end; ib += 64 * D) {
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    int idx = ib + 64 * (D + sl) + lane;
    idx = idx < i_end ? idx : i_end - 1;
    nf[sl] = fo[2 * (size_t)idx + 1];
    ny[sl] = EMIT ? yl[idx] : (pl_v2f){0.f, 0.f};
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    const int i0 = ib + 64 * sl;
    if (i0 >= i_end) break;
    const uint32_t phi = cf[sl];
    const pl_v2f yv = cy[sl];
    const int count = (i_end - i0 < 64) ? i_end - i0 : 64;
    // in words of 2^32: e and its scan as floats, the integrator's rate w0 R once per block
    const float w0r = __fmul_rn(w0, kRad2Word);
    const int g0 = __float2int_rn(w0r);
    uint32_t ph = ph0 + (uint32_t)lane * (uint32_t)g0;     // guess: free running at the integrator's rate ...
    if (a.pll.direct) {                                    // ... corrected by the linear solve of the block around that line
      const float u = am_detector(phi, ph);
      float s0 = kc.kpi * u, s1 = ki * u;
#define PYSDR_AM_STEP(CTRL, ROWS, BC, M)                                                      \
      {                                                                                       \
        const float t0 = dpp_f<CTRL, ROWS, BC>(s0), t1 = dpp_f<CTRL, ROWS, BC>(s1);           \
        const float n0 = __fmaf_rn(M[0], t0, __fmaf_rn(M[1], t1, s0));                        \
        const float n1 = __fmaf_rn(M[2], t0, __fmaf_rn(M[3], t1, s1));                        \
        s0 = n0; s1 = n1;                                                                     \
      }
      PYSDR_AM_STEP(0x111, 0xF, true, kc.a1)
      PYSDR_AM_STEP(0x112, 0xF, true, kc.a2)
      PYSDR_AM_STEP(0x114, 0xF, true, kc.a4)
      PYSDR_AM_STEP(0x118, 0xF, true, kc.a8)
      PYSDR_AM_STEP(0x142, 0xA, false, kc.m1)
      PYSDR_AM_STEP(0x143, 0xC, false, kc.m2)
#undef PYSDR_AM_STEP
      s0 = __fmaf_rn(kc.c0, w0r - (float)g0, s0);          // eps behind sample `lane` ...
      const float eps = dpp_f<0x138, 0xF, false>(s0);      // ... is eps in front of the next (wave_shr:1; lane 0 keeps 0)
      ph += (uint32_t)__float2int_rn(eps);
    }
    uint32_t tot = 0u;
    float sj = 0.f;
    auto sweep = [&](uint32_t pin) -> uint32_t {
      const float e = am_detector(phi, pin);               // the detector: wrap(phi - theta), the wrap is the overflow
      sj = wave_scan_add(e);
      const uint32_t corr = (uint32_t)__float2int_rn(__fmaf_rn(kp, e, __fmaf_rn(ki, sj, w0r)));
      tot = wave_scan_add(corr);
      return ph0 + tot - corr;
    };
    // two sweeps per trip: the phases alternate between two registers instead of being copied back every sweep
    if (max_it <= 8) {
      for (int it = 0;; it += 2) {
        const uint32_t p1 = sweep(ph);
        if (it + 1 >= max_it) { ph = p1; break; }
        ph = sweep(p1);
        if (it + 2 >= max_it) break;
      }
    } else {
      const unsigned long long mine = count >= 64 ? ~0ull : (1ull << count) - 1ull;
      for (int it = 0;;) {
        const uint32_t p1 = sweep(ph);
        if (!(__ballot(p1 != ph) & mine) || ++it >= max_it) { ph = p1; break; }
        ph = sweep(p1);
        if (!(__ballot(ph != p1) & mine) || ++it >= max_it) break;
      }
    }
    if (EMIT && lane < count) {
      const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
      const float s = __builtin_amdgcn_sinf(rev), c = __builtin_amdgcn_cosf(rev);
      // only Re v is stored (the detector stage reads nothing else), into .x: the warm-ups of neighbouring segments
      // are reading the .y words meanwhile -- no location is both read and written here
      reinterpret_cast<float*>(o + i0 + lane)[0] = yv.x * c + yv.y * s;
    }
    ph0 = ph0 + (uint32_t)__builtin_amdgcn_readlane((int)tot, count - 1);
    w0 = __fmaf_rn(ki * kWord2Rad, lane_bcast(sj, count - 1), w0);
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) { cf[sl] = nf[sl]; cy[sl] = ny[sl]; }
  }
}

// 1024 words of 2^32 = 1.5e-6 rad of carrier phase (x sin(e) ~ 0 in the output); the integrator within 2e-8 rad/sample.
// Two walks of one record from different starts end a few words apart once both have converged (measured 7-40 at 16
// time constants, scripts/experiments/am_pll_sweeps.py): the blocks of the two sum the integrator in different places.
__device__ __forceinline__ bool am_state_differs(uint32_t ph_a, float w_a, uint32_t ph_b, float w_b) {
  const int d = (int)(ph_a - ph_b);
  return !(d <= 1024 && d >= -1024 && fabsf(w_a - w_b) <= 2.0e-8f);
}

// ---- The warm-up as ONE LINEAR SOLVE (round 5; scripts/experiments/am_linear_seed.py).  Around the straight line
// g[j] = a + j G over the window (a, G: the guess below), with u[j] = wrap(phi[j] - g[j]) taken per sample, eps = theta - g
// and V = integrator - G in words, the recursion is, while the detector does not wrap,
//     eps[j+1] = (1 - kp - ki) eps[j] + V[j-1] + (kp + ki) u[j],      V[j] = V[j-1] - ki eps[j] + ki u[j]
// -- a CONSTANT matrix A and an input, so the state behind N samples is sum_k A^(N-1-k) b[k] (+ A^N x[0], which 16 time
// constants forget like the walk does).  Lane l takes the samples 64 j + l of the window (coalesced, all loads
// independent of the arithmetic), Horner in A^64 (4 fp64 FMAs per sample), then A^(63-l) and a wave sum: ~12
// instructions per 64 samples where the walk spends 4-10 sweeps of 21 DEPENDENT ones.  fp64: eps is up to 2^30 words and
// must come out to the word.  The detector does not wrap, provably, while |u| <= 0.2 revolutions throughout (eps follows u
// with < 20 % overshoot: |u - eps| < 0.44): that is checked, and a window that fails it -- noise in the troughs of deep
// modulation, a carrier more than ~2.7 Hz from the integrator's guess -- is walked as before.  Result against the exact
// walk on a clean carrier: 8 words of 2^32 (model), the integrator 1e-10.  It is a START, like the walked warm-up's: the
// joins are held to the same tolerance by the patch kernel.
constexpr int kAmLinearMaxU = 858993459;                    // 0.2 revolutions

// the state in front of sample s0 from the window [wb, s0) (multiples of 64 apart) and the line (anchor a, slope G);
// false: the window left the linear range
// exact: (anchor, v0) IS the loop's state in front of sample wb (the call's own start: eps = 0, V = v0) and its term
// A^N x[0] is added, instead of a guess that N samples are trusted to forget -- the first segments of a call, whose window
// would begin before it (walked from the true state they cost 2 T samples where every other segment walks T: segment 1
// was the kernel's longest wave, 64 blocks against 32 + this solve).
__device__ __forceinline__ bool am_linear_start(const Stage2Args& a, int r, int wb, int s0, uint32_t anchor, uint32_t G,
                                                int lane, uint32_t& ph, float& w, bool exact = false, double v0 = 0.0) {
  const float2* __restrict__ o = a.ypll[r];
  const double kp = (double)a.pll_kp, ki = (double)a.pll_ki, kpi = kp + ki;
  // A^64 for the rows, A^(63 - lane) for this lane's place in a row (powers of one matrix commute)
  M2 sq = M2{1.0 - kp - ki, 1.0, -ki, 1.0}, pl = M2{1.0, 0.0, 0.0, 1.0};
  const int ex = 63 - lane;
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    if ((ex >> b) & 1) pl = m2_mul(pl, sq);
    sq = m2_mul(sq, sq);
  }
  const M2 A64 = sq;
  const int m = (s0 - wb) >> 6;
  double c0 = 0.0, c1 = 0.0;
  int umax = 0;
  uint32_t dead = 0u;
  uint32_t gl = anchor + (uint32_t)lane * G;                // the line at this lane's sample of row j
  const uint32_t* f = reinterpret_cast<const uint32_t*>(o + wb + lane) + 1;
#pragma unroll 6
  for (int j = 0; j < m; ++j) {
    const uint32_t fj = f[128 * j];
    const int u = (int)(fj - gl);
    gl += 64u * G;
    dead |= fj & 1u;                                        // a sample without amplitude: e = 0 there, not u - eps -- not this model's case
    umax = max(umax, u < 0 ? -u : u);                       // (INT_MIN stays negative: it fails the test below as it should)
    const double ud = (double)u;
    const double n0 = fma(A64.a, c0, fma(A64.b, c1, kpi * ud));
    const double n1 = fma(A64.c, c0, fma(A64.d, c1, ki * ud));
    c0 = n0; c1 = n1;
  }
  double x0 = pl.a * c0 + pl.b * c1, x1 = pl.c * c0 + pl.d * c1;
  bool lin = umax >= 0 && umax <= kAmLinearMaxU && !dead;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { x0 += __shfl_xor(x0, d); x1 += __shfl_xor(x1, d); }
  if (__ballot(!lin)) return false;
  if (exact) {
    M2 sqn = M2{1.0 - kp - ki, 1.0, -ki, 1.0}, an = M2{1.0, 0.0, 0.0, 1.0};
    for (int e = s0 - wb; e > 0; e >>= 1) {
      if (e & 1) an = m2_mul(an, sqn);
      sqn = m2_mul(sqn, sqn);
    }
    x0 += an.b * v0;
    x1 += an.d * v0;
  }
  ph = anchor + (uint32_t)(s0 - wb) * G + (uint32_t)(int)__double2ll_rn(x0);
  w = (float)(((double)(int)G + x1) * (6.283185307179586476925 / 4294967296.0));
  return true;
}

// grid (K, nrx): segment k of RX r.  Segment k > 0 starts W samples early from a guess: the integrator as the call
// began, the phase the block mean of the signal's own (the loop sits on the carrier: off by the noise of 64 samples
// instead of up to pi, which saves ~4 time constants of warm-up).
__global__ __launch_bounds__(64) void am_pll_seg_kernel(const Stage2Args a) {
  const int r = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  if (a.det[r] != kDetPll) return;
  pll_wave_priority();
  const PllPlan& pl = a.pll;
  RxDevState* st = a.state + r;
  const int n = a.n_out;
  const int s0 = k * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
  const AmBlk kc = am_block_consts(a, lane);
  uint32_t ph = st->pll_phase;
  float w = st->pll_w;
  int wb = s0 - pl.W;
  if (k > 0 && wb > 0) {
    const uint32_t* f = reinterpret_cast<const uint32_t*>(a.ypll[r] + wb + lane) + 1;      // wb + 64 <= s0 <= n
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)*f);
    const uint32_t inc0 = (uint32_t)__float2int_rn(__fmul_rn(w, kRad2Word));
    const float dev = (float)(int)(*f - f0 - (uint32_t)lane * inc0);
    const float mean = lane_bcast(wave_scan_add(dev), 63) * (1.0f / 64.0f);
    ph = f0 + (uint32_t)__float2int_rn(mean);
  } else {
    wb = 0;                                  // the true state of the call: exact, however short
  }
  bool linear = false;
  if (k > 0 && pl.seeded && s0 - pl.Wseed - wb >= 64) {
    const float w0r = __fmul_rn(w, kRad2Word);
    const int g0 = __float2int_rn(w0r);
    const uint32_t inc0 = (uint32_t)g0;
    uint32_t ph_l = 0u; float w_l = 0.f;
    // the window up to s0 - Wseed, then Wseed samples of the exact walk (default 0); from the call's own state where the
    // window would begin before the call
    if (am_linear_start(a, r, wb, s0 - pl.Wseed, ph, inc0, lane, ph_l, w_l, wb == 0, (double)w0r - (double)g0)) {
      linear = true;
      ph = ph_l; w = w_l;
      wb = s0 - pl.Wseed;
    }
  }
  if (wb < s0) {
    // coarse sweeps first, the last Wexact samples to the fixed point (both bounds on multiples of 64)
    const int sx = (!linear && pl.coarse_sweeps > 0 && s0 - pl.Wexact > wb) ? s0 - pl.Wexact : wb;
    if (wb < sx) am_pll_walk<false>(a, r, wb, sx, ph, w, lane, kc, pl.coarse_sweeps);
    am_pll_walk<false>(a, r, sx, s0, ph, w, lane, kc);
  }
  uint32_t* sg = pl.seg + ((size_t)r * pl.K + k) * 4;
  if (lane == 0) { sg[0] = ph; sg[1] = __float_as_uint(w); pl.lin[(size_t)r * pl.K + k] = linear ? 1u : 0u; }
  am_pll_walk<true>(a, r, s0, s1, ph, w, lane, kc);
  if (lane == 0) {
    sg[2] = ph; sg[3] = __float_as_uint(w);
    if (pl.K == 1) {                         // every live call: nothing to join, no patch-up launch
      st->pll_phase = ph; st->pll_w = w;
      st->pll_segments = 1; st->pll_patched = 0;
      st->pll_join_words = 0; st->pll_join_dw = 0.f;
      st->pll_linear = 0;
    }
  }
}

// grid (nrx): walk the chain of segments, redo what does not join up; the widest join for pysdr_pll_join_margin
__global__ __launch_bounds__(64) void am_pll_patch_kernel(const Stage2Args a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (a.det[r] != kDetPll) return;
  const PllPlan& pl = a.pll;
  const int K = pl.K, n = a.n_out;
  const AmBlk kc = am_block_consts(a, lane);
  const uint32_t* sg = pl.seg + (size_t)r * K * 4;
  uint32_t ph_fin = sg[(size_t)(K - 1) * 4 + 2];
  float w_fin = __uint_as_float(sg[(size_t)(K - 1) * 4 + 3]);
  int patched = 0, jw = 0;
  float jd = 0.f;
  int k = 1;
  bool first = true;                                           // the margin is that of the first pass over the joins
  while (k < K) {
    int bad = K;
    for (int base = k; base < K && (bad == K || first); base += 512) {   // eight joins per lane in flight
      uint2 e[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        e[u] = b[u] = make_uint2(0u, 0u);
        if (kk < K) {
          e[u] = *reinterpret_cast<const uint2*>(sg + (size_t)(kk - 1) * 4 + 2);
          b[u] = *reinterpret_cast<const uint2*>(sg + (size_t)kk * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        const bool mm = kk < K && am_state_differs(e[u].x, __uint_as_float(e[u].y), b[u].x, __uint_as_float(b[u].y));
        const unsigned long long bal = __ballot(mm);
        if (bal && bad == K) bad = base + 64 * u + __builtin_ctzll(bal);
        if (first && kk < K) {
          const int d = (int)(e[u].x - b[u].x);
          jw = max(jw, d < 0 ? -d : d);
          jd = fmaxf(jd, fabsf(__uint_as_float(e[u].y) - __uint_as_float(b[u].y)));
        }
      }
    }
    first = false;
    if (bad >= K) break;
    uint32_t ph = sg[(size_t)(bad - 1) * 4 + 2];
    float w = __uint_as_float(sg[(size_t)(bad - 1) * 4 + 3]);
    int j = bad;
    bool joined = false;
    while (j < K) {
      const int s0 = j * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
      am_pll_walk<true>(a, r, s0, s1, ph, w, lane, kc);
      ++patched;
      ++j;
      if (j < K && !am_state_differs(ph, w, sg[(size_t)j * 4 + 0], __uint_as_float(sg[(size_t)j * 4 + 1]))) {
        joined = true;                        // segment j was started from (nearly) this state: it stands
        break;
      }
    }
    if (!joined) { ph_fin = ph; w_fin = w; break; }
    k = j + 1;
  }
  int nlin = 0;
  for (int base = 0; base < K; base += 512) {                 // eight independent loads per lane
    uint32_t fl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int kk = base + 64 * u + lane; fl[u] = kk < K ? pl.lin[(size_t)r * K + kk] : 0u; }
#pragma unroll
    for (int u = 0; u < 8; ++u) nlin += (int)fl[u];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { jw = max(jw, __shfl_xor(jw, o)); jd = fmaxf(jd, __shfl_xor(jd, o)); nlin += __shfl_xor(nlin, o); }
  if (lane == 0) {
    RxDevState* st = a.state + r;
    st->pll_phase = ph_fin;
    st->pll_w = w_fin;
    st->pll_segments = K;
    st->pll_patched = patched;
    st->pll_join_words = jw;
    st->pll_join_dw = jd;
    st->pll_linear = nlin;                     // segments that started from the linear solve
  }
}

// ---- detector + AF FIR + block peak.  grid = (tiles, RX of this launch), 2048 outputs per
// workgroup, EIGHT consecutive outputs per thread.  The detector output d is staged in LDS as two
// float arrays (re, im) with S[e + 3] = d[e]: a thread's outputs 8t .. 8t+7 need, for the four taps
// 4m .. 4m+3, the eleven values S[8t - 4m .. 8t - 4m + 10] -- three 16-byte chunks, of which only
// ONE is new per block of four taps.  So four taps cost one ds_read_b128 per array for the data, two
// broadcast ds_read_b128 per array for the taps (all lanes the same address) and 16 / 32 / 64 packed
// FMAs (real x real, Re(c*d), complex) -- see fir_block.  This file is compiled with
// -fno-slp-vectorize (hipcc's own v_pk pairs come with v_mov shuffles).
// A chunk read has lanes 32 bytes apart; 4 floats of padding after every 128 keep the sixteen
// lanes the LDS serves per cycle on sixteen different bank quads.
// Where the time goes (C3, 4 RX x 2.1 M outputs, ablations by rocprofv3, scripts/diag/stage2_kt.sh):
// 165 us with one dword of block peak per block side by side (the per-wave atomicMax of 32 blocks
// hit ONE 128-byte line and serialise: 60 us), 115 us with the accumulators 256 bytes apart
// (kBlkStride); of those the inner product is 46-63 us (114 as plain v_fmac), staging the tile
// (loads from HBM + detector) 44, the remaining atomics 21, the stores 3 (16-byte stores; 23 as
// eight strided dwords) -- the phases of the workgroups of a CU run in step and barely overlap.
constexpr int kW = 8;                  // outputs per thread
#ifndef FIRX_THREADS
#define FIRX_THREADS 256               // A/B: threads (x 8 outputs) per workgroup of the AF FIR
#endif
constexpr int kFirThreads = FIRX_THREADS;
constexpr int kFirOut = kW * kFirThreads;   // outputs per workgroup
constexpr int kFirRealReal = 0;        // d real, c real    -> real
constexpr int kFirRePart = 1;          // d cplx, c cplx    -> Re(c*d)
constexpr int kFirCplx = 2;            // d cplx, c cplx    -> cplx (IQ)
__host__ __device__ __forceinline__ constexpr int fir_pad(int e) { return e + 4 * (e >> 7); }
__host__ __device__ __forceinline__ constexpr int fir_taps_padded(int ntaps) { return (ntaps + 11) / 12 * 12; }

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

// ---- the inner product in packed FMAs.  v_pk_fma_f32 does two FMAs for 5.9 issue cycles where
// v_fmac_f32 does one for 5.4 (scripts/experiments/valu_rate.hip), but its three operands are
// 64-bit register PAIRS on even registers -- pairing two OUTPUTS of a lane needs, for every other
// tap, a data pair that starts on an odd register (hipcc then shuffles with v_mov and the gain is
// gone: round 1, and the first round-2 attempt).  So the pair runs over two TAPS of one output:
//     acc2[j] += (c[k] * S[e_j - k], c[k+1] * S[e_j - k - 1])
// whose data pair is the aligned one when e_j - k - 1 is even: even outputs pair the taps
// (0,1)(2,3).. and odd outputs (1,2)(3,4).., read from a second copy of the taps shifted by one
// (tap 0 of the odd outputs is their accumulator's initial value, tap H of the copy is zero).  The
// data pair is used crosswise (op_sel: low half x high half), the two halves of acc2 are added at
// the end.  Per block of four taps and 8 outputs: 16 / 32 / 64 packed FMAs where the scalar form
// issued 32 / 64 / 128.
typedef float v2f __attribute__((ext_vector_type(2)));
// acc += (c.lo * w.hi, c.hi * w.lo)
__device__ __forceinline__ void pk_fma_x(v2f& acc, const v2f c, const v2f w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(c), "v"(w));
}
// acc -= (c.lo * w.hi, c.hi * w.lo)
__device__ __forceinline__ void pk_fms_x(v2f& acc, const v2f c, const v2f w) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
      : "+v"(acc) : "v"(c), "v"(w));
}

// One block of four taps for the 8 outputs of a lane.  wr/wi[6] = the window of three 16-byte
// chunks as six pairs; ROT = block index mod 3 says which physical chunk is logical chunk 0 (the
// lowest addresses).  a4 = taps 4m .. 4m+3, b4 = taps 4m+1 .. 4m+4 (re / im).
template <int KIND, int ROT>
__device__ __forceinline__ void fir_block(const float4 ar4, const float4 ai4, const float4 br4, const float4 bi4,
                                          const v2f (&wr)[6], const v2f (&wi)[6], v2f (&ax)[kW], v2f (&ay)[kW]) {
  const v2f tr[2][2] = {{{ar4.x, ar4.y}, {ar4.z, ar4.w}}, {{br4.x, br4.y}, {br4.z, br4.w}}};
  const v2f ti[2][2] = {{{ai4.x, ai4.y}, {ai4.z, ai4.w}}, {{bi4.x, bi4.y}, {bi4.z, bi4.w}}};
#pragma unroll
  for (int j = 0; j < kW; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // logical pair holding the window positions of this tap pair (see above)
      const int lp = (j & 1) ? (h == 0 ? (1 + j) / 2 : (j - 1) / 2) : (h == 0 ? (2 + j) / 2 : j / 2);
      const int pp = (((lp >> 1) + 3 - ROT) % 3) * 2 + (lp & 1);       // physical pair
      const v2f cr = tr[j & 1][h], ci = ti[j & 1][h];
      pk_fma_x(ax[j], cr, wr[pp]);
      if (KIND != kFirRealReal) {
        pk_fms_x(ax[j], ci, wi[pp]);
        if (KIND == kFirCplx) {
          pk_fma_x(ay[j], cr, wi[pp]);
          pk_fma_x(ay[j], ci, wr[pp]);
        }
      }
    }
}

template <int KIND>
__device__ __forceinline__ void fir_run(const float* sre, const float* sim, const float* tre, const float* tim,
                                        const float* tre_s, const float* tim_s, int nblk, int H, int tid,
                                        float2 (&acc)[kW]) {
  v2f wr[6], wi[6], ax[kW], ay[kW];
  int e0 = kW * tid + H;                                       // logical chunk 0 of block 0
  auto chunk = [&](const float* s, int e) { return *reinterpret_cast<const float4*>(s + fir_pad(e)); };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float4 v = chunk(sre, e0 + 4 * c);
    wr[2 * c] = (v2f){v.x, v.y}; wr[2 * c + 1] = (v2f){v.z, v.w};
    if (KIND != kFirRealReal) {
      const float4 u = chunk(sim, e0 + 4 * c);
      wi[2 * c] = (v2f){u.x, u.y}; wi[2 * c + 1] = (v2f){u.z, u.w};
    } else {
      wi[2 * c] = wi[2 * c + 1] = (v2f){0.f, 0.f};
    }
  }
  // tap 0 of the odd outputs (their pairs start at tap 1): window position 3 + j of block 0
  {
    const float c0r = tre[0], c0i = (KIND == kFirRealReal) ? 0.f : tim[0];
#pragma unroll
    for (int j = 0; j < kW; ++j) {
      ax[j] = (v2f){0.f, 0.f};
      ay[j] = (v2f){0.f, 0.f};
      if (j & 1) {
        const int pos = 3 + j;                                  // even: the low half of pair pos / 2
        const float dr = wr[pos >> 1].x, di = wi[pos >> 1].x;
        ax[j].x = (KIND == kFirRealReal) ? c0r * dr : fmaf(c0r, dr, -c0i * di);
        if (KIND == kFirCplx) ay[j].x = fmaf(c0r, di, c0i * dr);
      }
    }
  }
  // nblk is a multiple of 3: the register roles repeat every three blocks
  for (int m = 0; m < nblk; m += 3) {
#pragma unroll
    for (int rot = 0; rot < 3; ++rot) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 ar4 = *reinterpret_cast<const float4*>(tre + 4 * (m + rot));
      const float4 br4 = *reinterpret_cast<const float4*>(tre_s + 4 * (m + rot));
      const float4 ai4 = (KIND == kFirRealReal) ? z4 : *reinterpret_cast<const float4*>(tim + 4 * (m + rot));
      const float4 bi4 = (KIND == kFirRealReal) ? z4 : *reinterpret_cast<const float4*>(tim_s + 4 * (m + rot));
      // the chunk the NEXT block adds below the window (never read past the staged history:
      // the last block's successor is e0 - 4 >= 0 and simply goes unused)
      const float4 nr = chunk(sre, e0 - 4);
      const float4 ni = (KIND == kFirRealReal) ? z4 : chunk(sim, e0 - 4);
      if (rot == 0) fir_block<KIND, 0>(ar4, ai4, br4, bi4, wr, wi, ax, ay);
      else if (rot == 1) fir_block<KIND, 1>(ar4, ai4, br4, bi4, wr, wi, ax, ay);
      else fir_block<KIND, 2>(ar4, ai4, br4, bi4, wr, wi, ax, ay);
      // it replaces this block's logical chunk 2 = physical chunk (2 - rot) mod 3
      const int pc = (5 - rot) % 3 * 2;
      wr[pc] = (v2f){nr.x, nr.y}; wr[pc + 1] = (v2f){nr.z, nr.w};
      if (KIND != kFirRealReal) { wi[pc] = (v2f){ni.x, ni.y}; wi[pc + 1] = (v2f){ni.z, ni.w}; }
      e0 -= 4;
    }
  }
#pragma unroll
  for (int j = 0; j < kW; ++j) acc[j] = make_float2(ax[j].x + ax[j].y, ay[j].x + ay[j].y);
}

// fixed-point form of one term of the squelch's block sums, and its inverse (see fir_epilogue): 2^48 per unit -- terms are weights
// <= 1e-3 times |detector output| <= ~10, a block has <= ~8k of them: the sum stays below 2^63 with room to spare
constexpr double kSqFix = 281474976710656.0;
__device__ __forceinline__ unsigned long long sq_fix(float v) { return __double2ull_rn((double)v * kSqFix); }
__device__ __forceinline__ float sq_unfix(unsigned long long q) { return (float)((double)q * (1.0 / kSqFix)); }
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
  return ((unsigned long long)hi << 32) | lo;
}

// What both forms of the kernel (packed FMAs below, matrix cores further down) do with a thread's eight consecutive outputs
// acc[0..8) = outputs i0 + 8 tid ..: store them, and fold their magnitudes into the block peaks (and the squelch's noise sums).
template <bool CPLX>
__device__ __forceinline__ void fir_epilogue(const Stage2Args& a, int r, int det, int i0, int H, int tid, const float* sre,
                                             float2 (&acc)[kW]) {
  const int ib = i0 + kW * tid;
  // Block peaks (and the squelch's noise sums): a wave's 512 outputs lie in its first block wb0
  // or the next one, so every lane sorts its values into those two slots, the wave reduces both
  // and lane 0 issues at most two atomics per quantity.  (One atomic per LANE whenever a wave
  // held a block boundary -- every other wave -- was 0.5 M atomics per call and 20 us.)  Blocks
  // shorter than that (tiny chunk_len) fall through to per-element atomics.
  const bool squelch = (a.sq_thresh[r] > 0.f) && (det == kDetFm);
  const bool valid = ib < a.n_out;                        // (ib = -1 for lane 0 of tile 0 when par: its outputs 0 .. 6 exist)
  uint32_t blk_lo = 0xFFFFFFFFu, blk_hi = 0xFFFFFFFFu;
  if (valid) {
    const int last = (ib + kW - 1 < a.n_out) ? ib + kW - 1 : a.n_out - 1;
    blk_lo = block_of(a, r, ib < 0 ? 0 : ib);
    blk_hi = block_of(a, r, last);
    // the lane's 8 outputs are 32 (64) contiguous bytes: 16-byte stores, not eight strided dwords
    if (ib >= 0 && ib + kW <= a.n_out) {
      if (CPLX) {
        float4* o = reinterpret_cast<float4*>(a.a[r] + ib);
#pragma unroll
        for (int j = 0; j < kW; j += 2) o[j / 2] = make_float4(acc[j].x, acc[j].y, acc[j + 1].x, acc[j + 1].y);
      } else {
        float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.a[r]) + ib);
        o[0] = make_float4(acc[0].x, acc[1].x, acc[2].x, acc[3].x);
        o[1] = make_float4(acc[4].x, acc[5].x, acc[6].x, acc[7].x);
      }
    }
  }
  const uint32_t wb0 = __shfl(blk_lo, 0);                 // lanes ascend: lane 0 invalid = wave invalid
  // The squelch's block sums are accumulated as 64-bit FIXED-POINT integers (kSqFix = 2^48 per unit): integer addition is
  // associative, so a block's sum does not depend on how a call's waves and lanes happen to partition it, nor on the order their
  // atomics arrive in -- as float sums they were reproducible to an ulp, and batch == chunk by chunk held bit for bit only while
  // the blocks kept their alignment to the waves (round 6: 1 MS/s, where a chunk is 1023.98 outputs, showed it).
  float m0 = 0.f, m1 = 0.f;
  unsigned long long nz0 = 0ull, nz1 = 0ull, lz0 = 0ull, lz1 = 0ull;
  unsigned cnt0 = 0u, cnt1 = 0u;
  // The RATIO squelch (sigs/squelch.m:92-145): z1 = low-pass < 3 kHz, z2 = high-pass > 4 kHz of the discriminator output
  // (FIRs of sq_ntaps taps over the staged detector values), per-sample one-pole envelopes of |z1|, |z2| with alpha = 0.001.
  // A one-pole is linear: behind a block of N samples it is (1 - alpha)^N s + sum_k alpha (1 - alpha)^(N-1-k) |z_k| -- this
  // kernel leaves the block's weighted sums (blknoise2 / blknoise) and count, agc_scan_kernel runs the block recursion.
  const bool ratio = squelch && a.sq_ratio[r] != 0;
  float z1[kW], z2[kW];
  int bend_lo = 0, bend_hi = 0;
  if (ratio && valid) {
    const int e0 = kW * tid + H + 3;                      // S element of output ib
    float win[kW];
#pragma unroll
    for (int j = 0; j < kW; ++j) { z1[j] = 0.f; z2[j] = 0.f; win[j] = sre[fir_pad(e0 + j)]; }
    for (int q = 0; q < a.sq_ntaps; ++q) {                // an output's sum runs over its taps in order: the same in any call
      const float tl = a.sqtaps[q], th = a.sqtaps[kSqTapsMax + q];
#pragma unroll
      for (int j = 0; j < kW; ++j) { z1[j] = __fmaf_rn(tl, win[j], z1[j]); z2[j] = __fmaf_rn(th, win[j], z2[j]); }
#pragma unroll
      for (int j = kW - 1; j > 0; --j) win[j] = win[j - 1];
      win[0] = sre[fir_pad(e0 - q - 1)];
    }
    bend_lo = block_end(a, r, blk_lo);
    bend_hi = (blk_hi == blk_lo) ? bend_lo : block_end(a, r, blk_hi);
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < kW; ++j)
      if (ib + j >= 0 && ib + j < a.n_out) {
        if (ib < 0 || ib + kW > a.n_out) {
          if (CPLX) a.a[r][ib + j] = acc[j];
          else reinterpret_cast<float*>(a.a[r])[ib + j] = acc[j].x;   // real outputs: 4 bytes each
        }
        const float m = CPLX ? sqrtf(acc[j].x * acc[j].x + acc[j].y * acc[j].y) : fabsf(acc[j].x);
        const uint32_t bj = (blk_lo == blk_hi) ? blk_lo : block_of(a, r, ib + j);
        const uint32_t slot = bj - wb0;
        float hp = 0.f, lp = 0.f;
        if (ratio) {
          const int bend = (bj == blk_lo) ? bend_lo : ((bj == blk_hi) ? bend_hi : block_end(a, r, bj));
          const float w = kSqAlpha * __builtin_amdgcn_exp2f(kSqLog2Decay * (float)(bend - 1 - (ib + j)));
          hp = w * fabsf(z2[j]);
          lp = w * fabsf(z1[j]);
        } else if (squelch) {
          // NFM noise squelch (sigs/squelch.m:92-145): block sum of |2nd difference| of the
          // detector output -- out-of-band noise rises when the carrier goes away
          const int e = kW * tid + H + 3 + j;             // S element of output ib+j
          hp = fabsf(sre[fir_pad(e)] - 2.f * sre[fir_pad(e - 1)] + sre[fir_pad(e - 2)]);
        }
        const unsigned long long hq = squelch ? sq_fix(hp) : 0ull, lq = ratio ? sq_fix(lp) : 0ull;
        if (slot == 0u) { m0 = fmaxf(m0, m); nz0 += hq; lz0 += lq; cnt0 += 1u; }
        else if (slot == 1u) { m1 = fmaxf(m1, m); nz1 += hq; lz1 += lq; cnt1 += 1u; }
        else {
          const size_t k = ((size_t)r * a.nchunks + bj) * kBlkStride;
          atomicMax(a.blkpeak + k, __float_as_uint(m));
          if (squelch) { atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise + k), hq); atomicAdd(a.blkcnt + k, 1u); }
          if (ratio) atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise2 + k), lq);
        }
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
  if (squelch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nz0 += shfl_xor_u64(nz0, o); nz1 += shfl_xor_u64(nz1, o);
      cnt0 += __shfl_xor(cnt0, o); cnt1 += __shfl_xor(cnt1, o);
    }
    if (ratio) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { lz0 += shfl_xor_u64(lz0, o); lz1 += shfl_xor_u64(lz1, o); }
    }
  }
  if ((tid & 63) == 0 && wb0 != 0xFFFFFFFFu) {
    // (a single-block RX -- broadcast FM -- deals its waves over single_spread accumulators: agc_scan_kernel folds them)
    const uint32_t sp = a.single_block[r] ? (blockIdx.x & (uint32_t)(a.single_spread - 1)) : 0u;
    const size_t k0 = ((size_t)r * a.nchunks + wb0 + sp) * kBlkStride, k1 = k0 + kBlkStride;
    if (m0 > 0.f) atomicMax(a.blkpeak + k0, __float_as_uint(m0));
    if (m1 > 0.f) atomicMax(a.blkpeak + k1, __float_as_uint(m1));
    if (squelch) {
      if (cnt0) { atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise + k0), nz0); atomicAdd(a.blkcnt + k0, cnt0); }
      if (cnt1) { atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise + k1), nz1); atomicAdd(a.blkcnt + k1, cnt1); }
      if (ratio && cnt0) atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise2 + k0), lz0);
      if (ratio && cnt1) atomicAdd(reinterpret_cast<unsigned long long*>(a.blknoise2 + k1), lz1);
    }
  }
}

// Two instantiations: the complex product (IQ mode, broadcast FM) needs twice the accumulators --
// compiled together, every RX would run at its 104 registers = 4 waves per SIMD instead of 7.
template <bool CPLX>
__global__ __launch_bounds__(kFirThreads) void demod_fir_kernel(const Stage2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int r = a.fir_rx[blockIdx.y];
  const int tid = threadIdx.x;
  // The pairing of taps below depends on whether an output is an even or an odd one of its lane (fir_block): the tiles
  // are laid out from an EVEN absolute output index, so that an output's sum runs in the same order whatever call it
  // falls into (a batch of chunks with odd output counts against the chunk-by-chunk loop: 1 ulp apart before round 4,
  // test_batch_of_chunks_with_odd_output_counts_equals_chunked_bit_exact).  par = 1: output -1 of tile 0 does not exist.
  const int par = (int)(a.m0_lo & 1u);
  const int i0 = blockIdx.x * kFirOut - par;
  const int det = a.det[r];
  const float2* y = (det == kDetPll) ? a.ypll[r] : a.y[r];
  const int H = fir_taps_padded(a.ntaps);                // taps, zero padded to whole groups of 12
  const int nblk = H / 4;
  const int E = kFirOut + H + 12;                        // staged elements: S[e] = d[i0 - H + e - 3]
  const int EP = (fir_pad(E) + 3) & ~3;
  float* sre = lds_f;                                    // [EP]
  float* sim = lds_f + EP;                               // [EP]
  float* tre = lds_f + 2 * EP;                           // [H]
  float* tim = tre + H;                                  // [H]
  float* tre_s = tim + H;                                // [H]  taps shifted by one: tre_s[k] = tre[k + 1]
  float* tim_s = tre_s + H;                              // [H]
  const float2* taps = a.aftaps[r];
  // a thread's ~9 elements are independent: unrolled, their loads are in flight together
#pragma unroll 5
  for (int e = tid; e < E; e += kFirThreads) {
    const int i = i0 - H + e - 3;
    // (indices in front of the kept history only ever meet zero-padded taps)
    const float2 d = (i >= -a.hy + 2) ? detect(a, r, det, y, i) : make_float2(0.f, 0.f);
    sre[fir_pad(e)] = d.x;
    sim[fir_pad(e)] = d.y;
  }
  for (int k = tid; k <= H; k += kFirThreads) {
    const float2 c = (k < a.ntaps) ? taps[k] : make_float2(0.f, 0.f);
    if (k < H) { tre[k] = c.x; tim[k] = c.y; }
    if (k > 0) { tre_s[k - 1] = c.x; tim_s[k - 1] = c.y; }
  }
  __syncthreads();

  float2 acc[kW];
  if (CPLX) {
    fir_run<kFirCplx>(sre, sim, tre, tim, tre_s, tim_s, nblk, H, tid, acc);
  } else {
    const bool real_det = (det == kDetAbs || det == kDetFm || det == kDetPll);
    if (real_det && a.taps_real[r]) fir_run<kFirRealReal>(sre, sim, tre, tim, tre_s, tim_s, nblk, H, tid, acc);
    else fir_run<kFirRePart>(sre, sim, tre, tim, tre_s, tim_s, nblk, H, tid, acc);
  }

  fir_epilogue<CPLX>(a, r, det, i0, H, tid, sre, acc);
}

// (Round 5 tried this inner product on the matrix cores -- taps shifted along the columns of B, 70 v_mfma_f32_16x16x4_f32 per
// 256 outputs, parity green -- and measured it SLOWER on every workload: C1's AF stage 48 -> 67 us, 6 RX 178 -> 309 us
// (profiles/r05_fir_mfma.txt).  The f32 matrix rate of this part equals its packed-FMA rate, 32 FMAs per cycle and SIMD, so
// there was nothing to win but LDS traffic, which is not what bounds the stage.  The kernel is in the history: LABNOTES 9.5.)

// ---- AGC recursion over the blocks of this call (sigs/agc.m:6-12 loop filter on decay,
// immediate attack).  One workgroup per RX.  Only the envelope recursion is serial
// (lane 0, out of LDS, ~5 dependent VALU ops per block); the gains -- one IEEE division
// each -- are then computed by all lanes in parallel.
// one step of the envelope recursion, exactly the oracle's float32 operations
__device__ __forceinline__ float agc_env_step(float env, float peak) {
  const float dec = __fadd_rn(env, __fmul_rn(0.1f, __fsub_rn(peak, env)));
  return (peak > env) ? peak : dec;
}

// Workgroups nrx .. 3 nrx - 1 of the same launch roll the histories of the FS_OUT-rate buffers (prefix <- last hy outputs,
// through LDS so that overlapping source / destination ranges are safe): nothing behind the AF FIR reads those buffers
// again in this call, and as a launch of its own (round 1-3) this copy of a few KB cost 4.2 us of stream time behind the
// gains instead of running beside them.
__global__ __launch_bounds__(256) void agc_scan_kernel(const Stage2Args a, const EpilogueArgs eh) {
  extern __shared__ __attribute__((aligned(16))) float agc_lds[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.nrx) {
    const int job = blockIdx.x - a.nrx;
    const float2* base = (job < eh.nrx) ? eh.ybase[job] : eh.ypllbase[job - eh.nrx];
    float2* dst = (job < eh.nrx) ? eh.ydst[job] : eh.yplldst[job - eh.nrx];
    if (base == nullptr) return;
    float2* sh = reinterpret_cast<float2*>(agc_lds);
    // element j of the new prefix = old element n_out + j  (prefix occupies [0,hy))
    for (int j = tid; j < eh.hy; j += 256) sh[j] = base[eh.n_out + j];
    __syncthreads();
    for (int j = tid; j < eh.hy; j += 256) dst[j] = sh[j];
    return;
  }
  const int r = blockIdx.x;
  // broadcast FM has no AGC blocks: the whole call is block 0 (walking 2047 empty blocks after it
  // let the envelope decay through hundreds of segment joins that never meet: 140 us)
  const int nch = a.single_block[r] ? (a.nchunks > 0 ? 1 : 0) : a.nchunks;
  const int nacc = a.single_block[r] ? (a.nchunks > 0 ? a.single_spread : 0) : a.nchunks;   // accumulators the AF FIR wrote
  // LDS index of block c: one pad word per 16 blocks.  The segments below are 16 blocks long and every lane walks
  // its own: at a pitch of 16 words the 64 lanes of a read sit on TWO banks (16-way conflict: the walk of 192 reads per
  // lane was most of this kernel's 26 us at 4096 blocks); at 17 they sit on all of them.
  auto px = [](int c) { return c + (c >> 4); };
  const int nlds = px(a.nchunks) + 1;
  float* pk = agc_lds;                 // [px(nchunks)] block peaks
  float* ev = agc_lds + nlds;          // [px(nchunks)] envelopes
  float* sS = ev + nlds;               // [256] start state each segment used, [256] end state it reached
  float* sE = sS + 256;
  // sixteen loads in flight per thread (a rolled loop waits for each 256-byte-strided load in turn; these lines were last
  // touched by the AF FIR's atomics and come from the memory side: a round of loads is ~4 us whatever its size -- 4096
  // blocks in two rounds of eight per thread took 8 of this kernel's 20 us, scripts/diag/agc_ablate.sh)
  for (int c0 = 0; c0 < nacc; c0 += 16 * 256) {
    unsigned v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = c0 + u * 256 + tid;
      v[u] = (c < nacc) ? a.blkpeak[((size_t)r * a.nchunks + c) * kBlkStride] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = c0 + u * 256 + tid;
      if (c < nacc) pk[px(c)] = __uint_as_float(v[u]);
    }
  }
  __syncthreads();
  if (a.single_block[r] && nacc > 1) {                 // the one block's peak = the largest of its accumulators
    if (tid == 0) {
      float m = pk[px(0)];
      for (int c = 1; c < nacc; ++c) m = fmaxf(m, pk[px(c)]);
      pk[px(0)] = m;
    }
    __syncthreads();
  }
  const RxDevState st = a.state[r];
  // The recursion is serial, but an attack (peak > env) overwrites the state and a decay forgets it
  // by 0.9 per block: segments of T blocks, each started kWarm blocks early from env = 0, reach the
  // serial walk's value BIT FOR BIT in practice; lane 0 then checks that every segment's start
  // equals its predecessor's end and redoes serially what does not (so the result is the serial
  // one by construction -- the batch / chunked identity tests compare bits).
  constexpr int kWarm = 176;             // 0.9^176 = 9e-9 < 2^-24: a decaying start value is gone from a float
  const int T = (nch + 255) / 256 < 16 ? 16 : (nch + 255) / 256;
  const int K = (nch + T - 1) / T;
  if (tid < K) {
    const int s0 = tid * T, s1 = (s0 + T < nch) ? s0 + T : nch;
    int wb = s0 - kWarm;
    float env = 0.f;
    if (tid == 0 || wb <= 0) { wb = 0; env = st.env; }
    // sixteen peaks are read ahead of the sixteen steps that use them: with one read per step the LDS latency sat on
    // the chain 192 times
    int c = wb;
    for (; c + 16 <= s0; c += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = pk[px(c + u)];
#pragma unroll
      for (int u = 0; u < 16; ++u) env = agc_env_step(env, v[u]);
    }
    for (; c < s0; ++c) env = agc_env_step(env, pk[px(c)]);
    sS[tid] = env;
    for (c = s0; c + 16 <= s1; c += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = pk[px(c + u)];
#pragma unroll
      for (int u = 0; u < 16; ++u) { env = agc_env_step(env, v[u]); ev[px(c + u)] = env; }
    }
    for (; c < s1; ++c) { env = agc_env_step(env, pk[px(c)]); ev[px(c)] = env; }
    sE[tid] = env;
  }
  __syncthreads();
  // (all segments checked at once; the serial walk below only runs when one of them missed)
  const bool miss = tid > 0 && tid < K && __float_as_uint(sE[tid - 1]) != __float_as_uint(sS[tid]);
  if (__syncthreads_or(miss) && tid == 0) {
    for (int k = 1; k < K; ++k) {
      if (__float_as_uint(sE[k - 1]) == __float_as_uint(sS[k])) continue;
      float env = sE[k - 1];
      const int s0 = k * T, s1 = (s0 + T < nch) ? s0 + T : nch;
      for (int c = s0; c < s1; ++c) { env = agc_env_step(env, pk[px(c)]); ev[px(c)] = env; }
      sE[k] = env;
      sS[k] = sE[k - 1];
    }
  }
  __syncthreads();
  const float last_peak = nch > 0 ? pk[px(nch - 1)] : st.maxbuf;
  for (int c = tid; c < nch; c += 256) pk[px(c)] = ev[px(c)];      // below: pk[] holds the envelopes
  __syncthreads();
  for (int c = tid; c < nch; c += 256) {
    const float g = st.agc_enable ? fminf(__fdiv_rn(st.ref, fmaxf(pk[px(c)], 1e-12f)), 1.0e4f) : 1.f;
    a.gain[(size_t)r * a.nchunks + c] = g;
    // the raw block peaks are consumed: leave them zeroed for the next call
    a.blkpeak[((size_t)r * a.nchunks + c) * kBlkStride] = 0u;
  }
  if (a.single_block[r])
    for (int c = 1 + tid; c < nacc; c += 256) a.blkpeak[((size_t)r * a.nchunks + c) * kBlkStride] = 0u;
  const float env_last = nch > 0 ? pk[px(nch - 1)] : 0.f;       // (the squelch section reuses pk[])
  if (a.sq_thresh[r] > 0.f && a.sq_ratio[r]) {
    // The ratio squelch's block recursion: behind a block of n samples each envelope is D s + W with D = (1 - alpha)^n and W the
    // weighted sum the AF FIR kernel left (sq2 from blknoise, sq1 from blknoise2); the gate is open while sq1 >= thresh * sq2
    // (sigs/squelch.m:145: the ratio).  D^17 < 2^-24 at 1024 samples per block, so it runs like the envelope above: segments
    // warmed up over the kSqWarm blocks in front of them, joins compared bit for bit, lane 0 redoes what does not meet.
    constexpr int kSqWarm = 32;
    __syncthreads();
    float* wH = ev;                                  // [px(nch)] W of sq2 (> 4 kHz)
    float* wL = pk;                                  // [px(nch)] W of sq1 (< 3 kHz)
    float* dk = sE + 256;                            // [px(nch)] D, < 0: a block without samples
    float* lvH = dk + nlds;                          // [px(nch)] sq2 behind each block
    float* lvL = lvH + nlds;                         // [px(nch)] sq1 behind each block
    float* sS2 = lvL + nlds;                         // [256] + [256]: the second envelope's segment states
    float* sE2 = sS2 + 256;
    for (int c = tid; c < nch; c += 256) {
      const size_t k = ((size_t)r * a.nchunks + c) * kBlkStride;
      const unsigned n = a.blkcnt[k];
      unsigned long long* qh = reinterpret_cast<unsigned long long*>(a.blknoise + k);
      unsigned long long* ql = reinterpret_cast<unsigned long long*>(a.blknoise2 + k);
      wH[px(c)] = sq_unfix(*qh);
      wL[px(c)] = sq_unfix(*ql);
      dk[px(c)] = n > 0u ? __builtin_amdgcn_exp2f(kSqLog2Decay * (float)n) : -1.f;
      *qh = 0ull;
      *ql = 0ull;
      a.blkcnt[k] = 0u;
    }
    __syncthreads();
    auto walk = [&](int c0, int c1, float& h, float& l, bool store) {
      for (int c = c0; c < c1; ++c) {
        const float d = dk[px(c)];
        if (d >= 0.f) { h = __fmaf_rn(d, h, wH[px(c)]); l = __fmaf_rn(d, l, wL[px(c)]); }
        if (store) { lvH[px(c)] = h; lvL[px(c)] = l; }
      }
    };
    if (tid < K) {
      const int s0 = tid * T, s1 = (s0 + T < nch) ? s0 + T : nch;
      int wb = s0 - kSqWarm;
      float h = 0.f, l = 0.f;
      if (tid == 0 || wb <= 0) { wb = 0; h = st.sq_hp; l = st.sq_lp; }
      walk(wb, s0, h, l, false);
      sS[tid] = h; sS2[tid] = l;
      walk(s0, s1, h, l, true);
      sE[tid] = h; sE2[tid] = l;
    }
    __syncthreads();
    const bool sq_miss = tid > 0 && tid < K && (__float_as_uint(sE[tid - 1]) != __float_as_uint(sS[tid]) ||
                                                __float_as_uint(sE2[tid - 1]) != __float_as_uint(sS2[tid]));
    if (__syncthreads_or(sq_miss) && tid == 0) {
      for (int k = 1; k < K; ++k) {
        if (__float_as_uint(sE[k - 1]) == __float_as_uint(sS[k]) && __float_as_uint(sE2[k - 1]) == __float_as_uint(sS2[k])) continue;
        const int s0 = k * T, s1 = (s0 + T < nch) ? s0 + T : nch;
        float h = sE[k - 1], l = sE2[k - 1];
        walk(s0, s1, h, l, true);
        sS[k] = sE[k - 1]; sS2[k] = sE2[k - 1];
        sE[k] = h; sE2[k] = l;
      }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += 256)
      if (dk[px(c)] >= 0.f && !(lvL[px(c)] >= a.sq_thresh[r] * lvH[px(c)])) a.gain[(size_t)r * a.nchunks + c] = 0.f;
    if (tid == 0 && nch > 0) {
      int open = st.sq_open;
      for (int c = nch - 1; c >= 0; --c)
        if (dk[px(c)] >= 0.f) { open = (lvL[px(c)] >= a.sq_thresh[r] * lvH[px(c)]) ? 1 : 0; break; }
      a.state[r].sq_hp = lvH[px(nch - 1)];
      a.state[r].sq_lp = lvL[px(nch - 1)];
      a.state[r].sq_open = open;
    }
    __syncthreads();
  } else if (a.sq_thresh[r] > 0.f) {
    // One-pole smoothing of the block noise, lvl += 0.64 (noise - lvl) over the blocks that hold samples, and the gate
    // (gain 0 = squelched).  The recursion forgets its start by 0.36 per block -- 0.36^17 < 2^-24 -- so it runs like the
    // envelope above: segments of T blocks, each warmed up over the kSqWarm blocks in front of it from lvl = 0, all joins
    // compared bit for bit and only a miss (a run of empty blocks in a warm-up) sends lane 0 on the serial walk.  (One lane
    // walking all blocks was 2048 x three dependent operations ~ 13 us per call with the squelch armed; bench.py does not arm it.)
    constexpr int kSqWarm = 32;
    __syncthreads();                                 // pk[] (envelopes) has been consumed by the gains above
    float* nz = ev;                                  // [px(nch)] block noise, < 0: a block without samples
    float* lv = pk;                                  // [px(nch)] smoothed level behind each block
    for (int c = tid; c < nch; c += 256) {
      const size_t k = ((size_t)r * a.nchunks + c) * kBlkStride;
      const unsigned n = a.blkcnt[k];
      unsigned long long* qh = reinterpret_cast<unsigned long long*>(a.blknoise + k);
      nz[px(c)] = n > 0u ? __fdiv_rn(sq_unfix(*qh), (float)n) : -1.f;
      *qh = 0ull;
      a.blkcnt[k] = 0u;
    }
    __syncthreads();
    auto sq_step = [](float lvl, float noise) { return __fadd_rn(lvl, __fmul_rn(0.64f, __fsub_rn(noise, lvl))); };
    if (tid < K) {
      const int s0 = tid * T, s1 = (s0 + T < nch) ? s0 + T : nch;
      int wb = s0 - kSqWarm;
      float lvl = 0.f;
      if (tid == 0 || wb <= 0) { wb = 0; lvl = st.sq_level; }
      for (int c = wb; c < s0; ++c) { const float v = nz[px(c)]; if (v >= 0.f) lvl = sq_step(lvl, v); }
      sS[tid] = lvl;
      for (int c = s0; c < s1; ++c) { const float v = nz[px(c)]; if (v >= 0.f) lvl = sq_step(lvl, v); lv[px(c)] = lvl; }
      sE[tid] = lvl;
    }
    __syncthreads();
    const bool sq_miss = tid > 0 && tid < K && __float_as_uint(sE[tid - 1]) != __float_as_uint(sS[tid]);
    if (__syncthreads_or(sq_miss) && tid == 0) {
      for (int k = 1; k < K; ++k) {
        if (__float_as_uint(sE[k - 1]) == __float_as_uint(sS[k])) continue;
        float lvl = sE[k - 1];
        const int s0 = k * T, s1 = (s0 + T < nch) ? s0 + T : nch;
        for (int c = s0; c < s1; ++c) { const float v = nz[px(c)]; if (v >= 0.f) lvl = sq_step(lvl, v); lv[px(c)] = lvl; }
        sE[k] = lvl;
        sS[k] = sE[k - 1];
      }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += 256)
      if (nz[px(c)] >= 0.f && !(lv[px(c)] <= a.sq_thresh[r])) a.gain[(size_t)r * a.nchunks + c] = 0.f;
    if (tid == 0 && nch > 0) {
      // the gate follows the last block that held samples
      int open = st.sq_open;
      for (int c = nch - 1; c >= 0; --c)
        if (nz[px(c)] >= 0.f) { open = (lv[px(c)] <= a.sq_thresh[r]) ? 1 : 0; break; }
      a.state[r].sq_level = lv[px(nch - 1)];
      a.state[r].sq_open = open;
    }
    __syncthreads();                                 // pk[] no longer holds envelopes: the block below reads its copy
  }
  if (tid == 0 && nch > 0) {
    const float env = env_last;
    const float g = st.agc_enable ? fminf(__fdiv_rn(st.ref, fmaxf(env, 1e-12f)), 1.0e4f) : 1.f;
    // field-wise: the squelch block above owns sq_level / sq_open
    a.state[r].env = env;
    a.state[r].gain = g;
    a.state[r].maxbuf = last_peak;
    a.state[r].err = __fsub_rn(st.ref, __fmul_rn(g, last_peak));
  }
}

// ---- apply the block gain, emit rx.am (real, or complex in IQ mode)
// Four consecutive outputs per thread: one pair of block_of() (two 32-bit divisions each) per four
// outputs instead of per output -- at one output per thread the kernel was bound by those divisions
// (21 us for 8.4 M outputs), not by its 68 MB of traffic -- and 16-byte accesses.
// the audio is written once and next read by the host: streaming stores (-DS2X_PLAIN_STREAMS for the A/B)
typedef float s2_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(void* p, float4 v) {
#ifdef S2X_PLAIN_STREAMS
  *reinterpret_cast<float4*>(p) = v;
#else
  __builtin_nontemporal_store((s2_f4){v.x, v.y, v.z, v.w}, (s2_f4*)p);
#endif
}

__global__ __launch_bounds__(256) void apply_kernel(const Stage2Args a) {
  const int r = blockIdx.y;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= a.n_out) return;
  const int n = (a.n_out - i < 4) ? a.n_out - i : 4;
  const float* gr = a.gain + (size_t)r * a.nchunks;
  const uint32_t b0 = block_of(a, r, i), b3 = block_of(a, r, i + n - 1);
  float g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = gr[b0];
  if (b3 != b0) {
#pragma unroll
    for (int j = 1; j < 4; ++j) if (j < n) g[j] = gr[block_of(a, r, i + j)];
  }
  const bool real_in = !a.fir_complex[r];               // layout of a.a as the FIR kernel stored it
  const bool cplx_out = a.out_complex[r] || a.matrix[r];
  if (n == 4) {
    if (real_in) {
      // real outputs were stored densely by the FIR kernel
      const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.a[r]) + i);
      st4_stream(a.am[r] + i, make_float4(v.x * g[0], v.y * g[1], v.z * g[2], v.w * g[3]));
      return;
    }
    const float4 u0 = *reinterpret_cast<const float4*>(a.a[r] + i), u1 = *reinterpret_cast<const float4*>(a.a[r] + i + 2);
    const float2 v[4] = {{u0.x, u0.y}, {u0.z, u0.w}, {u1.x, u1.y}, {u1.z, u1.w}};
    if (!cplx_out) {                                      // WFM mono: the real part of a complex pipeline
      st4_stream(a.am[r] + i, make_float4(v[0].x * g[0], v[1].x * g[1], v[2].x * g[2], v[3].x * g[3]));
      return;
    }
    float2 o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = a.matrix[r] ? make_float2((v[j].x + v[j].y) * g[j], (v[j].x - v[j].y) * g[j])
                         : make_float2(v[j].x * g[j], v[j].y * g[j]);
    float4* d = reinterpret_cast<float4*>(reinterpret_cast<float2*>(a.am[r]) + i);
    st4_stream(d, make_float4(o[0].x, o[0].y, o[1].x, o[1].y));
    st4_stream(d + 1, make_float4(o[2].x, o[2].y, o[3].x, o[3].y));
    return;
  }
  for (int j = 0; j < n; ++j) {                          // the ragged end of the call
    if (real_in) { a.am[r][i + j] = reinterpret_cast<const float*>(a.a[r])[i + j] * g[j]; continue; }
    const float2 v = a.a[r][i + j];
    if (!cplx_out) a.am[r][i + j] = v.x * g[j];
    else reinterpret_cast<float2*>(a.am[r])[i + j] =
        a.matrix[r] ? make_float2((v.x + v.y) * g[j], (v.x - v.y) * g[j]) : make_float2(v.x * g[j], v.y * g[j]);
  }
}

// ---- raw-sample history of a decimator: new = last hist_len samples of [old | x]
// (+ zero the raw-peak buffer the NEXT call will accumulate into: api.hip keeps two and flips)
__global__ __launch_bounds__(256) void hist_roll_kernel(const float2* __restrict__ x,
                                                        const float2* __restrict__ hist_old,
                                                        float2* __restrict__ hist_new, int hist_len,
                                                        uint32_t n_total, unsigned* __restrict__ zero, int zero_n) {
  roll_history(x, hist_old, hist_new, hist_len, n_total, zero, zero_n, threadIdx.x, 256);
}

// ---- broadcast FM at the IF rate: polar discriminator (all lanes), then the 19 kHz pilot
// PLL of WFM2 -- inherently serial, one lane per RX, on a 32-bit phase accumulator.
// grid (ceil(n1 / 4096), nrx): a workgroup takes a tile of kSeedTile = 4096 samples, sixteen per thread 256 apart.  For a
// stereo RX whose pilot loop is seeded (pllseed.hip) it also leaves mpx * norm in the seed kernels' order -- sample j of lane l
// of the tile at j * 64 + l, i.e. the tile transposed -- through LDS, so that both copies leave as whole lines (written straight
// from the registers the transposed copy was 32 eight-byte pieces per wave store and doubled this kernel's time: 36 -> 73 us).
__global__ __launch_bounds__(256) void wfm_disc_kernel(const WfmArgs a) {
  __shared__ float tile[kSeedTile + kSeedTile / kSeedRun];       // one pad word per lane run: the 64 runs start on different banks
  const int r = blockIdx.y, t = threadIdx.x;
  const int base = blockIdx.x * kSeedTile;
  const bool seed_copy = a.mnT[r] != nullptr;
#pragma unroll 4
  for (int q = 0; q < kSeedTile / 256; ++q) {
    const int il = q * 256 + t, i = base + il;
    float mn = 0.f;
    if (i < a.n1) {
      // (nontemporal loads here measured 36.1-36.6 us against 34.6-35.8: plain)
      const float2 yb = a.y1[r][i], ya = a.y1[r][i - 1];
      // the 1-sample IF history of the NEXT call (this kernel is the buffer's only reader; until round 5 the pilot loop's
      // patch-up kernel rolled it, which tied the next call's discriminator to this call's pilot loop) -- into the OTHER buffer
      // of the pair only: in the single-stream form the destination is the word thread 0 of workgroup 0 reads as y1[-1] in
      // this very launch, unordered across workgroups; launch_wfm_disc rolls it behind the kernel then
      if (i == a.n1 - 1 && a.y1dst[r] != a.y1base[r]) a.y1dst[r][1] = yb;
      const float re = yb.x * ya.x + yb.y * ya.y;
      const float im = yb.y * ya.x - yb.x * ya.y;
      const float mpx = atan2f(im, re) * a.scale;
      a.w[r][i] = make_float2(mpx, 0.f);
      mn = __fmul_rn(mpx, a.norm);
    }
    if (seed_copy) tile[il + il / kSeedRun] = mn;
  }
  if (!seed_copy) return;
  __syncthreads();
  float* __restrict__ out = a.mnT[r] + (size_t)blockIdx.x * kSeedTile;   // pll_seed_index(base + il) = tile start + (il % kSeedRun) * 64 + il / kSeedRun
#pragma unroll 4
  for (int q = 0; q < kSeedTile / 256; ++q) {
    const int p = q * 256 + t;                                            // position in the transposed tile: j = p >> 6, lane = p & 63
    const int il = (p & 63) * kSeedRun + (p >> 6);
    out[p] = tile[il + il / kSeedRun];
  }
}

// 19 kHz pilot PLL of the stereo decoder: a recursion through cos of its own phase
//   c = cos(theta), e = mpx*c*norm, w += ki*e, phase += fword0 + rint((w + kp*e) * 2^32/2pi)
// on a 32-bit phase accumulator.  History: per-sample loads/stores + cospif: 250 ns per sample;
// one wave walking blocks of 64 broadcast by v_readlane (nothing but ten dependent VALU operations
// per sample on the critical path): 37 ns; fixed-point sweeps over the 64 samples of a block
// (below): ~10 ns; and segments of a call run side by side (wfm_pll_seg_kernel).

// The pilot loop over a range, 64 samples (one per lane) at a time, by FIXED-POINT SWEEPS instead of
// 64 dependent steps: given a guess of the 64 phases every lane computes its sample's error signal
// e = mpx cos(theta) * norm in parallel; the integrator after sample j is w0 + ki * (inclusive scan
// of e), the correction corr_j = rint((w_j + kp e_j) * 2^32/2pi), and the phase in front of sample j
// is ph0 + j fword0 + (exclusive scan of corr).  Sample 0's phase is exact from the start, so sweep k
// makes at least samples 0..k exact and the iteration ends -- at the serial recursion's own
// trajectory -- when a sweep reproduces its input phases bit for bit: measured 6.6 sweeps per block on
// broadcast FM (max 10) against 64 dependent steps of ten instructions each, 37 -> ~10 ns per sample.
// The integrator is summed in scan order instead of sample by sample: up to ~370 words of 2^32 =
// 5e-7 rad away from the sample-by-sample float32 walk of the oracle, inside the 1e-5 audio bar.
// The "exact" walks stop after PllPlan::exact_cap sweeps (5): by the same 0.06^s that is < 3 words of 2^32 per block from
// the bit-stable fixed point (which takes 6.6 sweeps on average, up to 10 where a rounding sits on the fence) -- two
// orders below the scan-order deviation above -- and every walk, parallel or serial, uses the same cap, so the
// time-parallel result still equals the one-segment walk bit for bit where the tests compare them.
// max_it < 66: a COARSE walk for the early part of a warm-up -- after s sweeps the block's phases are
// off by about 0.06^s of the first guess's error (the loop gain over 64 samples), i.e. two sweeps leave
// ~1e-5 rad per block, which the exact tail of the warm-up (6 loop time constants: e^-6) forgets to
// below the 512-word join tolerance.  The patch-up pass still checks every join.
#ifndef WFMX_DEPTH
#define WFMX_DEPTH 8
#endif
constexpr int kWfmDepth = WFMX_DEPTH;                          // blocks per group of loads
template <bool EMIT>
__device__ __forceinline__ void wfm_pll_walk(const WfmArgs& a, float2* __restrict__ o, int i_begin, int i_end,
                                             uint32_t& ph0, float& w0, int lane, int max_it = 66) {
  // The mpx samples are loaded a GROUP of kWfmDepth blocks ahead (plain loads, first touched by the copy at the group's end:
  // am_pll_walk has the history -- one block ahead through inline asm until round 5, which left this walk waiting for its
  // loads: a block of five capped sweeps is ~0.35 us of issue, a load 1-2 us away; 112 us per call where the arithmetic
  // is ~60).  Past the end the index is clamped; dead lanes of a last partial block are zeroed below.
  constexpr int D = kWfmDepth;
  if (i_begin >= i_end) return;
  float cm[D], nm[D];
  const float* __restrict__ mo = reinterpret_cast<const float*>(o);
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    int idx = i_begin + 64 * sl + lane;
    idx = idx < i_end ? idx : i_end - 1;
    cm[sl] = mo[2 * (size_t)idx];
  }
  for (int ib = i_begin; ib < i_end; ib += 64 * D) {
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    int idx = ib + 64 * (D + sl) + lane;
    idx = idx < i_end ? idx : i_end - 1;
    nm[sl] = mo[2 * (size_t)idx];
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) {
    const int i0 = ib + 64 * sl;
    if (i0 >= i_end) break;
    const float m = cm[sl];
    const int count = (i_end - i0 < 64) ? i_end - i0 : 64;
    // once per block instead of once per sweep: the sample times the detector's normalisation, zero in the dead lanes of
    // a last partial block (e = (m norm) cos, where the oracle rounds (m cos) norm: one ulp of e, far inside the join tolerance)
    const float mn = (lane < count) ? __fmul_rn(m, a.norm) : 0.f;
    const uint32_t inc0 = a.fword0 + (uint32_t)__float2int_rn(__fmul_rn(w0, a.rad2word));
    uint32_t ph = ph0 + (uint32_t)lane * inc0;             // guess: free running at the integrator's rate
    const uint32_t base = ph0 + (uint32_t)lane * a.fword0; // the nominal advance is added once per block, the sweeps scan the corrections only
    uint32_t tot = 0u;
    float wj = w0;
    auto sweep = [&](uint32_t pin) -> uint32_t {
      const float rev = (float)(int)pin * (1.0f / 4294967296.0f);
      const float c = __builtin_amdgcn_cosf(rev);
      const float e = __fmul_rn(mn, c);
      // (fused multiply-adds: one rounding where the oracle's NumPy has two -- 1 ulp of an integrator that is already
      //  summed in scan order, see above)
      wj = __fmaf_rn(a.ki, wave_scan_add(e), w0);
      const uint32_t corr = (uint32_t)__float2int_rn(__fmul_rn(__fmaf_rn(a.kp, e, wj), a.rad2word));
      tot = wave_scan_add(corr);
      return base + tot - corr;
    };
    // two sweeps per trip, so that the phases alternate between two registers instead of being copied back every sweep
    if (max_it <= 8) {
      // a capped walk runs its sweeps without asking whether the last one changed anything: the bit-stable fixed point
      // takes 6.6 sweeps on average, so a test per sweep (a compare and six scalar instructions) almost never ends a walk
      // of 3 or 5 early -- and a sweep from the fixed point reproduces it, so the result is the same either way
      for (int it = 0;; it += 2) {
        const uint32_t p1 = sweep(ph);
        if (it + 1 >= max_it) { ph = p1; break; }
        ph = sweep(p1);
        if (it + 2 >= max_it) break;
      }
    } else {
      for (int it = 0;;) {
        const uint32_t p1 = sweep(ph);
        if (!__any(p1 != ph) || ++it >= max_it) { ph = p1; break; }
        ph = sweep(p1);
        if (!__any(ph != p1) || ++it >= max_it) break;
      }
    }
    if (EMIT && lane < count) {
      const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
      const float s2 = __builtin_amdgcn_sinf(2.f * rev);
      // only the carrier product is stored: .x already holds mpx (wfm_disc_kernel), and the warm-ups of
      // neighbouring segments are reading it meanwhile -- no location is both read and written here
      reinterpret_cast<float*>(o + i0 + lane)[1] = __fmul_rn(m, __fmul_rn(2.f, s2));
    }
    ph0 = ph0 + (uint32_t)count * a.fword0 + (uint32_t)__builtin_amdgcn_readlane((int)tot, count - 1);
    w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wj), count - 1));
  }
#pragma unroll
  for (int sl = 0; sl < D; ++sl) cm[sl] = nm[sl];
  }
}

// 512 words of 2^32 = 7.5e-7 rad of pilot phase (1.5e-6 of the 38 kHz carrier); the integrator
// within 1e-9 rad/sample (x 1/(zeta*wn) = 1900 samples of memory = 2e-6 rad)
__device__ __forceinline__ bool wfm_state_differs(uint32_t ph_a, float w_a, uint32_t ph_b, float w_b) {
  const int d = (int)(ph_a - ph_b);
  return !(d <= 512 && d >= -512 && fabsf(w_a - w_b) <= 1.0e-9f);
}

// grid (K, nrx): segment k of RX r.  The warm-ups read the .x (mpx) of samples whose .y (carrier
// product) other segments are storing: disjoint words.
__global__ __launch_bounds__(64) void wfm_pll_seg_kernel(const WfmArgs a) {
  const int r = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  if (!a.stereo[r]) return;
  pll_wave_priority();
  const PllPlan& pl = a.pll;
  const RxDevState* st = a.state + r;
  const int n = a.n1;
  const int s0 = k * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
  if (a.pll_pass == 1 && !st->wfm_redo) return;
  uint32_t ph = st->wfm_phase;
  float w = st->wfm_w;
  const bool fast = pl.Wfast > 0 && st->wfm_slope_ok;
  const int xcap = pl.exact_cap > 0 ? pl.exact_cap : 66;
  const bool seeded = pl.seeded && a.pll_pass == 0 && st->wfm_slope_ok && a.seed[r] != nullptr && k > 0 && s0 - pl.Wseed > 0;
  int wb = seeded ? s0 - pl.Wseed : s0 - (fast ? pl.Wfast : pl.W);
  if (seeded) {
    // the loop's state in front of sample wb from two Newton passes over the whole call (pllseed.hip): within ~30 words of
    // 2^32 of where the exact walk would be, so there is (next to) nothing to forget
    const uint32_t* sd = pl.seg + ((size_t)r * pl.K + k) * 4;
    ph = sd[0];
    w = __uint_as_float(sd[1]);
  } else if (k > 0 && wb > 0) {
    if (fast) {
      // guess: the call's initial phase carried forward at the MEAN increment of the previous call
      // (a locked loop follows the station's crystal: a straight line plus a bounded wobble, off by
      // < 0.005 revolutions where the instantaneous rate below drifts by 0.1), integrator at its mean
      const double sl = st->wfm_slope;
      ph = ph + (uint32_t)wb * a.fword0 + (uint32_t)(long long)llrint(sl * (double)wb);
      w = (float)(sl / (double)a.rad2word);
    } else {
      // guess: the call's initial state free-running at its own rate up to the warm-up start
      const int corr = __float2int_rn(__fmul_rn(w, a.rad2word));
      ph = ph + (uint32_t)wb * (a.fword0 + (uint32_t)corr);
    }
  } else {
    wb = 0;
  }
  if (seeded) {
    if (wb < s0) wfm_pll_walk<false>(a, a.w[r], wb, s0, ph, w, lane, xcap);
  } else if (wb < s0) {
    // coarse sweeps first, the last Wexact samples exactly (both bounds on multiples of 64)
    const int sx = (pl.coarse_sweeps > 0 && s0 - pl.Wexact > wb) ? s0 - pl.Wexact : wb;
    if (wb < sx) {
      // staged: the start error of a warm-up (0.03 rad from the mean-increment guess) is far above what few sweeps leave
      // behind per block (~1.5e-3 rad after one, 1e-4 after two, 6e-6 after three) until it has decayed to that level
      if (pl.Wc_hi > 0 || pl.Wc_mid > 0) {
        const int s_hi = (sx - pl.Wc_hi > wb) ? sx - pl.Wc_hi : wb;
        const int s_mid = (s_hi - pl.Wc_mid > wb) ? s_hi - pl.Wc_mid : wb;
        const int sw_mid = pl.coarse_sweeps > 1 ? pl.coarse_sweeps - 1 : 1, sw_lo = pl.coarse_sweeps > 2 ? pl.coarse_sweeps - 2 : 1;
        if (wb < s_mid) wfm_pll_walk<false>(a, a.w[r], wb, s_mid, ph, w, lane, sw_lo);
        if (s_mid < s_hi) wfm_pll_walk<false>(a, a.w[r], s_mid, s_hi, ph, w, lane, sw_mid);
        if (s_hi < sx) wfm_pll_walk<false>(a, a.w[r], s_hi, sx, ph, w, lane, pl.coarse_sweeps);
      } else {
        wfm_pll_walk<false>(a, a.w[r], wb, sx, ph, w, lane, pl.coarse_sweeps);
      }
    }
    wfm_pll_walk<false>(a, a.w[r], sx, s0, ph, w, lane, pl.tail_cap > 0 ? pl.tail_cap : xcap);
  }
  uint32_t* sg = pl.seg + ((size_t)r * pl.K + k) * 4;
  if (lane == 0) { sg[0] = ph; sg[1] = __float_as_uint(w); }
  wfm_pll_walk<true>(a, a.w[r], s0, s1, ph, w, lane, xcap);
  if (lane == 0) { sg[2] = ph; sg[3] = __float_as_uint(w); }
}

// Between the two passes: did the short warm-ups of pass 0 meet their neighbours?  A handful of
// misses is the patch-up pass's business; many mean that the call's initial state says nothing
// about the pilot phase further on (the stream is not continuous with the previous call, the pilot
// came back, ...): then the mean increment is withdrawn and pass 1 runs every segment again with
// the long warm-up, still side by side, instead of leaving the whole call to the serial pass.
__global__ __launch_bounds__(64) void wfm_pll_check_kernel(const WfmArgs a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (!a.stereo[r]) return;
  RxDevState* st = a.state + r;
  const PllPlan& pl = a.pll;
  int redo = 0;
  int jw = 0;
  float jd = 0.f;
  if ((pl.Wfast > 0 || pl.seeded) && st->wfm_slope_ok && pl.K > 1) {
    const uint32_t* sg = pl.seg + (size_t)r * pl.K * 4;
    int miss = 0;
    // eight joins per lane in flight (one load round trip per 512 joins instead of per 64: this wave is alone on the stream)
    for (int base = 1; base < pl.K; base += 512) {
      uint2 e[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        e[u] = b[u] = make_uint2(0u, 0u);
        if (kk < pl.K) {
          e[u] = *reinterpret_cast<const uint2*>(sg + (size_t)(kk - 1) * 4 + 2);
          b[u] = *reinterpret_cast<const uint2*>(sg + (size_t)kk * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        const bool mm = kk < pl.K && wfm_state_differs(e[u].x, __uint_as_float(e[u].y), b[u].x, __uint_as_float(b[u].y));
        miss += __popcll(__ballot(mm));
        if (kk < pl.K) {
          const int d = (int)(e[u].x - b[u].x);
          jw = max(jw, d < 0 ? -d : d);
          jd = fmaxf(jd, fabsf(__uint_as_float(e[u].y) - __uint_as_float(b[u].y)));
        }
      }
    }
    redo = miss > 2;
  }
  // how close the warm-ups came (pysdr_pll_join_margin: what a cheaper warm-up setting has to be judged by)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { jw = max(jw, __shfl_xor(jw, o)); jd = fmaxf(jd, __shfl_xor(jd, o)); }
  if (lane == 0) {
    st->pll_join_words = jw;
    st->pll_join_dw = jd;
    st->wfm_redo = redo;
    if (redo) st->wfm_slope_ok = 0;
  }
}

__global__ __launch_bounds__(64) void wfm_pll_patch_kernel(const WfmArgs a) {
  const int r = blockIdx.x, lane = threadIdx.x;
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
    for (int base = k; base < K && bad == K; base += 512) {     // eight joins per lane in flight, as in the check kernel
      uint2 e[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        e[u] = b[u] = make_uint2(0u, 0u);
        if (kk < K) {
          e[u] = *reinterpret_cast<const uint2*>(sg + (size_t)(kk - 1) * 4 + 2);
          b[u] = *reinterpret_cast<const uint2*>(sg + (size_t)kk * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = base + 64 * u + lane;
        const bool mm = kk < K && wfm_state_differs(e[u].x, __uint_as_float(e[u].y), b[u].x, __uint_as_float(b[u].y));
        const unsigned long long bal = __ballot(mm);
        if (bal && bad == K) bad = base + 64 * u + __builtin_ctzll(bal);
      }
    }
    if (bad >= K) break;
    uint32_t ph = sg[(size_t)(bad - 1) * 4 + 2];
    float w = __uint_as_float(sg[(size_t)(bad - 1) * 4 + 3]);
    int j = bad;
    bool joined = false;
    while (j < K) {
      const int s0 = j * pl.T, s1 = (s0 + pl.T < n) ? s0 + pl.T : n;
      wfm_pll_walk<true>(a, a.w[r], s0, s1, ph, w, lane, pl.exact_cap > 0 ? pl.exact_cap : 66);
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
  // the call's mean phase increment beyond fword0, for the next call's guesses: per segment the
  // deviation is far below half a revolution (2 Hz of pilot offset = 0.01 cycles over 2048 samples),
  // so the 32-bit differences are unambiguous and their 64-bit sum is exact
  // (the first quarter of the call is left out: a pull-in after a discontinuity would tilt the line)
  long long dev = 0;
  const int k_lo = K / 4;
  if (patched == 0 && K > 1)
    for (int kk = k_lo + lane; kk < K; kk += 64) {
      const int len = ((kk + 1) * pl.T < n ? (kk + 1) * pl.T : n) - kk * pl.T;
      dev += (int)(sg[(size_t)kk * 4 + 2] - sg[(size_t)kk * 4 + 0] - (uint32_t)len * a.fword0);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dev += __shfl_xor(dev, o, 64);
  if (lane == 0) {
    RxDevState* st = a.state + r;
    st->wfm_phase = ph_fin;
    st->wfm_w = w_fin;
    st->pll_segments = K;
    st->pll_patched = patched;
    st->wfm_slope_ok = (patched == 0 && K > 1) ? 1 : 0;
    st->wfm_slope = (double)dev / (double)(n - k_lo * pl.T);
  }
}

}  // namespace

int launch_am_phase(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(am_phase_kernel, dim3((a.n_out + 256 * kPhasePer - 1) / (256 * kPhasePer), a.nrx), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_pll(const Stage2Args& a, hipStream_t st) {
  hipLaunchKernelGGL(am_pll_seg_kernel, dim3(a.pll.K, a.nrx), dim3(64), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  if (a.pll.K > 1) {                          // (a one-segment call writes its end state itself)
    hipLaunchKernelGGL(am_pll_patch_kernel, dim3(a.nrx), dim3(64), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
  }
  return PYSDR_OK;
}

int launch_demod_fir(const Stage2Args& a, hipStream_t st) {
  if (a.n_out <= 0) return PYSDR_OK;
  const int H = fir_taps_padded(a.ntaps);
  const int EP = (fir_pad(kFirOut + H + 12) + 3) & ~3;
  const size_t lds = (size_t)(2 * EP + 4 * H) * sizeof(float);
  // pysdr_create accepts ntaps_af <= 2048: from 1993 taps on the staged detector output + the four tap
  // arrays pass the 64 KB a kernel gets without asking (66.8 KB at 2048).  The attribute is per
  // (function, device), as in launch_rj (mixdec.hip).
  if (lds > 48 * 1024) {
    static std::mutex attr_mu;
    static uint64_t attr_done = 0;
    int dev = 0;
    PYSDR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(demod_fir_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(demod_fir_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done |= 1ull << (dev & 63);
    }
  }
  for (int cplx = 0; cplx < 2; ++cplx) {
    Stage2Args b = a;
    int n = 0;
    for (int r = 0; r < a.nrx; ++r)
      if ((a.out_complex[r] ? 1 : 0) == cplx) b.fir_rx[n++] = r;
    if (n == 0) continue;
    dim3 grid((a.n_out + (int)(a.m0_lo & 1u) + kFirOut - 1) / kFirOut, n);      // tiles start at an even absolute output (demod_fir_kernel)
    if (cplx) hipLaunchKernelGGL(demod_fir_kernel<true>, grid, dim3(kFirThreads), lds, st, b);
    else hipLaunchKernelGGL(demod_fir_kernel<false>, grid, dim3(kFirThreads), lds, st, b);
    PYSDR_HIP_CHECK(hipGetLastError());
  }
  return PYSDR_OK;
}

int launch_agc_scan(const Stage2Args& a, const EpilogueArgs& e, hipStream_t st) {
  if (e.hy > 4096) {
    set_last_error("epilogue: history %d too long", e.hy);
    return PYSDR_ERR_ARG;
  }
  const size_t nlds = (size_t)a.nchunks + (a.nchunks >> 4) + 1;      // padded: one word per 16 blocks (agc_scan_kernel: px)
  bool any_ratio = false;
  for (int r = 0; r < a.nrx; ++r) any_ratio |= (a.sq_ratio[r] != 0 && a.sq_thresh[r] > 0.f);
  // (the ratio squelch's block recursion keeps three more arrays and a second pair of segment states)
  const size_t lds = std::max(((any_ratio ? 5 : 2) * nlds + (any_ratio ? 1024 : 512)) * sizeof(float), (size_t)e.hy * sizeof(float2));
  // from ~7.6k blocks per call on this passes the 64 KB a kernel gets without asking (pysdr_create bounds max_chunks so
  // that it stays inside the 160 KB a workgroup can have); the attribute is per (function, device)
  if (lds > 48 * 1024) {
    // what the kernel may ask for at run time = the workgroup's 160 KB minus what it holds statically (hipcc promotes a small
    // private array of this kernel to 256 bytes of LDS: asking for the full 160 KB was refused with "invalid argument", i.e.
    // every call of more than ~5500 blocks failed -- 1 MS/s batches; scripts/launch_script_rates.py found it in round 6)
    static std::mutex attr_mu;
    static uint64_t attr_done = 0;
    static size_t max_dyn = 0;
    int dev = 0;
    PYSDR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(attr_mu);
    if (max_dyn == 0) {
      hipFuncAttributes fa;
      PYSDR_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(agc_scan_kernel)));
      max_dyn = (size_t)160 * 1024 - fa.sharedSizeBytes;
    }
    if (lds > max_dyn) { set_last_error("agc: %d blocks per call need %zu bytes of LDS (%zu available)", a.nchunks, lds, max_dyn); return PYSDR_ERR_ARG; }
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      PYSDR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(agc_scan_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_dyn));
      attr_done |= 1ull << (dev & 63);
    }
  }
  hipLaunchKernelGGL(agc_scan_kernel, dim3(a.nrx + 2 * e.nrx), dim3(256), lds, st, a, e);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_apply(const Stage2Args& a, hipStream_t st) {
  if (a.n_out <= 0) return PYSDR_OK;
  dim3 grid((a.n_out + 1023) / 1024, a.nrx);
  hipLaunchKernelGGL(apply_kernel, grid, dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_hist_roll(const float2* x, const float2* hist_old, float2* hist_new, int hist_len,
                     uint32_t n_total, unsigned* zero, int zero_n, hipStream_t st) {
  hipLaunchKernelGGL(hist_roll_kernel, dim3(1), dim3(256), 0, st, x, hist_old, hist_new, hist_len, n_total, zero, zero_n);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

int launch_wfm_disc(const WfmArgs& a, hipStream_t st) {
  if (a.n1 > 0) {
    hipLaunchKernelGGL(wfm_disc_kernel, dim3((a.n1 + kSeedTile - 1) / kSeedTile, a.nrx), dim3(256), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
    for (int r = 0; r < a.nrx; ++r)            // one buffer (every live context): the roll in stream order behind its only reader
      if (a.y1dst[r] == a.y1base[r])
        PYSDR_HIP_CHECK(hipMemcpyAsync(a.y1base[r] + 1, a.y1base[r] + 2 + (a.n1 - 1), sizeof(float2), hipMemcpyDeviceToDevice, st));
  }
  return PYSDR_OK;
}

bool wfm_any_stereo(const WfmArgs& a) {
  bool any_stereo = false;
  for (int r = 0; r < a.nrx; ++r) any_stereo |= (a.stereo[r] != 0);
  return any_stereo && a.n1 > 0;
}

int launch_wfm_pll(const WfmArgs& a, hipStream_t st) {
  bool any_stereo = false;
  for (int r = 0; r < a.nrx; ++r) any_stereo |= (a.stereo[r] != 0);
  if (any_stereo && a.n1 > 0) {
    if (a.pll.seeded && a.pll.K > 1) {
      const int rc = launch_wfm_seed(a, st);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(wfm_pll_seg_kernel, dim3(a.pll.K, a.nrx), dim3(64), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
    if ((a.pll.Wfast > 0 || a.pll.seeded) && a.pll.K > 1) {
      hipLaunchKernelGGL(wfm_pll_check_kernel, dim3(a.nrx), dim3(64), 0, st, a);
      PYSDR_HIP_CHECK(hipGetLastError());
      WfmArgs b = a;
      b.pll_pass = 1;
      hipLaunchKernelGGL(wfm_pll_seg_kernel, dim3(a.pll.K, a.nrx), dim3(64), 0, st, b);
      PYSDR_HIP_CHECK(hipGetLastError());
    }
  }
  if (any_stereo && a.n1 > 0) {
    hipLaunchKernelGGL(wfm_pll_patch_kernel, dim3(a.nrx), dim3(64), 0, st, a);
    PYSDR_HIP_CHECK(hipGetLastError());
  }
  return PYSDR_OK;
}

}  // namespace pysdr
