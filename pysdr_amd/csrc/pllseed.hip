// Start states for the segments of the WFM2 pilot loop (stage2.hip wfm_pll_seg_kernel; the stereo decoder behind
// gui.py:1703-1704) by NEWTON'S METHOD IN TIME instead of warm-ups.
//
// A segment of the time-parallel loop needs the loop's state at its first sample.  Until round 5 it got
// it by walking the 13 time constants in front of the segment from a straight-line guess -- 77 % of all samples a
// segment walked.  The loop follows the station's crystal, so its phase is a straight line theta_nom (the previous
// call's mean increment) plus a wobble delta of +-0.03 rad that the programme drives through the 30 Hz loop.  Linearise
// the recursion around ANY guessed trajectory g[n] (phases in words of 2^32, eps = phase - g, W = integrator in words
// per sample, mn = mpx * norm):
//     e[n]    ~ c[n] - s[n] eps[n],                c = mn cos g,  s = mn sin g * 2pi/2^32
//     W[n+1]  = W[n] + kiR (c - s eps)
//     eps[n+1] = eps[n] + (g[n] + fword0 - g[n+1]) + W[n+1] + kpR (c - s eps)
// -- a linear time-varying recursion on the state (eps, W), i.e. one 2x2 AFFINE MAP per sample, and affine maps
// compose associatively: a parallel scan gives the state in front of every sample of the call at once.  Pass 1 linearises
// around theta_nom and lands within 8e-5 rad of the true (float32, sample-by-sample) trajectory; pass 2 linearises around
// pass 1's solution and lands within 33 words of 2^32 = 5e-8 rad -- Newton converges quadratically, and that is the
// rounding floor of the float32 recursion itself (scripts/experiments/pilot_linear_seed.py: 50 segment joins from such
// seeds WITHOUT any warm-up, widest 27 words against a tolerance of 512; integrator within 2.4e-11 of 1e-9).
//
// The result is a SEED, not an answer: the segments still walk their own samples with the exact recursion, the check
// kernel still holds every join against the tolerance, and a call whose seeds miss (no lock, a discontinuous stream) is
// redone with the long warm-ups exactly as before -- what changes is that a locked loop no longer pays for them.
//
// Scan layout: a lane owns R = 64 consecutive samples and composes their maps in registers (fp64: the products of
// thousands of near-identity maps must not lose the 1e-9 the integrator is held to); 64 lanes scan their maps by
// shuffles; a one-workgroup kernel scans the per-wave maps of the call; the next pass (or the finish kernel) applies
// prefix-of-waves and prefix-in-wave to get the state at its own first sample.  cos / sin are the same v_cos / v_sin of
// a 24-bit revolution the walk itself uses.
#include "common.h"

namespace pysdr {

namespace {

// like the walks (stage2.hip pll_wave_priority): beside the next call's matrix-core front end these waves got no issue slots at
// equal priority -- pass 1 took 690 us instead of 46, i.e. it ran when the front end was done
#ifndef SEEDX_PRIO
#define SEEDX_PRIO 3
#endif
#ifdef SEEDX_FLOAT
typedef float sreal;                       // A/B: the maps in float32 (buffers stay fp64)
#else
typedef double sreal;
#endif
constexpr int kSeedR = kSeedRun;           // samples per lane (common.h: segments are multiples of 64 samples, so a segment starts a lane)
constexpr double kWord2Rad = 6.283185307179586476925 / 4294967296.0;

struct Aff {                                // x -> M x + v on (eps, W)
  sreal a, b, c, d, p, q;
};
__device__ __forceinline__ Aff aff_identity() { return Aff{(sreal)1, (sreal)0, (sreal)0, (sreal)1, (sreal)0, (sreal)0}; }
// first f, then g
__device__ __forceinline__ Aff aff_then(const Aff& f, const Aff& g) {
  Aff r;
  r.a = g.a * f.a + g.b * f.c;
  r.b = g.a * f.b + g.b * f.d;
  r.c = g.c * f.a + g.d * f.c;
  r.d = g.c * f.b + g.d * f.d;
  r.p = g.a * f.p + g.b * f.q + g.p;
  r.q = g.c * f.p + g.d * f.q + g.q;
  return r;
}
__device__ __forceinline__ void aff_apply(const Aff& f, double& e, double& w) {
  const double e2 = (double)f.a * e + (double)f.b * w + (double)f.p, w2 = (double)f.c * e + (double)f.d * w + (double)f.q;
  e = e2; w = w2;
}
__device__ __forceinline__ Aff aff_shfl_up(const Aff& f, int d) {
  Aff r;
  r.a = __shfl_up(f.a, d, 64); r.b = __shfl_up(f.b, d, 64); r.c = __shfl_up(f.c, d, 64);
  r.d = __shfl_up(f.d, d, 64); r.p = __shfl_up(f.p, d, 64); r.q = __shfl_up(f.q, d, 64);
  return r;
}
__device__ __forceinline__ void aff_store(double* p, const Aff& f) { p[0] = f.a; p[1] = f.b; p[2] = f.c; p[3] = f.d; p[4] = f.p; p[5] = f.q; }
__device__ __forceinline__ Aff aff_load(const double* p) { return Aff{(sreal)p[0], (sreal)p[1], (sreal)p[2], (sreal)p[3], (sreal)p[4], (sreal)p[5]}; }

// cos / sin of a phase given in (fractional) words, as the walk takes them: the word rounded to 24 bits of a revolution
__device__ __forceinline__ void word_cossin(double ph_words, float& c, float& s) {
  const long long w = __double2ll_rn(ph_words);
  const float rev = (float)(int)(uint32_t)(unsigned long long)w * (1.0f / 4294967296.0f);
  c = __builtin_amdgcn_cosf(rev);
  s = __builtin_amdgcn_sinf(rev);
}

// One sample's map around the guess phase g (words), with the increment defect `gd` = g[n] + fword0 - g[n+1]: the rows of M
// are (a, 1) and (c, 1), so a sample map is four numbers and composing it onto an accumulated map six FMAs
struct SMap { sreal a, c, p, q; };
__device__ __forceinline__ SMap sample_map(float mn, double g, double gd, double kik, double kpk) {
  float c, s;
  word_cossin(g, c, s);
  const sreal cc = (sreal)__fmul_rn(mn, c), ss = (sreal)__fmul_rn(mn, s) * (sreal)kWord2Rad;
  SMap f;
  f.a = fma(-(sreal)(kik + kpk), ss, (sreal)1);
  f.c = -(sreal)kik * ss;
  f.p = fma((sreal)(kik + kpk), cc, (sreal)gd);
  f.q = (sreal)kik * cc;
  return f;
}
// first acc, then the sample map f
__device__ __forceinline__ void aff_push(Aff& acc, const SMap& f) {
  const sreal na = fma(f.a, acc.a, acc.c), nb = fma(f.a, acc.b, acc.d);
  const sreal nc = fma(f.c, acc.a, acc.c), nd = fma(f.c, acc.b, acc.d);
  const sreal np = fma(f.a, acc.p, acc.q) + f.p, nq = fma(f.c, acc.p, acc.q) + f.q;
  acc.a = na; acc.b = nb; acc.c = nc; acc.d = nd; acc.p = np; acc.q = nq;
}
__device__ __forceinline__ void smap_apply(const SMap& f, double& e, double& w) {
  const double e2 = fma((double)f.a, e, w) + (double)f.p, w2 = fma((double)f.c, e, w) + (double)f.q;
  e = e2; w = w2;
}

// Seed buffer of one RX (doubles): [lmap: nlanes x 6][wmap: nwaves x 6][wstate: nwaves x 2][s1: nlanes x 2][tot: 16 x 6]
struct SeedView {
  double *lmap, *wmap, *wstate, *s1, *tot;
  int nlanes, nwaves;
};
__device__ __forceinline__ SeedView seed_view(double* base, int n1) {
  SeedView v;
  v.nlanes = (n1 + kSeedR - 1) / kSeedR;
  v.nwaves = (v.nlanes + 63) / 64;
  v.lmap = base;
  v.wmap = v.lmap + (size_t)v.nlanes * 6;
  v.wstate = v.wmap + (size_t)v.nwaves * 6;
  v.s1 = v.wstate + (size_t)v.nwaves * 2;
  v.tot = v.s1 + (size_t)v.nlanes * 2;                       // [16 x 6] totals of the scan kernel's waves
  return v;
}

// grid (ceil(nwaves / 4), nrx), 256 threads, NO LDS: every wave composes the maps of its 64 x R samples.
//   PASS 1: around the straight line  g[n] = phase0 + n (fword0 + slope)
//   PASS 2: around pass 1's solution  g[n] = straight line + eps1[n]; eps1 is re-walked from the lane's pass-1 start state
// The samples come from mnT, the discriminator kernel's copy of mpx * norm in THIS kernel's order (common.h
// pll_seed_index): the j-th samples of the 64 lanes lie side by side.  History: read from the composite itself, a lane's
// run is 256 bytes from its neighbour's, every load of the wave touched 64 lines and the 16 KB working set of a wave IS the
// L1: 117 us per pass, as long for pass 1's arithmetic as for twice as much; staged through LDS (coalesced loads, 8.4 KB per
// wave): 37 / 46 us alone -- and 690 us beside the matrix-core front end of the next call, whose persistent workgroups
// leave 18 KB of LDS free on paper and none in practice (the kernel simply waited for it to end; without LDS it ran beside
// it: profiles/r05_c4_seed_*.txt).
template <int PASS>
__global__ __launch_bounds__(256) void seed_reduce_kernel(const WfmArgs a) {
  const int r = blockIdx.y;
  if (!a.stereo[r] || a.seed[r] == nullptr || a.mnT[r] == nullptr) return;
  const RxDevState* st = a.state + r;
  if (!st->wfm_slope_ok) return;
  const SeedView sv = seed_view(a.seed[r], a.n1);
  __builtin_amdgcn_s_setprio(SEEDX_PRIO);
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int L = wave * 64 + lane;
  const int i0 = L * kSeedR;
  const double R2W = (double)a.rad2word;
  const double kik = (double)a.ki * R2W, kpk = (double)a.kp * R2W;
  const double slope = st->wfm_slope;                       // words per sample beyond fword0
  const double inc = (double)a.fword0 + slope;
  const double ph0 = (double)st->wfm_phase;
  if (wave >= sv.nwaves) return;
  double e1 = 0.0, w1 = 0.0;                                 // pass-1 state in front of sample i (PASS 2 only)
  if (PASS == 2 && L < sv.nlanes) {
    // prefix of the waves in front, then prefix of the lanes in front inside the wave
    e1 = sv.wstate[(size_t)wave * 2 + 0];
    w1 = sv.wstate[(size_t)wave * 2 + 1];
    const Aff lp = aff_load(sv.lmap + (size_t)L * 6);
    aff_apply(lp, e1, w1);
    sv.s1[(size_t)L * 2 + 0] = e1;
    sv.s1[(size_t)L * 2 + 1] = w1;
  }
  Aff acc = aff_identity();
  if (L < sv.nlanes) {
    const float* __restrict__ mine = a.mnT[r] + (size_t)wave * (64 * kSeedR) + lane;     // sample j of this lane: mine[64 j]
    const int cnt = (a.n1 - i0 < kSeedR) ? a.n1 - i0 : kSeedR;
    // the straight line at the lane's first sample, reduced to one revolution (3.6e15 words into a call a double resolves
    // half a word; only the phase modulo 2^32 matters, the increment defect is known analytically) -- the finish kernel forms
    // the same expression for the same sample
    const double gbase = fmod(ph0 + (double)i0 * inc, 4294967296.0);
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) {
      const float mn = mine[64 * j];
      const double gline = gbase + (double)j * inc;
      if (PASS == 1) {
        aff_push(acc, sample_map(mn, gline, -slope, kik, kpk));
      } else {
        // pass 1's own map at this sample advances eps1; the increment defect of the new guess follows from it
        const SMap f1 = sample_map(mn, gline, -slope, kik, kpk);
        double e1n = e1, w1n = w1;
        smap_apply(f1, e1n, w1n);
        aff_push(acc, sample_map(mn, gline + e1, -slope + e1 - e1n, kik, kpk));
        e1 = e1n; w1 = w1n;
      }
    }
  }
  // inclusive scan over the lanes of the wave (lanes past the call hold the identity)
  Aff inc_map = acc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const Aff t = aff_shfl_up(inc_map, d);
    if (lane >= d) inc_map = aff_then(t, inc_map);
  }
  Aff exc = aff_shfl_up(inc_map, 1);
  if (lane == 0) exc = aff_identity();
  if (L < sv.nlanes) aff_store(sv.lmap + (size_t)L * 6, exc);
  if (lane == 63) aff_store(sv.wmap + (size_t)wave * 6, inc_map);
}

// grid (nrx), 256 threads = one wave per SIMD, no LDS: the state in front of every wave of the reduce kernels = (maps of the
// waves before it)(state of the call's first sample).  (What could run beside the next call's front end decided the shape:
// not even 192 bytes of LDS can be had there; one wave over all the maps took 44 us alone and 300 beside it; sixteen waves
// need four wave slots AND 4 x 44 registers per SIMD, which the front end's four waves of 101 do not leave -- that
// workgroup waited for the front end to end.)  Thread t owns a contiguous run of wave maps; each wave scans its
// 64 runs by shuffles and leaves its total in global scratch; behind ONE barrier every wave composes the totals in front of it.
__global__ __launch_bounds__(256) void seed_scan_kernel(const WfmArgs a) {
  const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (!a.stereo[r] || a.seed[r] == nullptr) return;
  const RxDevState* st = a.state + r;
  if (!st->wfm_slope_ok) return;
  __builtin_amdgcn_s_setprio(SEEDX_PRIO);
  const SeedView sv = seed_view(a.seed[r], a.n1);
  const int per = (sv.nwaves + 255) / 256;
  const int w_lo = t * per, w_hi = (w_lo + per < sv.nwaves) ? w_lo + per : sv.nwaves;
  Aff run = aff_identity();
  for (int w = w_lo; w < w_hi; ++w) run = aff_then(run, aff_load(sv.wmap + (size_t)w * 6));
  Aff inc_map = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const Aff u = aff_shfl_up(inc_map, d);
    if (lane >= d) inc_map = aff_then(u, inc_map);
  }
  Aff exc = aff_shfl_up(inc_map, 1);
  if (lane == 0) exc = aff_identity();
  if (lane == 63) aff_store(sv.tot + wv * 6, inc_map);
  __threadfence_block();
  __syncthreads();
  // the call's initial state: eps = 0 (the guess starts ON the true phase), W = the integrator in words per sample;
  // then the totals of the waves in front, then the runs in front inside the wave
  double e = 0.0, w = (double)st->wfm_w * (double)a.rad2word;
  for (int k = 0; k < wv; ++k) aff_apply(aff_load(sv.tot + k * 6), e, w);
  aff_apply(exc, e, w);
  for (int k = w_lo; k < w_hi; ++k) {
    sv.wstate[(size_t)k * 2 + 0] = e;
    sv.wstate[(size_t)k * 2 + 1] = w;
    aff_apply(aff_load(sv.wmap + (size_t)k * 6), e, w);
  }
}

// grid (ceil(K / 256), nrx): the seed of segment k = pass 2's state in front of sample k T - Wseed, as the (phase word,
// integrator in rad/sample) pair the segment kernel starts from.  Segment 0 starts from the call's true state.
__global__ __launch_bounds__(256) void seed_finish_kernel(const WfmArgs a) {
  const int r = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (!a.stereo[r] || a.seed[r] == nullptr) return;
  const RxDevState* st = a.state + r;
  const PllPlan& pl = a.pll;
  if (!st->wfm_slope_ok || k < 1 || k >= pl.K) return;
  const int i = k * pl.T - pl.Wseed;                        // a multiple of 64
  if (i <= 0) return;
  const SeedView sv = seed_view(a.seed[r], a.n1);
  const int L = i / kSeedR, wave = L >> 6;
  double e2 = sv.wstate[(size_t)wave * 2 + 0], w2 = sv.wstate[(size_t)wave * 2 + 1];
  aff_apply(aff_load(sv.lmap + (size_t)L * 6), e2, w2);
  const double g = fmod((double)st->wfm_phase + (double)i * ((double)a.fword0 + st->wfm_slope), 4294967296.0) + sv.s1[(size_t)L * 2 + 0];
  const long long ph = __double2ll_rn(g + e2);
  uint32_t* sg = pl.seg + ((size_t)r * pl.K + k) * 4;
  sg[0] = (uint32_t)(unsigned long long)ph;
  sg[1] = __float_as_uint((float)(w2 / (double)a.rad2word));
}

}  // namespace

size_t pll_seed_doubles(int n1max) {
  const size_t nlanes = ((size_t)n1max + kSeedR - 1) / kSeedR, nwaves = (nlanes + 63) / 64;
  return nlanes * 6 + nwaves * 6 + nwaves * 2 + nlanes * 2 + 16 * 6;
}

// The five launches that turn (initial state, mean increment of the previous call, mpx) into the segments' start states.
int launch_wfm_seed(const WfmArgs& a, hipStream_t st) {
  if (a.n1 <= 0 || a.pll.K <= 1) return PYSDR_OK;
  const int nlanes = (a.n1 + kSeedR - 1) / kSeedR, nwaves = (nlanes + 63) / 64;
  const dim3 gr((nwaves + 3) / 4, a.nrx);
  hipLaunchKernelGGL(seed_reduce_kernel<1>, gr, dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(seed_scan_kernel, dim3(a.nrx), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(seed_reduce_kernel<2>, gr, dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(seed_scan_kernel, dim3(a.nrx), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(seed_finish_kernel, dim3((a.pll.K + 255) / 256, a.nrx), dim3(256), 0, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace pysdr
