// The launch layer of libpysdr_hip.so for the sanitizer build of its HOST half (tests/host_san):
// every launch_* that api.hip calls, as a host function that
//   * reads every input element and writes every output element the real kernel is entitled to touch
//     (the "device" memory is malloc'ed by the fake HIP runtime, so AddressSanitizer checks the sizes and
//     offsets the host code computed: capacities, history prefixes, strides of the per-block arrays);
//   * for the mix + decimate kernel walks ALL tiles of the launch with the kernel's own geometry code
//     (pysdr_amd/csrc/mixdec_geom.h): the incremental step must equal the division, the LDS image must
//     fit the tile buffer and stay inside history + call, outputs and owned samples must partition the
//     call exactly.
// No DSP is computed: parity is the GPU tests' business.  Not part of the product.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "common.h"
#include "mixdec_geom.h"
#include "mixdec_mfma_geom.h"

namespace pysdr {

namespace {

#define SAN_CHECK(cond, ...)                                              \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "host_san: %s:%d: %s -- ", __FILE__, __LINE__, #cond); \
      std::fprintf(stderr, __VA_ARGS__);                                  \
      std::fputc('\n', stderr);                                           \
      std::abort();                                                       \
    }                                                                     \
  } while (0)

volatile float g_sink;

template <class T> void read_all(const T* p, size_t n) {
  const volatile unsigned char* b = reinterpret_cast<const volatile unsigned char*>(p);
  unsigned acc = 0;
  for (size_t i = 0; i < n * sizeof(T); i += 1) acc += b[i];
  g_sink = (float)acc;
}
template <class T> void write_all(T* p, size_t n) { std::memset(p, 0, n * sizeof(T)); }

bool same_tile(const Tile& a, const Tile& b) {
  return a.i_first == b.i_first && a.tile_n == b.tile_n && a.rel_f == b.rel_f && a.p_f == b.p_f && a.rel_l == b.rel_l &&
         a.lo == b.lo && a.hi == b.hi && a.own_lo == b.own_lo && a.own_hi == b.own_hi && a.npairs == b.npairs;
}

}  // namespace

size_t mixdec_lds_bytes(const MixDecArgs& a) {
  return (2 * (size_t)a.tile_cap + (a.taps_lds ? (size_t)a.nrx * a.up * a.kpad : 0) + (size_t)a.nrx * a.ycap) * sizeof(float2);
}

// The host's view of which instantiation a shape runs on (mixdec.hip md_dispatch / MdShape, restated: the templates do not
// compile without hipcc): the long-prototype multi-RX shapes are 12 (2-4 RX) or 8 (5, 6 RX) waves that hold their taps.
MixdecVariant mixdec_variant(int nrx, int up, int kpad, int threads) {
  MixdecVariant v{1024, 0, 1};
  if (kpad == 336 && up == 3 && threads == 1024 && nrx >= 2 && nrx <= 6) { v.tpb = (nrx <= 4) ? 768 : 512; v.can_hold = 1; v.nh = 1; }
  else if (kpad == 96) { v.nh = (nrx > 4) ? 2 : 1; v.can_hold = 1; }
  else if (nrx == 1 && (kpad == 64 || kpad == 256 || kpad == 336)) v.can_hold = 1;
  return v;
}

static void roll_on_host(const float2* x, const float2* hist_old, float2* hist_new, int hist_len, uint32_t n_total, unsigned* zero, int zero_n);
int launch_mixdec(const MixDecArgs& a, int threads, int grid, hipStream_t) {
  if (a.hist_new) roll_on_host(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n);
  SAN_CHECK(threads >= 64 && threads <= 1024 && (threads & 63) == 0, "threads %d", threads);
  if (!a.taps_lds) {
    // the waves hold their taps (no LDS copy of them): the plan must guarantee hold mode for the instantiation's own thread count
    const MixdecVariant v = mixdec_variant(a.nrx, a.up, a.kpad, threads);
    const int nwaves = std::min(threads, v.tpb) / 64;
    SAN_CHECK(v.tpb != 1024 && v.can_hold && a.up * v.nh <= nwaves && a.tile_out % a.up == 0, "taps_lds = 0 without hold mode (tile_out %d, %d waves)", a.tile_out, nwaves);
  }
  SAN_CHECK(grid >= 1, "grid %d", grid);
  SAN_CHECK(mixdec_lds_bytes(a) <= 160 * 1024, "LDS %zu", mixdec_lds_bytes(a));
  SAN_CHECK(a.nrx >= 1 && a.nrx <= PYSDR_MAX_RX && a.up >= 1 && a.down >= 1 && a.kpad % 16 == 0, "shape");
  SAN_CHECK(a.hist_len >= a.kpad + 2 && (a.hist_len & 1) == 0, "hist_len %d kpad %d", a.hist_len, a.kpad);
  SAN_CHECK(a.tile_out >= 2 && (a.tile_out & 1) == 0 && a.ycap == a.yflush * a.tile_out && a.yflush >= 1, "tile_out %d", a.tile_out);
  SAN_CHECK(a.ntiles >= 1 && (long long)a.ntiles * a.tile_out >= a.n_out && (long long)(a.ntiles - 1) * a.tile_out <= std::max(a.n_out, 1), "ntiles");
  SAN_CHECK(a.dq_tile == (int)(((long long)a.tile_out * a.down) / a.up) && a.dr_tile == (int)(((long long)a.tile_out * a.down) % a.up), "tile step");
  // every byte the kernel may read or write exists
  read_all(a.x, a.n_total);
  read_all(a.hist, (size_t)a.hist_len);
  read_all(a.taps, (size_t)a.nrx * a.up * a.kpad);
  const uint32_t nchunks = (a.n_total + a.chunk_len - 1) / a.chunk_len;
  write_all(a.peak, nchunks);
  for (int r = 0; r < a.nrx; ++r) write_all(a.y[r], (size_t)a.n_out);
  // the tile walk
  long long outs = 0, own_next = 0;
  Tile prev{};
  for (int b = 0; b < a.ntiles; ++b) {
    const Tile t = tile_geometry(a, b);
    if (b > 0 && b + 1 < a.ntiles && prev.tile_n == a.tile_out) {
      const Tile s = tile_advance(a, prev);
      SAN_CHECK(same_tile(s, t), "tile %d: incremental step != division (i_first %d/%d lo %d/%d hi %d/%d p_f %d/%d)", b,
                s.i_first, t.i_first, s.lo, t.lo, s.hi, t.hi, s.p_f, t.p_f);
    }
    SAN_CHECK(t.i_first == outs && t.tile_n >= 0 && t.tile_n <= a.tile_out, "tile %d outputs", b);
    outs += t.tile_n;
    SAN_CHECK(t.own_lo == own_next, "tile %d owns from %d, expected %lld", b, t.own_lo, own_next);
    own_next = (long long)t.own_hi + 1;
    SAN_CHECK((t.lo & 1) == 0 && t.lo >= -a.hist_len, "tile %d image starts at %d (history %d)", b, t.lo, a.hist_len);
    SAN_CHECK(t.hi < (int)a.n_total || t.tile_n == 0, "tile %d image ends at %d (call %u)", b, t.hi, a.n_total);
    SAN_CHECK(2 * t.npairs <= a.tile_cap, "tile %d: %d samples > tile_cap %d", b, 2 * t.npairs, a.tile_cap);
    if (t.tile_n > 0) {
      // first tap of the first output and last tap of ... lie inside the image
      SAN_CHECK(t.rel_f - (a.kpad - 1) >= t.lo && t.rel_l <= t.hi, "tile %d: taps outside the image", b);
      // whole-piece DMA of interior tiles may read up to 63 pairs past `hi`: still inside the tile buffer
      const int npieces = (t.npairs + 63) >> 6;
      if (a.aligned16 && t.lo >= 0 && (uint32_t)(t.lo + 128 * npieces) <= a.n_total)
        SAN_CHECK(128 * npieces <= a.tile_cap + 128, "tile %d: whole pieces (%d samples) overrun tile_cap %d", b, 128 * npieces, a.tile_cap);
    }
    prev = t;
  }
  SAN_CHECK(outs == a.n_out, "tiles hold %lld outputs, call has %d", outs, a.n_out);
  SAN_CHECK(own_next == (long long)a.n_total, "tiles own %lld samples, call has %u", own_next, a.n_total);
  return PYSDR_OK;
}

// ---- the short-prototype resampler (resamp_small.hip): every workgroup's input span fits what the launch reserves
int resamp_small_span(int up, int down, int kpad) {
  const long span = (256L * down + up - 1) / up + kpad + 4;
  const long bytes = (span + (long)up * (kpad + 1)) * (long)sizeof(float2);
  if (span > 4096 || bytes > 60 * 1024) return 0;
  return (int)span;
}
int g_small_launches = 0;
int launch_resamp_small(const MixDecArgs& a, int, int, hipStream_t) {
  ++g_small_launches;
  if (a.hist_new && a.n_out > 0) roll_on_host(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n);   // the real launcher starts no kernel without outputs
  const int span = resamp_small_span(a.up, a.down, a.kpad);
  SAN_CHECK(span > 0 && a.nrx == 1, "shape");
  SAN_CHECK(a.hist_len >= a.kpad - 1, "hist_len %d kpad %d", a.hist_len, a.kpad);
  read_all(a.x, a.n_total);
  read_all(a.hist, (size_t)a.hist_len);
  read_all(a.taps, (size_t)a.up * a.kpad);
  write_all(a.y[0], (size_t)a.n_out);
  for (int i0 = 0; i0 < a.n_out; i0 += 256) {
    const int n_here = std::min(256, a.n_out - i0);
    const long long lo = ((long long)a.t0 + (long long)i0 * a.down) / a.up - (a.kpad - 1);
    const long long hi = ((long long)a.t0 + (long long)(i0 + n_here - 1) * a.down) / a.up;
    SAN_CHECK(hi - lo + 1 <= span, "workgroup at output %d needs %lld samples, reserved %d", i0, hi - lo + 1, span);
    SAN_CHECK(hi < (long long)a.n_total, "output %d reads sample %lld of %u", i0 + n_here - 1, hi, a.n_total);
    SAN_CHECK(lo >= -(long long)a.hist_len - 1, "output %d reads %lld samples into the history of %d", i0, -lo, a.hist_len);
  }
  return PYSDR_OK;
}

// ---- the matrix-core form (mixdec_mfma.hip): the same walk the kernel takes, every DMA element read from the
// real buffers, every window checked against what the image holds
namespace {
template <class G>
int walk_mfma(const MixMfmaArgs& a, int grid) {
  SAN_CHECK(grid >= 1, "grid %d", grid);
  SAN_CHECK((a.hist_len & 1) == 0 && a.hist_len >= G::KT, "hist_len %d", a.hist_len);
  SAN_CHECK((a.origin_rel0 & 1) == 0 && a.origin_rel0 <= 0 && (a.d == 0 || a.d == 1), "origin %d d %d", a.origin_rel0, a.d);
  SAN_CHECK(a.nrel0 == a.origin_rel0 + G::KT - 1 + a.d, "nrel0 %d", a.nrel0);
  SAN_CHECK(a.mrel0 <= 0 && a.mrel0 > -G::US, "mrel0 %d", a.mrel0);
  SAN_CHECK(a.ntiles >= 1 && (long long)a.origin_rel0 + (long long)a.ntiles * G::TILE >= (long long)a.n_total, "tiles do not own the call");
  SAN_CHECK(a.mrel0 + (long long)a.ntiles * G::OUT_PER_TILE >= a.n_out, "tiles do not hold the outputs");
  read_all(a.taps, (size_t)G::UP * a.kpad);
  const uint32_t nchunks = (a.n_total + a.chunk_len - 1) / a.chunk_len;
  write_all(a.peak, nchunks);
  write_all(a.y, (size_t)a.n_out);
  std::vector<unsigned char> have((size_t)G::IMG_PIECES * 128);        // per 8-byte unit of the image: holds a stream sample
  long long own_next = 0, outs = 0;
  for (int tb = 0; tb < a.ntiles; ++tb) {
    const int origin = a.origin_rel0 + tb * G::TILE;
    std::fill(have.begin(), have.end(), 0);
    for (int q = 0; q < G::IMG_PIECES * 64; ++q) {
      const int seg = q / G::SPS, w = q - seg * G::SPS;
      const int rel = origin + seg * G::P + 2 * w;
      const bool ok = (w != G::P / 2) && rel >= -a.hist_len && rel + 1 < (int)a.n_total;
      if (!ok) continue;
      const float2* src = (rel >= 0) ? (a.x + rel) : (a.hist + (a.hist_len + rel));
      SAN_CHECK((reinterpret_cast<uintptr_t>(src) & 7u) == 0, "tile %d slot %d: source not 8-byte aligned", tb, q);
      read_all(src, 2);
      have[2 * (size_t)q] = have[2 * (size_t)q + 1] = 1;
    }
    if (a.n_total & 1u) {
      const int u = (int)a.n_total - 1 - origin;
      if (u >= 0) {
        const int seg = u / G::P, q = seg * G::SPS + ((u - seg * G::P) >> 1);
        if (q < G::IMG_PIECES * 64) { read_all(a.x + (a.n_total - 1), 1); have[2 * (size_t)q] = 1; }
      }
    }
    // ownership: the TILE samples from the image's origin
    const int r_lo = origin > 0 ? origin : 0;
    const int r_end = (origin + G::TILE < (int)a.n_total) ? origin + G::TILE : (int)a.n_total;
    if (r_end > r_lo) {
      SAN_CHECK(r_lo == own_next, "tile %d owns from %d, expected %lld", tb, r_lo, own_next);
      for (int r = r_lo; r < r_end; ++r) {
        const int u = r - origin, unit = u + 2 * (u / G::P);
        SAN_CHECK(unit < (int)have.size() && have[(size_t)unit], "tile %d: owned sample %d not in the image", tb, r);
      }
      own_next = r_end;
    }
    // every tap of every valid output reads a sample the image holds; every read of the padded window is inside the image
    for (int e = 0; e < G::OUT_PER_TILE; ++e) {
      const int idx = a.mrel0 + tb * G::OUT_PER_TILE + e;
      const int rho = e / G::US, rem = e - rho * G::US, t = rem / G::UP, c = rem - t * G::UP;
      for (int j : {0, G::KPP - 1}) {                                                  // first and last sample of the padded window
        const int unit = rho * (G::SEGB / 8) + (a.d + j) + 2 * ((a.d + j) / G::P);
        SAN_CHECK(unit >= 0 && unit < (int)have.size(), "tile %d row %d: window sample %d outside the image", tb, rho, j);
      }
      if (idx < 0 || idx >= a.n_out) continue;
      ++outs;
      const int jtop = G::KT - 1 + t * G::DOWN + (c * G::DOWN) / G::UP;
      const int rel_new = a.nrel0 + (tb * G::ROWS + rho) * G::P + t * G::DOWN + (c * G::DOWN) / G::UP;
      SAN_CHECK(rel_new >= 0 && rel_new < (int)a.n_total, "tile %d output %d: newest sample %d outside the call", tb, idx, rel_new);
      for (int k = 0; k < G::KT; ++k) {
        const int j = jtop - k;
        const int unit = rho * (G::SEGB / 8) + (a.d + j) + 2 * ((a.d + j) / G::P);
        SAN_CHECK(have[(size_t)unit], "tile %d output %d tap %d: sample not in the image", tb, idx, k);
      }
    }
  }
  SAN_CHECK(own_next == (long long)a.n_total, "tiles own %lld samples, call has %u", own_next, a.n_total);
  SAN_CHECK(outs == a.n_out, "tiles hold %lld outputs, call has %d", outs, a.n_out);
  return PYSDR_OK;
}
}  // namespace

// the history roll as hist_roll.h does it (one workgroup of the decimator kernel, or the launch of its own)
static void roll_on_host(const float2* x, const float2* hist_old, float2* hist_new, int hist_len, uint32_t n_total, unsigned* zero, int zero_n) {
  SAN_CHECK(hist_new != nullptr && hist_new != hist_old, "history roll in place");
  if (zero_n > 0) write_all(zero, (size_t)zero_n);
  for (int j = 0; j < hist_len; ++j) {
    const long long rel = (long long)n_total - hist_len + j;
    hist_new[j] = (rel >= 0) ? x[rel] : hist_old[hist_len + rel];
  }
}

int mixdec_mfma_shape(int up, int down, int kdec) {
#define PYSDR_MFMA_MATCH(ID, UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY) \
  if (up == UP && down == DOWN && kdec == KT) return ID;
  PYSDR_MFMA_SHAPES(PYSDR_MFMA_MATCH)
#undef PYSDR_MFMA_MATCH
  return -1;
}
bool mixdec_mfma_plan(int shape, unsigned long long s0, unsigned long long m0, unsigned long long n, MfmaPlan* p) {
#define PYSDR_MFMA_PLAN(ID, UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY) \
  if (shape == ID) return mfma_plan<MfmaGeo<UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY>>(s0, m0, n, p);
  PYSDR_MFMA_SHAPES(PYSDR_MFMA_PLAN)
#undef PYSDR_MFMA_PLAN
  return false;
}
int g_mfma_launches = 0;
int launch_mixdec_mfma(int shape, const MixMfmaArgs& a, int grid, hipStream_t) {
  ++g_mfma_launches;
  if (a.hist_new) roll_on_host(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n);
#define PYSDR_MFMA_LAUNCH(ID, UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY) \
  if (shape == ID) return walk_mfma<MfmaGeo<UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY>>(a, grid);
  PYSDR_MFMA_SHAPES(PYSDR_MFMA_LAUNCH)
#undef PYSDR_MFMA_LAUNCH
  SAN_CHECK(false, "no shape %d", shape);
  return PYSDR_ERR_ARG;
}

int launch_hist_roll(const float2* x, const float2* hist_old, float2* hist_new, int hist_len, uint32_t n_total, unsigned* zero, int zero_n, hipStream_t) {
  roll_on_host(x, hist_old, hist_new, hist_len, n_total, zero, zero_n);
  return PYSDR_OK;
}

static void check_plan(const PllPlan& p, int n, int nrx) {
  SAN_CHECK(p.K >= 1 && p.T >= 64 && (long long)p.K * p.T >= n && (long long)(p.K - 1) * p.T < std::max(n, 1), "PLL plan K %d T %d n %d", p.K, p.T, n);
  SAN_CHECK(p.W % 64 == 0 && p.Wfast % 64 == 0 && p.Wexact % 64 == 0 && p.T % 64 == 0, "PLL plan alignment");
  SAN_CHECK(p.Wc_hi % 64 == 0 && p.Wc_mid % 64 == 0 && p.Wc_hi >= 0 && p.Wc_mid >= 0 && p.tail_cap >= 0, "staged warm-up plan");
  write_all(p.seg, (size_t)nrx * p.K * 4);
}

int launch_am_phase(const Stage2Args& a, hipStream_t) {
  for (int r = 0; r < a.nrx; ++r)
    if (a.det[r] == kDetPll) {
      SAN_CHECK(a.ypll[r] != nullptr, "AM-Synch rx %d has no PLL buffer", r);
      read_all(a.y[r], (size_t)a.n_out);
      write_all(a.ypll[r], (size_t)a.n_out);
    }
  return PYSDR_OK;
}

int launch_pll(const Stage2Args& a, hipStream_t) {
  check_plan(a.pll, a.n_out, a.nrx);
  for (int r = 0; r < a.nrx; ++r)
    if (a.det[r] == kDetPll) {
      SAN_CHECK(a.ypll[r] != nullptr, "AM-Synch rx %d has no PLL buffer", r);
      read_all(a.y[r], (size_t)a.n_out);
      write_all(a.ypll[r], (size_t)a.n_out);
      // am_pll_seg_kernel: grid (K, nrx), one wave per segment; segment k > 0 with s0 - W > 0 reads the phase words of
      // [s0 - W, s1) -- the first 64 of them for its start guess, so W >= 64 must hold -- and segment 0 those of [0, s1)
      const int K = a.pll.K, T = a.pll.T, W = a.pll.W;
      SAN_CHECK(W >= 64 && a.pll.Wexact <= W && a.pll.coarse_sweeps >= 0 && a.pll.coarse_sweeps <= 8, "carrier-loop plan W %d Wexact %d coarse %d", W, a.pll.Wexact, a.pll.coarse_sweeps);
      for (int k = 0; k < K; ++k) {
        const long long s0 = (long long)k * T, s1 = std::min<long long>(s0 + T, a.n_out);
        const long long wb = (k > 0 && s0 - W > 0) ? s0 - W : 0;
        SAN_CHECK(s1 > s0 && s1 <= a.n_out, "carrier-loop segment %d is [%lld, %lld) of %d", k, s0, s1, a.n_out);
        if (wb > 0) SAN_CHECK(wb + 64 <= s0 && s0 <= a.n_out, "carrier-loop segment %d guesses from [%lld, %lld) of %d", k, wb, wb + 64, a.n_out);
      }
    }
  read_all(a.state, (size_t)a.nrx);
  return PYSDR_OK;
}

int launch_demod_fir(const Stage2Args& a, hipStream_t) {
  for (int r = 0; r < a.nrx; ++r) {
    const float2* src = (a.det[r] == kDetPll) ? a.ypll[r] : a.y[r];
    read_all(src - a.hy, (size_t)a.hy + a.n_out);                  // history prefix + the call
    read_all(a.aftaps[r], (size_t)((a.ntaps + 3) & ~3));
    write_all(a.a[r], (size_t)a.n_out);
    for (int k = 0; k < a.nchunks; ++k) {                           // per-block accumulators, kBlkStride words apart
      a.blkpeak[((size_t)r * a.nchunks + k) * kBlkStride] = 0u;
      a.blknoise[((size_t)r * a.nchunks + k) * kBlkStride] = 0.f;
      a.blkcnt[((size_t)r * a.nchunks + k) * kBlkStride] = 0u;
    }
    if (a.sq_ratio[r] && a.sq_thresh[r] > 0.f && a.det[r] == kDetFm) {   // the ratio squelch's second set of block sums and its two FIRs
      SAN_CHECK(a.blknoise2 != nullptr && a.sqtaps != nullptr && a.sq_ntaps >= 1 && a.sq_ntaps <= kSqTapsMax, "ratio squelch armed without its buffers (%d taps)", a.sq_ntaps);
      SAN_CHECK(a.sq_ntaps - 1 <= a.hy - 2, "ratio squelch: %d taps reach behind the %d outputs of history", a.sq_ntaps, a.hy);
      read_all(a.sqtaps, (size_t)2 * kSqTapsMax);
      for (int k = 0; k < a.nchunks; ++k) a.blknoise2[((size_t)r * a.nchunks + k) * kBlkStride] = 0.f;
    }
  }
  return PYSDR_OK;
}

static int stub_epilogue(const EpilogueArgs& a);
int launch_agc_scan(const Stage2Args& a, const EpilogueArgs& e, hipStream_t) {
  for (int r = 0; r < a.nrx; ++r) {
    for (int k = 0; k < a.nchunks; ++k) g_sink = (float)a.blkpeak[((size_t)r * a.nchunks + k) * kBlkStride];
    write_all(a.gain + (size_t)r * a.nchunks, (size_t)a.nchunks);
  }
  read_all(a.state, (size_t)a.nrx);
  return stub_epilogue(e);
}

int launch_apply(const Stage2Args& a, hipStream_t) {
  for (int r = 0; r < a.nrx; ++r) {
    read_all(a.a[r], (size_t)a.n_out);
    write_all(a.am[r], (size_t)a.n_out * (a.out_complex[r] ? 2 : 1));
  }
  return PYSDR_OK;
}

static int stub_epilogue(const EpilogueArgs& a) {
  SAN_CHECK(a.hy <= 4096, "hy %d", a.hy);
  for (int r = 0; r < a.nrx; ++r) {
    SAN_CHECK(a.ydst[r] != nullptr, "rx %d: no destination for the next call's prefix", r);
    SAN_CHECK((a.ypllbase[r] == nullptr) == (a.yplldst[r] == nullptr), "rx %d: PLL buffer and its prefix destination", r);
    const std::pair<float2*, float2*> jobs[2] = {{a.ybase[r], a.ydst[r]}, {a.ypllbase[r], a.yplldst[r]}};
    for (const auto& j : jobs)
      if (j.first) {
        read_all(j.first, (size_t)a.hy + a.n_out);
        std::memmove(j.second, j.first + a.n_out, (size_t)a.hy * sizeof(float2));     // (the pair's other buffer when the calls overlap)
      }
  }
  return PYSDR_OK;
}

int launch_wfm_disc(const WfmArgs& a, hipStream_t) {
  for (int r = 0; r < a.nrx; ++r) {
    SAN_CHECK(a.y1[r] == a.y1base[r] + 2, "IF buffer layout");
    read_all(a.y1[r] - 1, (size_t)a.n1 + 1);
    write_all(a.w[r], (size_t)a.n1);
    if (a.mnT[r] != nullptr)                                 // the seed kernels' copy: every sample of the call lands inside the padded buffer
      for (int i = 0; i < a.n1; i += (a.n1 > 4096 ? 997 : 1)) { SAN_CHECK(pll_seed_index(i) < pll_seed_mnt_floats(a.n1), "mnT index of sample %d", i); a.mnT[r][pll_seed_index(i)] = 0.f; }
    SAN_CHECK(a.y1dst[r] != nullptr, "IF prefix destination");
    if (a.n1 > 0) a.y1dst[r][1] = a.y1[r][a.n1 - 1];
  }
  return PYSDR_OK;
}

size_t pll_seed_doubles(int n1max) {
  const size_t nlanes = ((size_t)n1max + kSeedRun - 1) / kSeedRun, nwaves = (nlanes + 63) / 64;
  return nlanes * 6 + nwaves * 6 + nwaves * 2 + nlanes * 2 + 16 * 6;
}

int launch_wfm_seed(const WfmArgs& a, hipStream_t) {
  // pllseed.hip: a lane per 32 samples, a wave per 2048; the scan buffers of a stereo RX hold both levels and pass 1's states
  for (int r = 0; r < a.nrx; ++r)
    if (a.stereo[r] && a.seed[r] != nullptr && a.mnT[r] != nullptr) {
      read_all(a.mnT[r], pll_seed_mnt_floats(a.n1));
      write_all(a.seed[r], pll_seed_doubles(a.n1));
      SAN_CHECK(a.pll.T % kSeedRun == 0 && a.pll.Wseed % kSeedRun == 0 && a.pll.Wseed >= 0, "seed plan T %d Wseed %d", a.pll.T, a.pll.Wseed);
    }
  return PYSDR_OK;
}

bool wfm_any_stereo(const WfmArgs& a) {
  bool any = false;
  for (int r = 0; r < a.nrx; ++r) any |= (a.stereo[r] != 0);
  return any && a.n1 > 0;
}

int launch_wfm_pll(const WfmArgs& a, hipStream_t st) {
  check_plan(a.pll, a.n1, a.nrx);
  if (a.pll.seeded && a.pll.K > 1) { const int rc = launch_wfm_seed(a, st); if (rc) return rc; }
  for (int r = 0; r < a.nrx; ++r)
    if (a.stereo[r]) { read_all(a.w[r], (size_t)a.n1); write_all(a.w[r], (size_t)a.n1); }
  read_all(a.state, (size_t)a.nrx);
  return PYSDR_OK;
}

int launch_quad_mixer(const float2* x, float2* y, size_t n, uint32_t, uint32_t, hipStream_t) {
  read_all(x, n);
  write_all(y, n);
  return PYSDR_OK;
}

int launch_fir_real(const float* xx, const float* h, int nt, float* y, int n, hipStream_t) {
  read_all(xx, (size_t)n + nt - 1);
  read_all(h, (size_t)nt);
  write_all(y, (size_t)n);
  return PYSDR_OK;
}

int launch_psd_pre(const float2* x, size_t hop, int nframes, int chunk, int nfft, const float* win, float2* work, int is_complex,
                   hipStream_t) {
  read_all(win, (size_t)chunk);
  for (int f = 0; f < nframes; ++f) {
    if (is_complex) read_all(x + (size_t)f * hop, (size_t)chunk);
    else read_all(reinterpret_cast<const float*>(x) + (size_t)f * hop, (size_t)chunk);
  }
  write_all(work, (size_t)nframes * nfft);
  return PYSDR_OK;
}

int launch_psd_post(const float2* work, int nframes, int nfft, int half, int, float* out, hipStream_t) {
  read_all(work, (size_t)nframes * nfft);
  write_all(out, (size_t)nframes * (half ? nfft / 2 : nfft));
  return PYSDR_OK;
}

int launch_psd64k(const float2* x, size_t hop, int nframes, const float* win, float2* work, float* out, int, hipStream_t, int) {
  read_all(win, 32768);
  for (int f = 0; f < nframes; ++f) read_all(x + (size_t)f * hop, 32768);
  write_all(work, (size_t)nframes * 65536);
  write_all(out, (size_t)nframes * 65536);
  return PYSDR_OK;
}

}  // namespace pysdr
