// Geometry of the matrix-core form of the mix + decimate kernel (mixdec_mfma.hip): compile-time shape of one
// instantiation and the per-call plan the host derives from the absolute sample / output counters.  Plain
// integer arithmetic shared by the device code, api.hip and the host-side sanitizer harness (tests/host_san),
// and modelled index for index by scripts/experiments/mfma_fir_model.py.
//
// Shifted-tap (Toeplitz) formulation.  Output m = UP*(S*G + t) + c  (G = super-group = one MFMA row, t < S
// shift, c < UP position inside the UP outputs that share DOWN inputs) uses the newest sample
//     n_m = P*G + t*DOWN + off_c,   off_c = floor(c*DOWN/UP),   P = S*DOWN,   branch p_c = (c*DOWN) mod UP
//     y[m] = sum_{k<KT} g[p_c][k] x[n_m - k].
// Row G's window = the KP = KT + (S-1)*DOWN + off_{UP-1} samples from L(G) = P*G - (KT-1); window index
// j = n - L(G).  Column 2*(t*UP+c)+part of the 16-wide B operand holds branch p_c's taps displaced to
// j = KT-1 + t*DOWN + off_c - k, zero elsewhere; two v_mfma_f32_16x16x4_f32 chains (A = Re x against
// [Re g | Im g], A = Im x against [-Im g | Re g]) leave Re y / Im y in adjacent columns.
#pragma once
#include <cstdint>

namespace pysdr {

// FLAGS_: bit 0 = CARRY (the operand ring is carried from tile to tile), bit 1 = NT (the LDS-DMA loads are nontemporal)
template <int UP_, int DOWN_, int S_, int KT_, int NB_, int WK_, int NPROD_, int NBUF_, int FLAGS_>
struct MfmaGeo {
  static constexpr int UP = UP_, DOWN = DOWN_, S = S_, KT = KT_, NB = NB_, WK = WK_, NPROD = NPROD_, NBUF = NBUF_;
  static constexpr bool CARRY = (FLAGS_ & 1) != 0;       // the operand ring is carried from tile to tile (needs the NEXT tile landed too)
  static constexpr bool NT = (FLAGS_ & 2) != 0;          // nontemporal copies: measured per shape (profiles/r04_glds_nt.txt)
  static constexpr int P = S * DOWN;                       // samples between consecutive rows
  static constexpr int US = UP * S;                        // outputs per row
  static constexpr int OFF_LAST = ((UP - 1) * DOWN) / UP;
  static constexpr int KP = KT + (S - 1) * DOWN + OFF_LAST;   // window length K'
  static constexpr int NSTEPS = (KP + 3) / 4;              // MFMA k-steps (4 samples each) per chain
  static constexpr int SPW = (NSTEPS + WK - 1) / WK;       // steps of the longest slice: WK contiguous slices of the window,
  static constexpr int kExtra = NSTEPS % WK;               // the first NSTEPS % WK of them one step longer than the rest
  static constexpr int slice_steps(int q) { return NSTEPS / WK + (q < kExtra ? 1 : 0); }
  static constexpr int slice_first(int q) { return q * (NSTEPS / WK) + (q < kExtra ? q : kExtra); }
  static constexpr int KPP = 4 * NSTEPS;                   // window padded to whole k-steps
  static constexpr int SEGB = 8 * P + 16;                  // bytes per LDS segment: P samples + one 16-byte pad
  static constexpr int SPS = P / 2 + 1;                    // 16-byte slots per segment
  static constexpr int ROWS = 16 * NB;
  static constexpr int TILE = ROWS * P;                    // samples a tile advances by (and owns)
  static constexpr int OUT_PER_TILE = ROWS * US;
  static constexpr int IMG_SAMPLES = (ROWS - 1) * P + KPP + 1;
  static constexpr int IMG_SLOTS = (IMG_SAMPLES / P) * SPS + ((IMG_SAMPLES % P) + 1) / 2;
  static constexpr int IMG_PIECES = (IMG_SLOTS + 63) / 64;    // 1 KiB LDS-DMA pieces
  static constexpr int IMG_BYTES = IMG_PIECES * 1024;
  static constexpr int NCONS = NB * WK;                    // consumer waves: (row block, window slice)
  static constexpr int NEPI = (OUT_PER_TILE + 63) / 64;    // producer waves that run the epilogue (one thread per output)
  static constexpr int NDMA = NPROD - NEPI;                // producer waves that issue the copies
  static constexpr int NWAVES = NCONS + NPROD, NTHREADS = 64 * NWAVES;
  static constexpr int AHEAD = (NSTEPS / WK) < 8 ? (NSTEPS / WK) : 8;   // k-steps of operands in flight per consumer wave
  static constexpr int PART_BYTES = NB * WK * 1024;        // one tile's partial accumulators [NB][WK][16 cols][16 rows]
  static constexpr int LDS_BYTES = NBUF * IMG_BYTES + 2 * PART_BYTES;   // NBUF images: NBUF-1 tiles of copies in flight
  static_assert(P % 4 == 0, "a k-step of four samples must not straddle two segments for both parities");
  static_assert(2 * S * UP <= 16, "columns");
  static_assert(NBUF >= 2 + ((FLAGS_ & 1) ? 1 : 0) && NBUF <= 4, "images: one being worked on, (CARRY) one landed ahead of it, the rest in flight");
  static_assert(KT - 1 >= (DOWN + UP - 1) / UP, "tile 0 must own the call's first sample");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(NDMA >= 1, "producer waves");
  // the furthest byte a dot product reads: last row, last sample of the padded window, d = 1
  static_assert((ROWS - 1) * SEGB + KPP * 8 + 16 * (KPP / P) + 8 <= IMG_BYTES, "window outside the image");
};

// Per-call plan (host, 64-bit counters -> 32-bit offsets relative to the call's first sample S0)
struct MfmaPlan {
  int origin_rel0;   // first sample of tile 0's image, relative to S0: even, <= 0
  int d;             // 0 / 1: row 0's window starts d samples into the image (keeps every DMA pair 16-byte aligned)
  int nrel0;         // P*G_first - S0: newest sample of (tile 0, row 0, t 0, c 0), relative to S0
  int mrel0;         // US*G_first - m0: that output's index in the call (<= 0)
  int ntiles;
};

// s0 / m0: absolute index of the call's first input sample / first output; n samples in the call
template <class G>
inline bool mfma_plan(unsigned long long s0, unsigned long long m0, unsigned long long n, MfmaPlan* p) {
  const long long gf = (long long)(m0 / (unsigned long long)G::US);
  const long long l0 = (long long)G::P * gf - (G::KT - 1);
  const long long diff = l0 - (long long)s0;
  const int d = (int)(((diff % 2) + 2) % 2);
  const long long c0 = G::KT - 1 + d;
  const long long origin = (long long)G::P * gf - c0 - (long long)s0;
  if (origin > 0 || origin < -(1LL << 30) || n >= (1ULL << 30)) return false;
  p->origin_rel0 = (int)origin;
  p->d = d;
  p->nrel0 = (int)((long long)G::P * gf - (long long)s0);
  p->mrel0 = (int)((long long)G::US * gf - (long long)m0);
  const long long span = (long long)n - origin;
  long long nt = (span + G::TILE - 1) / G::TILE;
  if (nt < 1) nt = 1;
  p->ntiles = (int)nt;
  return true;
}

}  // namespace pysdr
