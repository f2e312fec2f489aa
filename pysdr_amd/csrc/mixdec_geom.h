// Tile geometry of the mix + decimate kernel (mixdec.hip): which outputs a tile holds and which input
// samples its LDS image spans, by division (tile_geometry) and by the incremental step the kernel
// takes from one full tile to the next (tile_advance).  Plain integer arithmetic, shared between the
// device code and the host-side sanitizer harness (tests/host_san), which walks every tile of a launch
// both ways and checks that they agree and stay inside the buffers.
#pragma once
#include "common.h"

#if defined(__HIPCC__)
#define PYSDR_HD __device__ __forceinline__
#define PYSDR_UMULHI(a, b) __umulhi((a), (b))
#else
#define PYSDR_HD inline
#define PYSDR_UMULHI(a, b) ((uint32_t)(((uint64_t)(a) * (uint64_t)(b)) >> 32))
#endif

namespace pysdr {

// t / d and t % d with the host's magic = floor(2^32/d)+1 (exact for any 32-bit t: the
// multiply-high estimate is q or q+1); d == 1 has magic 0
PYSDR_HD void divmod_magic(uint32_t t, uint32_t d, uint32_t magic, uint32_t& q,
                                             uint32_t& r) {
  q = (d == 1u) ? t : PYSDR_UMULHI(t, magic);
  r = t - q * d;
  if (r >= d) { q -= 1u; r += d; }
}
PYSDR_HD uint32_t div_magic(uint32_t t, uint32_t d, uint32_t magic) {
  uint32_t q, r;
  divmod_magic(t, d, magic, q, r);
  return q;
}

// Geometry of tile b: outputs [i_first, i_first + tile_n), LDS image = samples [lo, hi]
// (relative to the first sample of the call; negative = history), samples [own_lo, own_hi]
// are the ones this tile contributes to the raw-chunk peak.  (rel_f, p_f) = divmod(t0 +
// i_first*DOWN, UP) and rel_l = floor((t0 + i_last*DOWN)/UP) seed the per-task index
// arithmetic and the incremental step to the next tile.
struct Tile {
  int i_first, tile_n;
  int rel_f, p_f, rel_l;
  int lo, hi, own_lo, own_hi, npairs;
};

PYSDR_HD Tile tile_geometry(const MixDecArgs& a, int b) {
  Tile t;
  t.i_first = b * a.tile_out;
  int n = a.n_out - t.i_first;
  if (n > a.tile_out) n = a.tile_out;
  if (n < 0) n = 0;
  t.tile_n = n;
  int need_lo, need_hi;
  uint32_t q, r;
  divmod_magic(a.t0 + (uint32_t)t.i_first * (uint32_t)a.down, (uint32_t)a.up, a.magic, q, r);
  t.rel_f = (int)q;
  t.p_f = (int)r;
  if (n > 0) {
    need_hi = (int)div_magic(a.t0 + (uint32_t)(t.i_first + n - 1) * (uint32_t)a.down, (uint32_t)a.up, a.magic);
    need_lo = t.rel_f - (a.kpad - 1);
    t.own_hi = need_hi;
  } else {
    need_hi = -1; need_lo = 0; t.own_hi = -1;
  }
  t.rel_l = need_hi;
  t.own_lo = (b == 0) ? 0
                      : (int)div_magic(a.t0 + (uint32_t)(t.i_first - 1) * (uint32_t)a.down, (uint32_t)a.up, a.magic) + 1;
  if (b == a.ntiles - 1) t.own_hi = (int)a.n_total - 1;
  int lo = need_lo < t.own_lo ? need_lo : t.own_lo;
  if (n == 0) lo = t.own_lo;
  t.lo = lo & ~1;
  t.hi = need_hi > t.own_hi ? need_hi : t.own_hi;
  t.npairs = (t.hi - t.lo + 2) >> 1;
  return t;
}

// The same for the tile after the FULL tile `c` when that next tile is full and not the
// last one: additions only (dq/dr = divmod(tile_out*DOWN, UP) and divmod((tile_out-1)*DOWN,
// UP) come from the host), about 20 scalar instructions instead of three divisions.
PYSDR_HD Tile tile_advance(const MixDecArgs& a, const Tile& c) {
  Tile t;
  t.i_first = c.i_first + a.tile_out;
  t.tile_n = a.tile_out;
  int p = c.p_f + a.dr_tile, q = c.rel_f + a.dq_tile;
  if (p >= a.up) { p -= a.up; q += 1; }
  t.rel_f = q;
  t.p_f = p;
  t.rel_l = q + a.dq_last + ((p + a.dr_last >= a.up) ? 1 : 0);
  t.own_lo = c.rel_l + 1;
  t.own_hi = t.rel_l;
  const int need_lo = q - (a.kpad - 1);
  t.lo = (need_lo < t.own_lo ? need_lo : t.own_lo) & ~1;
  t.hi = t.rel_l;
  t.npairs = (t.hi - t.lo + 2) >> 1;
  return t;
}

}  // namespace pysdr
