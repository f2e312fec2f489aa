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
// an HBM-bound streaming read of interleaved IQ shared by every RX.
//
// Structure: a PERSISTENT grid (one or two workgroups per CU).  Each workgroup walks a
// contiguous run of tiles with two LDS buffers: while the half-waves compute the dot
// products of tile t out of one buffer, the LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction, no VGPR round trip) of tile t+1 is in flight into the other, so HBM
// never waits for the VALU.  The LO-modulated taps are staged once per workgroup.
//   tile       = `tile_out` consecutive outputs = one contiguous input span
//                (tile_out*DOWN/UP + K samples).  The raw-chunk peak |x|^2
//                (rx.auto_mute, receiver.py:239) is reduced from LDS, so every input
//                sample is read from HBM exactly once (+ the K-sample halo).
//   DPP row    = one output: 16 lanes split the K taps, each lane accumulates all RX from
//                one x read (2 packed FMAs per tap and RX), then four DPP steps fold the
//                row; lanes 0..nrx-1 of the row rotate by the LO phase (v_sin/v_cos take
//                revolutions: exact 32-bit phase -> 1.2e-7 abs error) and store.  The four
//                outputs of a wave belong to the same polyphase branch (outputs UP apart),
//                so their tap reads are one broadcast address and the x reads of the four
//                rows (DOWN samples apart) overlap on few banks.
#include "common.h"
#include "mixdec_geom.h"
#include "hist_roll.h"

namespace pysdr {

namespace {

// Work-skipping ablation switches (DESIGN.md 4.1 store-cost / DMA-only measurements) exist
// only in a diagnostic build (-DPYSDR_DIAG); the shipped kernel has no such branches.
#ifdef PYSDR_DIAG
#define PYSDR_DBG(a, bit) ((a).dbg & (bit))
// phase stamps of the tile loop (scripts/diag/mixdec_stamps.py): workgroups 3 and 131, every wave, first 24 tiles
#define PYSDR_STAMP(k)                                                                                              \
  do {                                                                                                              \
    if (a.stamps && lane == 0 && (tb - t_begin) < 24 && (blockIdx.x == 3 || blockIdx.x == 131))                      \
      a.stamps[(((size_t)((blockIdx.x == 3 ? 0 : 1) * 16 + wave) * 24) + (tb - t_begin)) * 8 + (k)] =               \
          __builtin_readcyclecounter();                                                                             \
  } while (0)
#else
#define PYSDR_DBG(a, bit) 0
#define PYSDR_STAMP(k) do {} while (0)
#endif

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
// sum over each row of 16 lanes; every lane of the row gets the total
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_quad_xor1(v);
  v += dpp_quad_xor2(v);
  v += dpp_half_mirror(v);
  v += dpp_mirror(v);
  return v;
}

// max over the 64 lanes (DPP only); result valid in lane 63
__device__ __forceinline__ float wave_max63(float v) {
  v = fmaxf(v, dpp_quad_xor1(v));
  v = fmaxf(v, dpp_quad_xor2(v));
  v = fmaxf(v, dpp_half_mirror(v));
  v = fmaxf(v, dpp_mirror(v));
  // row_bcast15: rows 1,3 <- lane 15 of rows 0,2 ; row_bcast31: rows 2,3 <- lane 31
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)));
  return v;
}

// One 16-byte LDS-DMA element per lane: LDS destination = wave-uniform base (M0) +
// lane*16, source = per-lane global address.  Issued from inline asm on purpose: hipcc
// counts a __builtin_amdgcn_global_load_lds as a pending LDS write and drains it
// (s_waitcnt vmcnt(0)) in front of EVERY later ds_read, which would serialise the copy
// of tile t+1 with the compute of tile t.  The asm form is invisible to that bookkeeping;
// the kernel waits for it explicitly (dma_wait) before the barrier that publishes a tile.
__device__ __forceinline__ void glds16(const void* gsrc, const void* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_wave_base);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" PYSDR_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(dst)
      : "memory");
}
// Same copy with M0 handled by the caller (m0_save / m0_restore around a run of pieces):
// the scalar unit is shared by all waves of a CU, so every s_* instruction saved here is
// saved 16 times per tile.
__device__ __forceinline__ unsigned m0_save() {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
  return keep;
}
__device__ __forceinline__ void m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory"); }
__device__ __forceinline__ void glds16_m0(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" PYSDR_GLDS_POLICY ::"v"(gsrc), "s"(lds_dst) : "memory");
}
// LDS read through an explicit address-space-3 pointer + a constant element offset: the constant
// goes into the DS instruction's offset field (through a generic pointer hipcc spent one VALU add per
// read on the address)
typedef float mx_v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) mx_v2f* lds_cf2;
__device__ __forceinline__ float2 lds_ld(lds_cf2 p, int i) {
  const mx_v2f v = p[i];
  return make_float2(v.x, v.y);
}
// (the empty asm keeps hipcc from folding a rebasing constant back into the offsets, which would
//  make them negative again)
__device__ __forceinline__ lds_cf2 to_lds(const float2* p) {
  unsigned a = (unsigned)(size_t)(lds_cf2)p;
  asm volatile("" : "+v"(a));
  return (lds_cf2)(size_t)a;
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Start the copy of tile `t` into `xs` (does not wait).
__device__ __forceinline__ void stage_tile(const MixDecArgs& a, const Tile& t, float2* xs, int tid,
                                           int nthr) {
  const int npieces = (t.npairs + 63) >> 6;          // 1 KiB (64 pairs) per wave-instruction
  if (a.aligned16 && t.lo >= 0 && (uint32_t)(t.lo + 128 * npieces) <= a.n_total) {
    // interior tile: whole pieces, every pair exists (reads up to 63 pairs past `hi`, still
    // inside the call; tile_cap leaves room for them)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6, lane = tid & 63;
    const char* src = reinterpret_cast<const char*>(a.x + t.lo) + (size_t)(wave * 1024 + lane * 16);
    unsigned dst = __builtin_amdgcn_readfirstlane(
                       (unsigned)(size_t)(const __attribute__((address_space(3))) void*)xs) + (unsigned)wave * 1024u;
    const unsigned step = (unsigned)nwaves * 1024u;
    const unsigned keep = m0_save();
    for (int q = wave; q < npieces; q += nwaves) {
      glds16_m0(src, dst);
      src += step;
      dst += step;
    }
    m0_restore(keep);
  } else if (a.aligned16) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6, lane = tid & 63;
    float4* dst = reinterpret_cast<float4*>(xs);
    for (int q = wave; q * 64 < t.npairs; q += nwaves) {
      const int pi = q * 64 + lane;
      const int rel = t.lo + 2 * pi;
      // a pair is DMA-able when both samples exist: history (rel < 0) or rel+1 < n_total
      const bool ok = pi < t.npairs && (rel < 0 || (uint32_t)rel + 1u < a.n_total);
      const float2* src = (rel >= 0) ? (a.x + rel) : (a.hist + (a.hist_len + rel));
      if (ok) glds16(src, dst + q * 64);
    }
    // the one pair that straddles the end of an odd-length call
    if (tid == 0 && (a.n_total & 1u)) {
      const int rel = (int)a.n_total - 1;
      if (rel >= t.lo && rel <= t.hi) {
        const float2 p0 = a.x[rel];
        *reinterpret_cast<float4*>(xs + (rel - t.lo)) = make_float4(p0.x, p0.y, 0.f, 0.f);
      }
    }
  } else {
    // generic staging for inputs that are only 8-byte aligned (slow path)
    for (int pi = tid; pi < t.npairs; pi += nthr) {
      const int rel = t.lo + 2 * pi;
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
}

// The raw-peak scan of one tile as a function, for the matrix-core shapes (which call it at different points of the tile loop);
// the same code as the block in mixdec_kernel's tile loop, which the vector shapes keep inline: their code generation is pinned
// by measurements.  (Interleaving the copies of the next tile with this scan -- one 1 KiB copy, one pair per lane, ... -- so that
// the 69 copy instructions of a tile do not queue up in front of the CU's address unit measured neutral: ft8tri 0.674 / 0.666
// fused against 0.688 / 0.672 plain, profiles/r06_long_multirx_variants.txt.)
__device__ __forceinline__ void peak_scan(const MixDecArgs& a, const Tile& cur, const float2* xs, int tid, int nthr, int lane,
                                          float& pk_run, uint32_t& pk_chunk, int& pk_lo, int& pk_hi) {
  if (cur.own_hi >= cur.own_lo) {
    const float4* xv = reinterpret_cast<const float4*>(xs);
    const int e_lo = cur.own_lo & ~1, e_hi = cur.own_hi | 1;
    if (e_lo >= pk_lo && e_hi <= pk_hi) {
      // whole pairs inside the current chunk: a maximum does not mind the neighbour
      // sample being counted by two tiles
      const int p_hi = (e_hi - cur.lo) >> 1;
      int pi = ((e_lo - cur.lo) >> 1) + tid;
      // four reads in flight per thread: a tile is 4-5 trips of this loop, and one read per trip
      // put 4-5 LDS latencies in front of every tile's dot products (0.04 of C1's 0.43 ms)
      for (; pi + 3 * nthr <= p_hi; pi += 4 * nthr) {
        const float4 v0 = xv[pi], v1 = xv[pi + nthr], v2 = xv[pi + 2 * nthr], v3 = xv[pi + 3 * nthr];
        const float m0 = fmaxf(fmaf(v0.x, v0.x, v0.y * v0.y), fmaf(v0.z, v0.z, v0.w * v0.w));
        const float m1 = fmaxf(fmaf(v1.x, v1.x, v1.y * v1.y), fmaf(v1.z, v1.z, v1.w * v1.w));
        const float m2 = fmaxf(fmaf(v2.x, v2.x, v2.y * v2.y), fmaf(v2.z, v2.z, v2.w * v2.w));
        const float m3 = fmaxf(fmaf(v3.x, v3.x, v3.y * v3.y), fmaf(v3.z, v3.z, v3.w * v3.w));
        pk_run = fmaxf(fmaxf(pk_run, fmaxf(m0, m1)), fmaxf(m2, m3));
      }
      for (; pi <= p_hi; pi += nthr) {
        const float4 v = xv[pi];
        pk_run = fmaxf(pk_run, fmaxf(fmaf(v.x, v.x, v.y * v.y), fmaf(v.z, v.z, v.w * v.w)));
      }
    } else {
      // the tile straddles chunk boundaries (or the odd end of the call): one masked scan
      // per chunk it touches
      const uint32_t c_lo = div_magic((uint32_t)cur.own_lo, a.chunk_len, a.magic_chunk);
      const uint32_t c_hi = div_magic((uint32_t)cur.own_hi, a.chunk_len, a.magic_chunk);
      for (uint32_t c = c_lo; c <= c_hi; ++c) {
        const long long cb = (long long)c * a.chunk_len;
        const int s_lo = cur.own_lo > cb ? cur.own_lo : (int)cb;
        const long long ce = cb + a.chunk_len - 1;
        const int s_hi = cur.own_hi < ce ? cur.own_hi : (int)ce;
        const int p_lo = (s_lo - cur.lo) >> 1, p_hi = (s_hi - cur.lo) >> 1;
        if (c != pk_chunk) {
          pk_run = wave_max63(pk_run);
          if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
          pk_run = 0.f;
          pk_chunk = c;
        }
        // interior pairs need no masking; the two edge pairs are handled by one lane
        for (int pi = p_lo + 1 + tid; pi < p_hi; pi += nthr) {
          const float4 v = xv[pi];
          pk_run = fmaxf(pk_run, fmaxf(fmaf(v.x, v.x, v.y * v.y), fmaf(v.z, v.z, v.w * v.w)));
        }
        if (tid == 0) {
          const float4 v0 = xv[p_lo], v1 = xv[p_hi];
          const int r0 = cur.lo + 2 * p_lo, r1 = cur.lo + 2 * p_hi;
          if (r0 >= s_lo) pk_run = fmaxf(pk_run, fmaf(v0.x, v0.x, v0.y * v0.y));
          if (r0 + 1 <= s_hi) pk_run = fmaxf(pk_run, fmaf(v0.z, v0.z, v0.w * v0.w));
          if (r1 >= s_lo) pk_run = fmaxf(pk_run, fmaf(v1.x, v1.x, v1.y * v1.y));
          if (r1 + 1 <= s_hi) pk_run = fmaxf(pk_run, fmaf(v1.z, v1.z, v1.w * v1.w));
        }
      }
      const long long cb = (long long)pk_chunk * a.chunk_len;
      const long long ce = cb + a.chunk_len < (long long)a.n_total ? cb + a.chunk_len : (long long)a.n_total;
      pk_lo = (int)cb;
      pk_hi = (int)ce - 1;
    }
  }

}

// The copies and the raw-peak scan of a STEADY tile (mixdec_kernel's add-only tile loop): every argument is a constant of the run
// but the image's source address.  Copies: this wave's 1 KiB pieces, SGPR base + per-lane offset (no vector address arithmetic).
__device__ __forceinline__ void steady_stage(const char* src, unsigned voff0, unsigned lds_dst, int npieces, int wave, int nwaves) {
  const unsigned keep = m0_save();
  for (int q = wave, i = 0; q < npieces; q += nwaves, ++i)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" PYSDR_GLDS_POLICY
                 ::"v"(voff0 + (unsigned)(i * nwaves) * 1024u), "s"(src), "s"(lds_dst + (unsigned)q * 1024u) : "memory");
  m0_restore(keep);
}
__device__ __forceinline__ float steady_peak(const float4* xv, int pi, int p_hi, int nthr, float pk_run) {
  for (; pi + 3 * nthr <= p_hi; pi += 4 * nthr) {
    const float4 v0 = xv[pi], v1 = xv[pi + nthr], v2 = xv[pi + 2 * nthr], v3 = xv[pi + 3 * nthr];
    const float m0 = fmaxf(fmaf(v0.x, v0.x, v0.y * v0.y), fmaf(v0.z, v0.z, v0.w * v0.w));
    const float m1 = fmaxf(fmaf(v1.x, v1.x, v1.y * v1.y), fmaf(v1.z, v1.z, v1.w * v1.w));
    const float m2 = fmaxf(fmaf(v2.x, v2.x, v2.y * v2.y), fmaf(v2.z, v2.z, v2.w * v2.w));
    const float m3 = fmaxf(fmaf(v3.x, v3.x, v3.y * v3.y), fmaf(v3.z, v3.z, v3.w * v3.w));
    pk_run = fmaxf(fmaxf(pk_run, fmaxf(m0, m1)), fmaxf(m2, m3));
  }
  // (one or two trips: left alone hipcc vectorises this remainder by two, with a scalar epilogue and NaN bookkeeping around it:
  //  80 instructions of the steady loop's 440; time within the spread, profiles/r06_long_multirx_variants.txt section 7)
#pragma clang loop vectorize(disable) unroll(disable)
  for (; pi <= p_hi; pi += nthr) {
    const float4 v = xv[pi];
    pk_run = fmaxf(pk_run, fmaxf(fmaf(v.x, v.x, v.y * v.y), fmaf(v.z, v.z, v.w * v.w)));
  }
  return pk_run;
}

// Fold the 16 lanes of each row for N sub-receivers (all 2N partial sums advance one DPP step
// at a time, so consecutive instructions are independent: no DPP hazard stalls), then lane
// s < ncount of every row rotates RX rbase + s by its LO phase and puts the sample into the LDS
// output stage.  The outputs go to that stage, not to memory: stores share vmcnt with the tile
// copies and the wait in front of the barrier is vmcnt(0), so a store per tile holds the next
// tile hostage to its write acknowledge (measured 4.5 vs 5.1 TB/s); the stage is flushed every
// `yflush` tiles with whole-line coalesced stores.
template <int N>
__device__ __forceinline__ void fold_rotate_stage(const float2 (&A)[N], const float2 (&B)[N], int ncount,
                                                  int rbase, int s, bool valid, uint32_t rel, uint32_t p0,
                                                  uint32_t fw, float2* ys, int ycap, int io) {
  float red[2 * N];
#pragma unroll
  for (int r = 0; r < N; ++r) { red[2 * r] = A[r].x - B[r].y; red[2 * r + 1] = A[r].y + B[r].x; }
#pragma unroll
  for (int q = 0; q < 2 * N; ++q) red[q] += dpp_quad_xor1(red[q]);
#pragma unroll
  for (int q = 0; q < 2 * N; ++q) red[q] += dpp_quad_xor2(red[q]);
#pragma unroll
  for (int q = 0; q < 2 * N; ++q) red[q] += dpp_half_mirror(red[q]);
#pragma unroll
  for (int q = 0; q < 2 * N; ++q) red[q] += dpp_mirror(red[q]);
  float sr = 0.f, si = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r)
    if (s == r) { sr = red[2 * r]; si = red[2 * r + 1]; }
  if (valid && s < ncount) {
    const uint32_t ph = p0 + fw * rel;
    const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
    const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
    float2 o;
    o.x = sr * cs - si * sn;
    o.y = sr * sn + si * cs;
    ys[(rbase + s) * ycap + io] = o;
  }
}

// Compile-time shape of one instantiation.  NJ = kpad/16 known at compile time (fully unrolled tap loop) or 0 for a runtime
// loop; TPB = threads per workgroup (the register budget of a wave follows from it: 128 at 1024 threads, 168 at 768);
// NHX = RX groups a tile's work is dealt out in (0: the default of that TPB).
//   TPB = 1024 (the BASELINE configurations, 255-tap prototypes): up to 24 tap pairs per lane are held in registers, the
//        sub-receivers split into two halves above 4.
//   TPB = 768 (round 6: the reference's DEFAULT 1001-tap prototype with SEVERAL sub-receivers -- FT8tri:47-74, TEST:30,
//        params.py:134 -- 21 tap pairs per lane and RX): 12 waves of 168 registers hold the taps of up to THREE sub-receivers
//        (126 registers), so one LDS read of x still serves every RX of the group; with the taps in LDS (<R,0>, where these
//        shapes ran until now) every tap step was 1 + R LDS reads: ft8tri 0.50, 4 RX 0.44, 6 RX 0.30 of the HBM peak
//        (profiles/r06_baseline_long_prototype_multirx.txt).
#ifndef MD_LONG_TPB
#define MD_LONG_TPB 768      // threads of the long-prototype multi-RX shapes (A/B: 512, 768, 1024)
#endif
#ifndef MD_LONG_MM
#define MD_LONG_MM 1         // the long-prototype shapes of 2 - 4 sub-receivers on the matrix cores (A/B: 0 = vector form)
#endif
#ifndef MD_STEADY_VEC1
#define MD_STEADY_VEC1 1     // steady runs for the vector shape <1,11> (1 MS/s x 1 RX: front end 0.376 -> 0.402 of HBM, 5 MS/s x 1: 0.735 -> 0.743; A/B: 0)
#endif
#ifndef MD_UP6_MM
#define MD_UP6_MM 2          // 1001 taps at UP = 6: the matrix-core form from this many sub-receivers (0: never).  One box, front end as a fraction
                             // of HBM, matrix cores / vector form: 1 MS/s x 1 RX 0.32 / 0.375, x 2 0.31 / 0.27, x 3 0.245 / 0.187; 5 MS/s x 2 0.66 / 0.67,
                             // x 4 0.61 / 0.445; 7 MS/s x 3 0.73 / 0.61 (scripts/diag/up6_mm_ab.sh, profiles/r06_launch_script_rates.txt)
#endif
#ifndef MD_UP6_HOLD
#define MD_UP6_HOLD 1        // 1001 taps at UP = 6, one or two sub-receivers: taps held in registers (A/B: 0 = the generic form, taps in LDS)
#endif
#ifndef MD_STEADY
#define MD_STEADY 1          // matrix-core shapes: runs of full interior tiles through the add-only tile loop (A/B: 0 = the generic body for every tile)
#endif
#ifndef MD_MM_EPCONST
#define MD_MM_EPCONST -1     // matrix-core shapes: the epilogue lane's LO constants looked up per task (1) or held in registers (0); -1: held
                             // up to 2 RX (hipcc then hoists more of the task's index arithmetic out of the tile loop: test2rx 0.55 -> 0.58), looked up from 3 (registers)
#endif
#ifndef MD_MM_ONEACC
#define MD_MM_ONEACC 1       // matrix-core shapes with >= 2 RX pairs: one accumulator per pair (0: one per chain; A/B)
#endif
#ifndef MD_PHASE_ORDERS
#define MD_PHASE_ORDERS 2    // matrix-core shapes: how many different phase orders the three waves of a SIMD run (A/B: 1, 2, 3)
#endif
#ifndef MD_LONG_NH
#define MD_LONG_NH 0         // their RX groups: 0 = as few as the registers allow, n = n groups (A/B)
#endif
#ifndef MD_XAHEAD
#define MD_XAHEAD 8          // matrix-core shapes: reads of x in flight ahead of the MFMAs (A/B)
#endif
#ifndef MD_XGROUP
#define MD_XGROUP 7          // x reads per group of the register-tight shapes (A/B)
#endif
//   MM = 1 (round 6, the same shapes): the dot products of a task on the MATRIX cores, v_mfma_f32_4x4x1_16b_f32 -- sixteen
//        independent 4x4 blocks with K = 1.  A task is still (branch, quad of outputs); the sixteen blocks are the sixteen tap
//        residues k mod 16 (what the sixteen lanes of a DPP row are in the vector form), the four ROWS of a block the four
//        outputs of the quad, its four COLUMNS (Re, Im) of two sub-receivers.  So lane 4 b + i reads x[n_i - 16 jj - b] from
//        LDS (one 8-byte read per step, as before), Re x and Im x go through two MFMAs per pair of sub-receivers against the
//        wave's tap registers ([Re g | Im g] and [-Im g | Re g]) -- 256 MACs per instruction, every one of them useful
//        when R is even (75 % at R = 3), where the 16x16x4 shape of mixdec_mfma.hip would fill 2 R of 16 columns.  The tap
//        operands of 2 / 3-4 sub-receivers are 42 / 84 registers (126 held as complex pairs for the vector form), so 12
//        waves fit with nothing spilled; the 16 block sums meet through two DPP row rotations and three permlane swaps.
//        Why: the vector form at these shapes is bound by how many INSTRUCTIONS its few, register-heavy waves can issue
//        (126 packed FMAs per task at 3 RX; profiles/r06_long_multirx_variants.txt), not by arithmetic throughput.
template <int R, int NJ, int TPB, int NHX, int MM = 0>
struct MdShape {
  static constexpr bool kMm = MM != 0;
  static constexpr int G = (R + 1) / 2;                       // MM: pairs of sub-receivers = 4-column groups
  static constexpr int kBudget = TPB > 768 ? 128 : (TPB > 512 ? 168 : 256);     // registers per lane
  static constexpr int kMaxHeld = TPB > 768 ? 24 : (kBudget - 80) / 2;          // tap pairs per lane that may stay in registers (everything else
                                                                                // of the tile loop takes ~75: <1,21> = 42 + 75)
  static constexpr int nh_default() {
    if (MM) return 1;
    if (TPB > 768) return R > 4 ? 2 : 1;
    int nh = 1;
    while (NJ > 0 && ((R + nh - 1) / nh) * NJ > kMaxHeld && nh < R) ++nh;
    return nh;
  }
  static constexpr int NH = NHX > 0 ? NHX : nh_default();     // RX groups
  static constexpr int RH = (R + NH - 1) / NH;                // RX per task in hold mode
  static constexpr bool kCanHold = (NJ > 0) && (MM ? (2 * G * NJ + 60 <= kBudget) : (RH * NJ <= kMaxHeld));
};

typedef float md_f4 __attribute__((ext_vector_type(4)));
// (a, b) -> [a.row0 + a.row1, b.row0 + b.row1, a.row2 + a.row3, b.row2 + b.row3]   (rows of 16 lanes; v_permlane16_swap:
// the odd rows of the first operand change places with the even rows of the second)
__device__ __forceinline__ float swap16_add(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// (a, b) -> [a.row0 + a.row2, a.row1 + a.row3, b.row0 + b.row2, b.row1 + b.row3]   (v_permlane32_swap: the upper half of the
// first operand changes places with the lower half of the second)
__device__ __forceinline__ float swap32_add(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float dpp_row_ror4(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_ror8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

// The matrix-core form's epilogue.  acc[g][i] of lane 4 b + j = block b's sum for output i of the quad, column j of RX pair g
// (the Re x chain and the Im x chain already added).  The sixteen blocks meet in a fixed order -- the four of a 16-lane row by
// two rotations, the four rows by swaps -- which leaves output rho's totals in row rho; there lane (q = pair, j even) has
// Re, its neighbour Im: it rotates RX 2 q + j / 2 by the LO phase and stages the sample like fold_rotate_stage.
template <int G, int R>
__device__ __forceinline__ void mm_fold_rotate_stage(const md_f4 (&acc)[G], int lane, bool valid, uint32_t ph, float2* ys, int ycap,
                                                     int io) {
  float tot[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = acc[g][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += dpp_row_ror4(c[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += dpp_row_ror8(c[i]);
    tot[g] = swap32_add(swap16_add(c[0], c[1]), swap16_add(c[2], c[3]));
  }
  const int q = (lane >> 2) & 3, j = lane & 3;
  float t = tot[0];
#pragma unroll
  for (int g = 1; g < G; ++g)
    if (q == g) t = tot[g];
  const float other = dpp_quad_xor1(t);               // the Im column beside a Re column
  const int rx = 2 * q + (j >> 1);
  if (valid && q < G && (j & 1) == 0 && rx < R) {
    const float rev = (float)(int)ph * (1.0f / 4294967296.0f);       // ph = the LO's 32-bit phase at the output's newest sample
    const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
    float2 o;
    o.x = t * cs - other * sn;
    o.y = t * sn + other * cs;
    ys[rx * ycap + io] = o;
  }
}

// The dot products of one task on the matrix cores: 16 blocks = 16 tap residues, rows = the quad's outputs, columns = (Re, Im) of
// an RX pair; xr = this lane's LDS pointer (block b, row i: x[n_i - b - 16 (NJ - 1)], so that step jj reads element 16 (NJ - 1 - jj)).
// G = 1: the Re x chain and the Im x chain have an accumulator each (a chain on ONE accumulator issues every 15 cycles instead of
// 8: scripts/diag/mfma4x4_probe.hip); G >= 2: the pairs alternate, one accumulator per pair is enough.
// The reads of x go through a RING of kXA register pairs, kXA steps ahead of the MFMAs that use them, pinned by
// sched_group_barrier: left alone hipcc reuses four registers and puts a full s_waitcnt lgkmcnt(0) in front of every four MFMAs --
// one exposed LDS latency per 40 cycles of matrix work (the same finding as mixdec_mfma.hip's consumer).
template <int G, int NJ>
__device__ __forceinline__ void mm_task_dots(lds_cf2 xr, const float (&bre)[G][NJ], const float (&bim)[G][NJ], md_f4 (&acc)[G]) {
  constexpr int kNA = (G == 1 || !MD_MM_ONEACC) ? 2 * G : G;
  md_f4 accs[kNA];
#pragma unroll
  for (int q = 0; q < kNA; ++q) accs[q] = (md_f4){0.f, 0.f, 0.f, 0.f};
  constexpr int kTop = 16 * (NJ - 1);
  constexpr int kXA = (MD_XAHEAD < NJ) ? MD_XAHEAD : NJ;
  float2 ring[kXA];
#pragma unroll
  for (int u = 0; u < kXA; ++u) ring[u] = lds_ld(xr, kTop - 16 * u);
  __builtin_amdgcn_sched_group_barrier(0x100, kXA, 0);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const float2 xv = ring[jj % kXA];
    if constexpr (kNA == 2 * G) {
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        accs[2 * gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.x, bre[gq][jj], accs[2 * gq], 0, 0, 0);
        accs[2 * gq + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.y, bim[gq][jj], accs[2 * gq + 1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int gq = 0; gq < G; ++gq) accs[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.x, bre[gq][jj], accs[gq], 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < G; ++gq) accs[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.y, bim[gq][jj], accs[gq], 0, 0, 0);
    }
    if (jj + kXA < NJ) ring[jj % kXA] = lds_ld(xr, kTop - 16 * (jj + kXA));
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * G, 0);
    if (jj + kXA < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  }
  if constexpr (kNA == 2 * G) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) acc[gq] = accs[2 * gq] + accs[2 * gq + 1];
  } else {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) acc[gq] = accs[gq];
  }
}

// Vector form of a task's dot products with the taps in registers, for the steady runs of the one vector shape that has them
// (<1,11>: 1001 taps at UP = 6, one sub-receiver -- FT8:42,70, 1 MS/s, eight tasks per wave and tile): every read of x up front.
template <int RH, int NJ>
__device__ __forceinline__ void vec_task_dots(lds_cf2 xr, const float2 (&greg)[RH][NJ], float2 (&A)[RH], float2 (&B)[RH]) {
  constexpr int kTop = 16 * (NJ - 1);
  float2 xg[NJ];
#pragma unroll
  for (int u = 0; u < NJ; ++u) xg[u] = lds_ld(xr, kTop - 16 * u);
#pragma unroll
  for (int r = 0; r < RH; ++r) { A[r] = make_float2(0.f, 0.f); B[r] = make_float2(0.f, 0.f); }
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    const float2 xv = xg[u];
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      const float2 gg = greg[r][u];
      A[r].x = fmaf(gg.x, xv.x, A[r].x);
      A[r].y = fmaf(gg.y, xv.x, A[r].y);
      B[r].x = fmaf(gg.x, xv.y, B[r].x);
      B[r].y = fmaf(gg.y, xv.y, B[r].y);
    }
  }
}

template <int R, int NJ, int TPB, int NHX, int MM>
__global__ __launch_bounds__(TPB) void mixdec_kernel(const MixDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  float2* const buf0 = lds;                    // [tile_cap]
  float2* const buf1 = lds + a.tile_cap;       // [tile_cap]
  float2* const tl = lds + 2 * a.tile_cap;     // [R][up][kpad] (taps_lds) or nothing (the waves hold their taps, read from memory)
  float2* const ys = tl + (a.taps_lds ? R * a.up * a.kpad : 0);   // [R][ycap] output stage

  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthr >> 6;   // wave-uniform: SALU
  const int lane = tid & 63;
  // vector form: g = DPP row (the output of a quad), s = lane within the row (tap residue k mod 16);
  // matrix-core form (MdShape::kMm): lane 4 b + i = block b (tap residue) and row i of the block (the output of the quad)
  const int g = MM ? (lane & 3) : (lane >> 4), s = MM ? (lane >> 2) : (lane & 15);

  // contiguous run of tiles for this workgroup
  const int ng = gridDim.x;
  const int base = a.ntiles / ng, rem = a.ntiles % ng;
  const int wb = blockIdx.x;
  const int t_begin = wb * base + (wb < rem ? wb : rem);
  const int t_end = t_begin + base + (wb < rem ? 1 : 0);
  if (t_begin >= t_end) return;

  // ---- stage the LO-modulated taps once
  if (!a.taps_lds) {
    // (hold mode guaranteed by the host, mixdec_variant(): every wave reads its own taps straight from memory below)
  } else if (a.aligned16) {
    const int nt4 = (R * a.up * a.kpad) >> 1;                   // taps as 16-B slots
    const float4* src = reinterpret_cast<const float4*>(a.taps);
    float4* dst = reinterpret_cast<float4*>(tl);
    for (int q = wave; q * 64 < nt4; q += nwaves) {
      const int slot = q * 64 + lane;
      if (slot < nt4) glds16(src + slot, dst + q * 64);
    }
  } else {
    const int nt = R * a.up * a.kpad;
    for (int i = tid; i < nt; i += nthr) tl[i] = a.taps[i];
  }
  // Taps in registers: when tile_out is a multiple of UP, output i of every tile is on branch
  // (p_f + i*DOWN) mod UP with the same p_f, so a wave that only ever works on one branch can
  // read its taps from LDS once and keep them in VGPRs for the whole launch (24 of the 30 LDS
  // reads of a C3 task disappear).  For that the tasks are dealt out by (branch, RX half): wave w
  // belongs to group w % (UP*NH), works on that group's branch and -- above 4 RX -- on one half
  // of the sub-receivers only, and walks the quads with stride nwaves / (UP*NH).
  typedef MdShape<R, NJ, TPB, NHX, MM> Sh;
  constexpr bool kMm = Sh::kMm;
  constexpr bool kEpConst = (MD_MM_EPCONST < 0) ? (R >= 3) : (MD_MM_EPCONST != 0);
  constexpr int NH = Sh::NH;                            // RX groups (halves at 1024 threads)
  constexpr int RH = Sh::RH;                            // RX per task in hold mode
  constexpr bool kCanHold = Sh::kCanHold;
  const int ngrp = a.up * NH;
  const bool hold = kCanHold && ngrp <= nwaves && (a.tile_out % a.up) == 0;
  const int hold_grp = wave % ngrp;
  const int hold_c = hold_grp % a.up;                   // this wave's branch
  const int hold_rbase = hold ? (hold_grp / a.up) * RH : 0;   // first RX of this wave's tasks
  const int hold_rcount = hold ? ((R - hold_rbase < RH) ? R - hold_rbase : RH) : R;
  const int hold_step = nwaves / ngrp;                  // waves per group
  const int hold_q0 = (wave < ngrp * hold_step) ? wave / ngrp : (1 << 29);
  float2 greg[(kCanHold && !kMm) ? RH : 1][(kCanHold && !kMm) ? NJ : 1];
  // matrix-core form: the B operands of the two chains, lane 4 b + j = tap residue b, column j of the RX pair
  constexpr int kG = kMm ? Sh::G : 1;
  float bre[kG][kMm ? NJ : 1], bim[kG][kMm ? NJ : 1];
  // per-lane constants of the epilogue: lane s of every row finishes RX hold_rbase + s
  // (kTight: instantiations whose tap registers leave no room -- 2*RH*NJ >= 40 of the 128 a
  // 1024-thread workgroup may use -- recompute these four per task instead: a spilled one is reloaded
  // with a scratch load, and the s_waitcnt vmcnt(0) behind it also waits for the NEXT tile's copies,
  // i.e. serialises the DMA with the dot products: mixdec<1,21> ran at 0.36 of HBM for that reason)
  constexpr bool kTight = !kMm && kCanHold && (2 * RH * NJ >= Sh::kBudget - 88) && (R != 4 || TPB != 1024);
  uint32_t my_p0 = 0u, my_fw = 0u;
  int v_gup = g * a.up, v_gdown = g * a.down - s;
  if (kMm) {
    if (!kEpConst) {     // the lane of the matrix-core epilogue: row rho = lane >> 4 finishes output rho, (pair q, column j) RX 2 q + j / 2
      const int rx = 2 * ((lane >> 2) & 3) + ((lane & 3) >> 1);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (rx == r) { my_p0 = a.phase0[r]; my_fw = a.fword[r]; }
      asm volatile("" : "+v"(my_p0), "+v"(my_fw));
    }
    asm volatile("" : "+v"(v_gup), "+v"(v_gdown));
  } else if (!kTight) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (hold_rbase + s == r) { my_p0 = a.phase0[r]; my_fw = a.fword[r]; }
    asm volatile("" : "+v"(my_p0), "+v"(my_fw), "+v"(v_gup), "+v"(v_gdown));
  }
  Tile cur = tile_geometry(a, t_begin);
  int i_base = cur.i_first;      // first output held in the LDS output stage
  float pk_run = 0.f;            // running raw-chunk peak of chunk pk_chunk (per lane)
  uint32_t pk_chunk = 0u;
  int pk_lo = 0;                 // samples [pk_lo, pk_hi] of chunk pk_chunk exist in this call
  int pk_hi = (int)(a.chunk_len < a.n_total ? a.chunk_len : a.n_total) - 1;
  if (!PYSDR_DBG(a, 2)) stage_tile(a, cur, buf0, tid, nthr);
  // the decimator's history roll, by workgroup 0 while its first tile's copies are in flight (hist_roll.h)
  if (blockIdx.x == 0 && a.hist_new != nullptr)
    roll_history(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n, tid, nthr);

  if (kCanHold && hold) {
    // this wave's taps: one branch (p_f is the same for every tile: tile_out*DOWN is a multiple of UP),
    // one RX half, read from LDS once for the whole launch
    if (a.taps_lds) {
      dma_wait();
      __syncthreads();
    }
    uint32_t qc0, pc0;
    divmod_magic((uint32_t)cur.p_f + (uint32_t)hold_c * (uint32_t)a.down, (uint32_t)a.up, a.magic, qc0, pc0);
    const int kp0 = (NJ > 0) ? 16 * NJ : a.kpad;
    const float2* th = (a.taps_lds ? tl : a.taps) + (int)pc0 * kp0 + s + hold_rbase * a.up * kp0;
    if constexpr (kMm) {
      // column j of pair gq: RX 2 gq + j / 2, part j & 1.  y = sum g x:  Re x chain against [Re g | Im g], Im x chain against [-Im g | Re g]
      const int j = lane & 3;
#pragma unroll
      for (int gq = 0; gq < kG; ++gq) {
        const int rx = 2 * gq + (j >> 1);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const float2 gg = (rx < R) ? th[rx * a.up * kp0 + 16 * jj] : make_float2(0.f, 0.f);
          bre[gq][jj] = (j & 1) ? gg.y : gg.x;
          bim[gq][jj] = (j & 1) ? gg.x : -gg.y;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < RH; ++r)
#pragma unroll
        for (int jj = 0; jj < (kCanHold ? NJ : 1); ++jj) {
          greg[r][jj] = (r < hold_rcount) ? th[r * a.up * kp0 + 16 * jj] : make_float2(0.f, 0.f);
        }
    }
  }

  int flush_in = a.yflush;       // tiles until the output stage is flushed (a countdown: `(tb - t_begin + 1) % a.yflush` was a 32-bit
                                 // division by a run-time value, ~25 scalar + 6 vector instructions per tile and wave)
  for (int tb = t_begin; tb < t_end; ++tb) {
    // ---- STEADY RUNS (matrix-core shapes).  When tile_out is a multiple of UP every full interior tile is the one before it
    // shifted by dq = tile_out DOWN / UP samples: its image, the samples it owns, where every lane of every task reads and where
    // its outputs go are arithmetic progressions.  The generic body below spends ~230 scalar + ~280 vector instructions per
    // wave and tile on re-deriving them (SQ_INSTS_SALU / _VALU, profiles/r06_ft8tri_pmc_mfma.json) beside 84 MFMAs and 21 LDS
    // reads of real work -- twelve waves of that are what the kernel's issue slots go to, not its memory traffic.  So: the
    // longest stretch of tiles from here that are full, not at either end of the call or of this workgroup's share, copied whole
    // (LDS-DMA pieces inside the call) and owned by ONE chunk (fast peak) is found by two divisions, its per-lane constants are
    // set up once, and its tiles run a loop that only adds.  Same reads, same MFMA chains, same block reduction, same phases:
    // bit for bit the generic body's results (tests: every cut-independence test crosses both paths; MD_STEADY=0 for the A/B).
    constexpr bool kSteadyVec = !kMm && kCanHold && R == 1 && NJ == 11 && MD_STEADY_VEC1;     // ... and one vector shape, below
    if constexpr ((kMm && MD_STEADY) || kSteadyVec) {
      int run = 0;
      const int dq = a.dq_tile;
      const int npieces_s = (cur.npairs + 63) >> 6;
      const int e_lo0 = cur.own_lo & ~1, e_hi0 = cur.own_hi | 1;
      const int ntask_w = (hold_q0 < a.tpc) ? ((a.tpc - hold_q0 + hold_step - 1) / hold_step) : 0;   // this wave's tasks per tile
      if (kCanHold && hold && a.aligned16 && a.dr_tile == 0 && (dq & 1) == 0 && tb > 0 && cur.tile_n == a.tile_out && a.dbg == 0 &&
          e_lo0 >= pk_lo && e_hi0 <= pk_hi && cur.lo + dq >= 0) {
        // tiles t = tb .. tb + run - 1: t + 2 < ntiles, t + 1 < t_end, the copy of t + 1 inside the call, the peak of t inside the chunk
        int lim = min(a.ntiles - 2, t_end - 1) - tb;
        const long long room = (long long)a.n_total - 128LL * npieces_s - (long long)(cur.lo + dq);
        if (room < 0) lim = 0;
        else lim = min(lim, 1 + (int)(room / dq));
        lim = min(lim, 1 + (pk_hi - e_hi0) / dq);
        run = lim;
      }
      if (run >= 2) {
        // ---- per-run constants
        const int d0 = cur.rel_f - cur.lo;                      // tap 0 of the tile's first output, inside its image (the same for every tile of the run)
        uint32_t qcr, pcr;
        divmod_magic((uint32_t)cur.p_f + (uint32_t)hold_c * (uint32_t)a.down, (uint32_t)a.up, a.magic, qcr, pcr);
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int rho = lane_l >> 4;
        uint32_t ep_p0 = 0u, ep_fw = 0u;
        {
          // the lane that finishes a sample: matrix-core form (pair q, column j) of row rho; vector form lane s of row rho = RX rbase + s
          const int rx = kMm ? 2 * ((lane_l >> 2) & 3) + ((lane_l & 3) >> 1) : hold_rbase + (lane_l & 15);
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (rx == r) { ep_p0 = a.phase0[r]; ep_fw = a.fword[r]; }
        }
        const int vgd = kMm ? v_gdown : (lane_l >> 4) * a.down - (lane_l & 15);   // (vector shapes that recompute it per task: here per run)
        constexpr int kTopS = 16 * ((NJ > 0 ? NJ : 1) - 1);
        // this wave's tasks of a tile are quads hold_q0, hold_q0 + hold_step, ...: one more arithmetic progression, so the run keeps
        // the first task's per-lane constants and the step from task to task (any number of tasks per wave: 1 at 8 MS/s, 8 at 1 MS/s)
        const int qq0 = (ntask_w > 0) ? hold_q0 : 0;            // (waves without tasks: hold_q0 = 1 << 29)
        const int sb0 = d0 + (int)qcr + 4 * qq0 * a.down;
        const int xoff0 = (sb0 + vgd - kTopS) * 8;              // bytes from the image's first sample: this lane's lowest read
        const int ioff0 = hold_c + 4 * qq0 * a.up + rho * a.up;
        uint32_t ph0 = ep_p0 + ep_fw * (uint32_t)(sb0 + cur.lo + rho * a.down);
        const uint32_t dph = ep_fw * (uint32_t)dq;              // ... per tile
        const int dxo = 4 * hold_step * a.down * 8, dio = 4 * hold_step * a.up;      // ... per task (scalar)
        const uint32_t dph_t = ep_fw * (uint32_t)(4 * hold_step * a.down);
        // peak: pair indices of this thread inside the image
        const int pk_p0 = ((e_lo0 - cur.lo) >> 1) + tid, pk_phi = (e_hi0 - cur.lo) >> 1;
        // copies: this wave's pieces of an image, 1 KiB each: SGPR base + per-lane offset
        const unsigned voff0 = (unsigned)(wave * 1024 + lane * 16);
        const unsigned lds_b0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)buf0;
        const char* src_next = reinterpret_cast<const char*>(a.x + (cur.lo + dq));      // image of tile tb + 1
        int i_first_s = cur.i_first;
        for (int k = 0; k < run; ++k) {
          const int io_base = i_first_s - i_base;
          const int par = (tb + k - t_begin) & 1;
          const unsigned tb8 = (unsigned)a.tile_cap * 8u;
          const unsigned xs_b = lds_b0 + (par ? tb8 : 0u), xn_b = lds_b0 + (par ? 0u : tb8);
          const float4* const xs_v = (const float4*)(const __attribute__((address_space(3))) float4*)(size_t)xs_b;
          dma_wait();
          __syncthreads();
          const int ord_s = (kMm && MD_PHASE_ORDERS > 1) ? ((MD_PHASE_ORDERS == 2) ? ((wave >> 2) & 1) : (wave >> 2) % 3) : 0;
          if (ord_s != 1) steady_stage(src_next, voff0, xn_b, npieces_s, wave, nwaves);
          if (ord_s == 0) pk_run = steady_peak(xs_v, pk_p0, pk_phi, nthr, pk_run);
          {
            int xo = xoff0, io = io_base + ioff0;
            uint32_t pht = ph0;
            for (int u = 0; u < ntask_w; ++u) {
              if constexpr (kMm) {
                md_f4 acc[kG];
                mm_task_dots<kG, (kMm ? NJ : 1)>((lds_cf2)(size_t)(xs_b + (unsigned)xo), bre, bim, acc);
                mm_fold_rotate_stage<kG, R>(acc, lane_l, true, pht, ys, a.ycap, io);
              } else {
                float2 A[RH], B[RH];
                vec_task_dots<RH, (kCanHold && !kMm) ? NJ : 1>((lds_cf2)(size_t)(xs_b + (unsigned)xo), greg, A, B);
                fold_rotate_stage<RH>(A, B, hold_rcount, hold_rbase, lane_l & 15, true, 0u, pht, 0u, ys, a.ycap, io);
              }
              xo += dxo; io += dio; pht += dph_t;
            }
            ph0 += dph;
          }
          if (ord_s == 1) steady_stage(src_next, voff0, xn_b, npieces_s, wave, nwaves);
          if (ord_s != 0) pk_run = steady_peak(xs_v, pk_p0, pk_phi, nthr, pk_run);
          // flush the output stage (the generic body's, with this tile's counters)
          if (--flush_in == 0) {
            flush_in = a.yflush;
            __syncthreads();
            const int n_st = i_first_s + a.tile_out - i_base;
            const int cpr = (n_st + 63) >> 6;
            for (int ch = wave; ch < R * cpr; ch += nwaves) {
              int r = 0, c2 = ch;
              while (c2 >= cpr) { c2 -= cpr; ++r; }
              const int j = c2 * 64 + lane;
              if (j < n_st) {
                typedef float md_f2 __attribute__((ext_vector_type(2)));
                const float2 v = ys[r * a.ycap + j];
                __builtin_nontemporal_store((md_f2){v.x, v.y}, (md_f2*)(a.y[r] + i_base + j));
              }
            }
            i_base = i_first_s + a.tile_out;
          }
          i_first_s += a.tile_out;
          src_next += (size_t)dq * 8;
        }
        // ---- back to the generic body with the geometry of tile tb + run (its image was copied by the run's last tile): a full
        // tile that is not the call's last, `run` steps of dq further on
        {
          const int sh = run * dq;
          cur.i_first += run * a.tile_out;
          cur.rel_f += sh; cur.rel_l += sh; cur.own_lo += sh; cur.own_hi += sh; cur.lo += sh; cur.hi += sh;
        }
        tb += run - 1;
        continue;
      }
    }
    float2* const xs = ((tb - t_begin) & 1) ? buf1 : buf0;
    float2* const xn = ((tb - t_begin) & 1) ? buf0 : buf1;
    // tile tb has landed; after the barrier everybody is also done reading the other
    // buffer (tile tb-1), so it can be refilled while we compute.
    PYSDR_STAMP(0);
    dma_wait();
    PYSDR_STAMP(1);
    __syncthreads();
    PYSDR_STAMP(2);
    Tile nxt = cur;
    // Matrix-core shapes: the three waves that share a SIMD (w, w + 4, w + 8) would run their MFMA chains at the same time, and
    // the matrix pipe then idle through everybody's copy / peak phases (scripts/diag/mixdec_stamps.py ft8tri,
    // profiles/r06_ft8tri_stamps.txt: the dot-product phase of a task took 2800 cycles where 84 MFMAs alone take ~900).
    // Each of the three takes the phases of a tile in its own order:  0: copies, peak, DOTS   1: DOTS, copies, peak
    // 2: copies, DOTS, peak   (MD_PHASE_ORDERS = 1: all of them order 0, 2: orders 0 1 0; A/B)
    const int ord = (kMm && MD_PHASE_ORDERS > 1) ? ((MD_PHASE_ORDERS == 2) ? ((wave >> 2) & 1) : (wave >> 2) % 3) : 0;
    const bool have_next = tb + 1 < t_end;
    if (have_next) {
      nxt = (tb + 2 < a.ntiles) ? tile_advance(a, cur) : tile_geometry(a, tb + 1);
      if constexpr (!kMm) {
        if (!PYSDR_DBG(a, 2)) stage_tile(a, nxt, xn, tid, nthr);
      }
    }
    if constexpr (kMm) {
      if (ord != 1 && have_next && !PYSDR_DBG(a, 2)) stage_tile(a, nxt, xn, tid, nthr);
      if (ord == 0 && !PYSDR_DBG(a, 4)) peak_scan(a, cur, xs, tid, nthr, lane, pk_run, pk_chunk, pk_lo, pk_hi);
    }
    PYSDR_STAMP(3);

    // ---- raw-chunk peak |x|^2 over the samples this tile owns (rx.auto_mute input).
    // The running maximum of a chunk stays in a register across tiles; the atomic is only
    // issued when the run moves on to another chunk (and once at the end): same-address
    // atomics are slow, and one per tile would sit in vmcnt and stall the next dma_wait.
    if (!kMm && cur.own_hi >= cur.own_lo && !PYSDR_DBG(a, 4)) {
      const float4* xv = reinterpret_cast<const float4*>(xs);
      const int e_lo = cur.own_lo & ~1, e_hi = cur.own_hi | 1;
      if (e_lo >= pk_lo && e_hi <= pk_hi) {
        // whole pairs inside the current chunk: a maximum does not mind the neighbour
        // sample being counted by two tiles
        const int p_hi = (e_hi - cur.lo) >> 1;
        int pi = ((e_lo - cur.lo) >> 1) + tid;
        // four reads in flight per thread: a tile is 4-5 trips of this loop, and one read per trip
        // put 4-5 LDS latencies in front of every tile's dot products (0.04 of C1's 0.43 ms)
        for (; pi + 3 * nthr <= p_hi; pi += 4 * nthr) {
          const float4 v0 = xv[pi], v1 = xv[pi + nthr], v2 = xv[pi + 2 * nthr], v3 = xv[pi + 3 * nthr];
          const float m0 = fmaxf(fmaf(v0.x, v0.x, v0.y * v0.y), fmaf(v0.z, v0.z, v0.w * v0.w));
          const float m1 = fmaxf(fmaf(v1.x, v1.x, v1.y * v1.y), fmaf(v1.z, v1.z, v1.w * v1.w));
          const float m2 = fmaxf(fmaf(v2.x, v2.x, v2.y * v2.y), fmaf(v2.z, v2.z, v2.w * v2.w));
          const float m3 = fmaxf(fmaf(v3.x, v3.x, v3.y * v3.y), fmaf(v3.z, v3.z, v3.w * v3.w));
          pk_run = fmaxf(fmaxf(pk_run, fmaxf(m0, m1)), fmaxf(m2, m3));
        }
        for (; pi <= p_hi; pi += nthr) {
          const float4 v = xv[pi];
          pk_run = fmaxf(pk_run, fmaxf(fmaf(v.x, v.x, v.y * v.y), fmaf(v.z, v.z, v.w * v.w)));
        }
      } else {
        // the tile straddles chunk boundaries (or the odd end of the call): one masked scan
        // per chunk it touches
        const uint32_t c_lo = div_magic((uint32_t)cur.own_lo, a.chunk_len, a.magic_chunk);
        const uint32_t c_hi = div_magic((uint32_t)cur.own_hi, a.chunk_len, a.magic_chunk);
        for (uint32_t c = c_lo; c <= c_hi; ++c) {
          const long long cb = (long long)c * a.chunk_len;
          const int s_lo = cur.own_lo > cb ? cur.own_lo : (int)cb;
          const long long ce = cb + a.chunk_len - 1;
          const int s_hi = cur.own_hi < ce ? cur.own_hi : (int)ce;
          const int p_lo = (s_lo - cur.lo) >> 1, p_hi = (s_hi - cur.lo) >> 1;
          if (c != pk_chunk) {
            pk_run = wave_max63(pk_run);
            if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
            pk_run = 0.f;
            pk_chunk = c;
          }
          // interior pairs need no masking; the two edge pairs are handled by one lane
          for (int pi = p_lo + 1 + tid; pi < p_hi; pi += nthr) {
            const float4 v = xv[pi];
            pk_run = fmaxf(pk_run, fmaxf(fmaf(v.x, v.x, v.y * v.y), fmaf(v.z, v.z, v.w * v.w)));
          }
          if (tid == 0) {
            const float4 v0 = xv[p_lo], v1 = xv[p_hi];
            const int r0 = cur.lo + 2 * p_lo, r1 = cur.lo + 2 * p_hi;
            if (r0 >= s_lo) pk_run = fmaxf(pk_run, fmaf(v0.x, v0.x, v0.y * v0.y));
            if (r0 + 1 <= s_hi) pk_run = fmaxf(pk_run, fmaf(v0.z, v0.z, v0.w * v0.w));
            if (r1 >= s_lo) pk_run = fmaxf(pk_run, fmaf(v1.x, v1.x, v1.y * v1.y));
            if (r1 + 1 <= s_hi) pk_run = fmaxf(pk_run, fmaf(v1.z, v1.z, v1.w * v1.w));
          }
        }
        const long long cb = (long long)pk_chunk * a.chunk_len;
        const long long ce = cb + a.chunk_len < (long long)a.n_total ? cb + a.chunk_len : (long long)a.n_total;
        pk_lo = (int)cb;
        pk_hi = (int)ce - 1;
      }
    }

    PYSDR_STAMP(4);
    // ---- polyphase dot products: one output per DPP row (16 lanes), four outputs of the
    // same polyphase branch per wave.  Task = (branch c, quad qq): outputs
    // i = i_first + c + UP*(4*qq + g), whose input index is rel_c + DOWN*(4*qq + g) with
    // (rel_c, p_c) = divmod(t0 + (i_first + c)*DOWN, UP) = (rel_f, 0) + divmod(p_f + c*DOWN, UP):
    // one small scalar division per task, one VALU add per lane.
    const int upc = a.up;
    const int kp = (NJ > 0) ? 16 * NJ : a.kpad;
    const int ntasks = (cur.tile_n > 0 && !PYSDR_DBG(a, 1)) ? a.ntasks : 0;
    const int i_last = cur.i_first + cur.tile_n - 1;
    // generic order: task = (branch, quad) round robin over the waves; hold order: see above
    const int t_lim = hold ? (ntasks > 0 ? a.tpc : 0) : ntasks;
    const int t_step = hold ? hold_step : nwaves;
    for (int task = hold ? hold_q0 : wave; task < t_lim; task += t_step) {
      const int c = hold ? hold_c : ((a.tpc == 1) ? task : (int)__umulhi((uint32_t)task, a.magic_tpc));
      const int qq = hold ? task : task - c * a.tpc;
      uint32_t qc, pc;
      divmod_magic((uint32_t)cur.p_f + (uint32_t)c * (uint32_t)a.down, (uint32_t)upc, a.magic, qc, pc);
      int s_l = s;                                        // (laundered: hipcc would hoist the recomputation
      if (kTight) {                                       //  out of the loop and spill it again)
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        s_l = lane_l & 15;
        v_gup = (lane_l >> 4) * a.up;
        v_gdown = (lane_l >> 4) * a.down - s_l;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (hold_rbase + s_l == r) { my_p0 = a.phase0[r]; my_fw = a.fword[r]; }
      }
      const int i = cur.i_first + c + 4 * qq * upc + v_gup;
      const bool valid = (i <= i_last);
      const int sb = cur.rel_f + (int)qc + 4 * qq * a.down - cur.lo;      // scalar
      const int xi = valid ? sb + v_gdown : (cur.rel_f - cur.lo) - s;      // x index of tap 0, minus s
      const uint32_t rel = (uint32_t)(sb + cur.lo + v_gdown + s);
      const float2* xp = xs + xi;
      const float2* tp = tl + (int)pc * kp + s;
      // A += g*x.re, B += g*x.im per RX (two packed FMAs per tap, no operand shuffles);
      // y = (A.re - B.im, A.im + B.re)
      if (kCanHold && hold && kMm) {
        // ---- matrix cores (mm_task_dots)
        constexpr int kTop = 16 * ((NJ > 0 ? NJ : 1) - 1);
        md_f4 acc[kG];
        mm_task_dots<kG, (kMm ? NJ : 1)>(to_lds(xp - kTop), bre, bim, acc);
        // the epilogue's lane finishes output rho = lane >> 4 of the quad (the reads above were for output lane & 3)
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int rho = lane_l >> 4;
        const int i_ep = cur.i_first + c + 4 * qq * upc + rho * upc;
        const uint32_t rel_ep = (uint32_t)(sb + cur.lo + rho * a.down);
        // (its RX's LO constants are looked up here, per task: kept in registers across the tile loop they were what spilled at 4 RX)
        uint32_t ep_p0 = my_p0, ep_fw = my_fw;
        if (kEpConst) {
          const int rx = 2 * ((lane_l >> 2) & 3) + ((lane_l & 3) >> 1);
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (rx == r) { ep_p0 = a.phase0[r]; ep_fw = a.fword[r]; }
        }
        mm_fold_rotate_stage<kG, R>(acc, lane_l, i_ep <= i_last, ep_p0 + ep_fw * rel_ep, ys, a.ycap, i_ep - i_base);
      } else if (kCanHold && hold) {
        float2 A[RH], B[RH];
#pragma unroll
        for (int r = 0; r < RH; ++r) { A[r] = make_float2(0.f, 0.f); B[r] = make_float2(0.f, 0.f); }
        // (reads go through the LOWEST address + a non-negative offset: a DS offset field is unsigned,
        //  so xp[-16*jj] cost one VALU address add per read)
        constexpr int kTop = 16 * ((NJ > 0 ? NJ : 1) - 1);
        const lds_cf2 xr = to_lds(xp - kTop);
        // Where the taps leave few registers (three RX x 21 tap pairs = 126 of 168) the reads of x come in GROUPS of kXG with a
        // compiler fence between them: left alone hipcc hoists all 21 reads to the top (42 more registers) and spills.
        constexpr int kNJ = kCanHold ? NJ : 1;
        constexpr int kXG = (Sh::kBudget - 2 * RH * NJ < 64) ? MD_XGROUP : kNJ;
#pragma unroll
        for (int j0 = 0; j0 < kNJ; j0 += kXG) {
          float2 xg[kXG];
#pragma unroll
          for (int u = 0; u < kXG; ++u)
            if (j0 + u < kNJ) xg[u] = lds_ld(xr, kTop - 16 * (j0 + u));
#pragma unroll
          for (int u = 0; u < kXG; ++u) {
            if (j0 + u >= kNJ) continue;
            const float2 xv = xg[u];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
              const float2 gg = greg[r][j0 + u];
              A[r].x = fmaf(gg.x, xv.x, A[r].x);
              A[r].y = fmaf(gg.y, xv.x, A[r].y);
              B[r].x = fmaf(gg.x, xv.y, B[r].x);
              B[r].y = fmaf(gg.y, xv.y, B[r].y);
            }
          }
          if (kXG < kNJ && j0 + kXG < kNJ) asm volatile("" ::: "memory");
        }
        fold_rotate_stage<RH>(A, B, hold_rcount, hold_rbase, s, valid, rel, my_p0, my_fw, ys, a.ycap, i - i_base);
      } else {
        float2 A[R], B[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { A[r] = make_float2(0.f, 0.f); B[r] = make_float2(0.f, 0.f); }
        const lds_cf2 xlow = to_lds(xp - (kp - 16));
        auto tap_step = [&](int j) {
          const float2 xv = lds_ld(xlow, kp - 16 - j);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float2 gg = tp[r * upc * kp + j];
            A[r].x = fmaf(gg.x, xv.x, A[r].x);
            A[r].y = fmaf(gg.y, xv.x, A[r].y);
            B[r].x = fmaf(gg.x, xv.y, B[r].x);
            B[r].y = fmaf(gg.y, xv.y, B[r].y);
          }
        };
        if (NJ > 0) {
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) tap_step(16 * jj);
        } else {
#pragma unroll 2
          for (int j = 0; j < kp; j += 16) tap_step(j);
        }
        fold_rotate_stage<R>(A, B, R, 0, s, valid, rel, my_p0, my_fw, ys, a.ycap, i - i_base);
      }
    }
    if constexpr (kMm) {
      if (ord == 1 && have_next && !PYSDR_DBG(a, 2)) stage_tile(a, nxt, xn, tid, nthr);
      if (ord != 0 && !PYSDR_DBG(a, 4)) peak_scan(a, cur, xs, tid, nthr, lane, pk_run, pk_chunk, pk_lo, pk_hi);
    }
    PYSDR_STAMP(5);
    // ---- flush the output stage: RX r, 64 outputs per wave-store (512 contiguous bytes)
    if (tb + 1 == t_end || --flush_in == 0) {
      flush_in = a.yflush;
      __syncthreads();
      const int n_st = cur.i_first + cur.tile_n - i_base;
      const int cpr = (n_st + 63) >> 6;
      if (!PYSDR_DBG(a, 8))
        for (int ch = wave; ch < R * cpr; ch += nwaves) {
          int r = 0, c2 = ch;
          while (c2 >= cpr) { c2 -= cpr; ++r; }
          const int j = c2 * 64 + lane;
#ifdef MD_Y_PLAIN                     // A/B: plain stores
          if (j < n_st) a.y[r][i_base + j] = ys[r * a.ycap + j];
#else
          if (j < n_st) {
            typedef float md_f2 __attribute__((ext_vector_type(2)));
            const float2 v = ys[r * a.ycap + j];
            __builtin_nontemporal_store((md_f2){v.x, v.y}, (md_f2*)(a.y[r] + i_base + j));
          }
#endif
        }
      i_base = cur.i_first + cur.tile_n;
    }
    PYSDR_STAMP(6);
    cur = nxt;
  }
  pk_run = wave_max63(pk_run);
  if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
}

template <int R, int NJ, int TPB, int NHX, int MM>
int launch_rj(const MixDecArgs& a, int threads, int grid, size_t lds, hipStream_t st) {
  // the attribute is per (function, device): one bit per device, guarded against contexts on
  // other threads / other devices of the same process (P.GPU_DEVICE, cfg.device)
  static std::mutex attr_mu;
  static uint64_t attr_done = 0;
  {
    int dev = 0;
    PYSDR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mixdec_kernel<R, NJ, TPB, NHX, MM>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        set_last_error("hipFuncSetAttribute(mixdec<%d,%d,%d,%d,%d>): %s", R, NJ, TPB, NHX, MM, hipGetErrorString(e));
        return PYSDR_ERR_HIP;
      }
      attr_done |= 1ull << (dev & 63);
    }
  }
  if (threads > TPB) threads = TPB;
  hipLaunchKernelGGL((mixdec_kernel<R, NJ, TPB, NHX, MM>), dim3(grid), dim3(threads), lds, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

// Which instantiation a decimator's shape runs on -- ONE table for the launch and for the host's plan (mixdec_variant).
//   f.template go<R, NJ, TPB, NHX, MM>()
template <int R, class F>
int md_dispatch_r(int up, int kpad, int threads, F& f) {
  // 255-tap prototypes at UP = 3 (the BASELINE configurations) have 96 taps per branch
  if (kpad == 96) return f.template go<R, 6, 1024, 0, 0>();
  // the default 1001-tap prototype at UP = 6 (1, 5, 7 MS/s -> 48 kHz: FT8:42, FT8FT4:34, FT8dual:43): 167 taps per branch, eleven
  // tap pairs per lane held in registers by six groups of waves (one sub-receiver; two with MD_UP6_MM > 2) ...
#if MD_UP6_MM
  // ... and on the matrix cores (12 waves, two per branch) from MD_UP6_MM sub-receivers
  if constexpr (R >= MD_UP6_MM && R <= 6) {
    if (kpad == 176 && up == 6 && threads == 1024) return f.template go<R, 11, 768, 0, 1>();
  }
#endif
#if MD_UP6_HOLD
  if constexpr (R <= 2 && (MD_UP6_MM == 0 || R < MD_UP6_MM)) {
    if (kpad == 176 && up == 6 && threads == 1024) return f.template go<R, 11, 1024, 0, 0>();
  }
#endif
  // single-RX long filters: the 255-tap video filter of the broadcast-FM front end (UP = 1, 256
  // taps in one branch) and the reference's default 1001-tap prototype at UP = 3 (336 per branch)
  if constexpr (R == 1) {
    // the fs1 -> FS_OUT resampler of broadcast FM: 24/125 with 64 taps per branch (more branches than waves: generic
    // task order, but a compile-time tap loop)
    if (kpad == 64) return f.template go<R, 4, 1024, 0, 0>();
    if (kpad == 256) return f.template go<R, 16, 1024, 0, 0>();
    if (kpad == 336) return f.template go<R, 21, 1024, 0, 0>();
  }
  // the default 1001-tap prototype at UP = 3 with several sub-receivers (8 and 4 MS/s -> 48 kHz: FT8tri, TEST): 12 waves that
  // hold their taps.  Only with the default thread count: pysdr_set_tile(threads) asks for the generic form (A/B).
  if constexpr (R >= 2 && R <= 6) {
    if (kpad == 336 && up == 3 && threads == 1024) {
#if MD_LONG_MM
      // 2 - 4 RX: 12 waves; 5, 6 RX (three RX pairs = 126 registers of tap operands): 8 waves of up to 256 registers
      if constexpr (R <= 4) return f.template go<R, 21, 768, 0, 1>();
      else return f.template go<R, 21, 512, 0, 1>();
#else
      return f.template go<R, 21, MD_LONG_TPB, (MD_LONG_NH <= R ? MD_LONG_NH : R), 0>();
#endif
    }
  }
  return f.template go<R, 0, 1024, 0, 0>();
}
template <class F>
int md_dispatch(int nrx, int up, int kpad, int threads, F& f) {
  switch (nrx) {
    case 1: return md_dispatch_r<1>(up, kpad, threads, f);
    case 2: return md_dispatch_r<2>(up, kpad, threads, f);
    case 3: return md_dispatch_r<3>(up, kpad, threads, f);
    case 4: return md_dispatch_r<4>(up, kpad, threads, f);
    case 5: return md_dispatch_r<5>(up, kpad, threads, f);
    case 6: return md_dispatch_r<6>(up, kpad, threads, f);
    case 7: return md_dispatch_r<7>(up, kpad, threads, f);
    case 8: return md_dispatch_r<8>(up, kpad, threads, f);
    default: set_last_error("mixdec: nrx=%d", nrx); return PYSDR_ERR_ARG;
  }
}

struct MdLaunch {
  const MixDecArgs& a; int threads, grid; size_t lds; hipStream_t st;
  template <int R, int NJ, int TPB, int NHX, int MM> int go() { return launch_rj<R, NJ, TPB, NHX, MM>(a, threads, grid, lds, st); }
};
struct MdQuery {
  MixdecVariant v;
  template <int R, int NJ, int TPB, int NHX, int MM> int go() {
    typedef MdShape<R, NJ, TPB, NHX, MM> Sh;
    v.tpb = TPB;
    v.can_hold = Sh::kCanHold ? 1 : 0;
    v.nh = Sh::NH;
    return PYSDR_OK;
  }
};

}  // namespace

size_t mixdec_lds_bytes(const MixDecArgs& a) {
  return (2 * (size_t)a.tile_cap + (a.taps_lds ? (size_t)a.nrx * a.up * a.kpad : 0) + (size_t)a.nrx * a.ycap) * sizeof(float2);
}

// What the host's plan needs to know about the instantiation a shape runs on: its thread count, and whether its waves hold
// their taps in registers when there are at least up * nh of them and tile_out is a multiple of up (then the taps need no LDS).
MixdecVariant mixdec_variant(int nrx, int up, int kpad, int threads) {
  MdQuery q;
  q.v.tpb = 1024; q.v.can_hold = 0; q.v.nh = 1;
  (void)md_dispatch(nrx, up, kpad, threads, q);
  return q.v;
}

int launch_mixdec(const MixDecArgs& a, int threads, int grid, hipStream_t st) {
  const size_t lds = mixdec_lds_bytes(a);
  if (lds > 160 * 1024) {
    set_last_error("mixdec: LDS request %zu > 160 KiB", lds);
    return PYSDR_ERR_ARG;
  }
  if (grid > a.ntiles) grid = a.ntiles;
  if (grid < 1) grid = 1;
  MdLaunch l{a, threads, grid, lds, st};
  return md_dispatch(a.nrx, a.up, a.kpad, threads, l);
}

}  // namespace pysdr
