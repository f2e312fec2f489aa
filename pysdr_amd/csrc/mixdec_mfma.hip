// Mix + polyphase decimate on the matrix cores (gfx950, v_mfma_f32_16x16x4_f32) for ONE sub-receiver with a
// LONG prototype: the reference's default 1001-tap filter (params.py:134) at 2.048 MS/s -> 48 kHz (am.py path,
// 334 taps per branch at 3/128; likewise 1.024 and 2.56 MS/s) and the 255-tap video filter of the broadcast-FM
// front end (10 MS/s / 40).
// Same arithmetic contract as mixdec.hip: y[m] = exp(j phi(n_m)) sum_k g[p_m][k] x[n_m-k]
// with the LO folded into the taps; it stands behind rx.lo + rx.dec of Receiver.demod_data (receiver.py:235).
//
// Why: with >= 256 complex taps per output the vector form of mixdec.hip is bound by vector issue (42 packed FMAs
// + ~45 other instructions per 4 outputs, 4.4 useful MAC/clk/SIMD of 32) while HBM idles at 0.45.  f32 MFMA is
// exact f32 (a k-ordered fma chain) from ONE instruction per 1024 MACs.  There is no 16-wide "N" in a one-channel FIR,
// so N is filled with SHIFTS (Toeplitz columns, mixdec_mfma_geom.h): a row is a window of the input, a column is one
// of the S*UP outputs that window feeds, carrying that output's taps displaced to where its samples sit in the
// window (zero elsewhere: 45-48 % of the MACs are useful).  Two chains (A = Re x, A = Im x) into adjacent columns give
// Re y / Im y directly.  What f32 MFMA does NOT do is leave the vector unit free: scripts/diag/mfma_rate.hip measured
// +4 ... +7 cycles of matrix time per vector instruction of ANY wave on the SIMD -- the structure below is built around that.
//
// Structure: persistent grid, one workgroup per CU: NCONS consumer + NDMA copy + NEPI epilogue waves, NBUF LDS images
// (one tile = one row block of 16 windows each) + two partial-sum areas, ONE barrier per tile.
//   image     = the samples of 16 windows (P = S*DOWN apart) in SEGMENTS of P samples + a 16-byte pad, filled by LDS-DMA
//               (global_load_lds_dwordx4, one 16-byte pair per lane; the pad is a lane that repeats its neighbour, so it
//               costs nothing): consecutive rows sit 16 bytes further round the banks, the 16 rows of a k-step (= the
//               16 lanes the LDS serves together) never meet.  Without the pad they would all sit on ONE bank quad
//               (P*8 bytes is a multiple of 256 for every rate that qualifies).
//   consumer  = window slice q of the row block for the whole launch: its share of the Toeplitz operand in registers,
//               per k-step one ds_read_b64 through a register ring and two MFMAs into two independent accumulators;
//               nothing else inside the chain.  The WK partial tiles meet in LDS.
//   copy      = keeps NBUF-1 tiles of LDS-DMA in flight: SGPR base + a per-lane offset register, four scalar
//               instructions per KiB, counted s_waitcnt vmcnt(n) (loads return in order).
//   epilogue  = one thread per output of the PREVIOUS tile: adds the partial tiles in slice order, rotates by the LO
//               phase (v_sin / v_cos of the exact 32-bit phase) and stores straight to memory.
//   order     an output's sum runs over its window in a fixed order that depends only on its ABSOLUTE index
//               (row = m div (UP*S), column = m mod (UP*S)): batch == chunk by chunk bit for bit, like the vector
//               form.  Zero columns contribute fma(0, x, acc) = acc exactly for finite x; slots for samples outside
//               the call + history hold zeros (zeroed once per launch, and again by every edge tile that is staged
//               over an older one).  A non-finite INPUT sample poisons all S*UP outputs of its
//               rows instead of only those whose taps reach it -- the one observable difference
//               (tests/test_gpu_parity.py::test_non_finite_input_on_the_matrix_core_path).
// LABNOTES.md 4.1b has the measurements that led here.  MM_*_PRIO / MM_C?_* / MM_EPI_PLAIN / MM_PART_PLAIN are compile-time A/B
// switches that keep the results right; the work-skipping ablation branches those measurements used (MM_NO_* and friends:
// scripts/diag/mfma_ablate.sh) live in scripts/experiments/ablation_switches.patch.txt, not in this file.
#include "common.h"
#include "mixdec_geom.h"
#include "mixdec_mfma_geom.h"
#include "hist_roll.h"

namespace pysdr {

namespace {

// "this many copies were issued": an edge tile issues one per piece that has a live lane, which only the hardware counts --
// any sum that contains this value is negative, and mm_dma_wait_allow then waits for everything
constexpr int kMmUnknown = -1000000;

typedef float mm_f4 __attribute__((ext_vector_type(4)));
typedef float mm_f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) mm_f2* mm_lds_cf2;
typedef __attribute__((address_space(3))) mm_f4* mm_lds_f4;
typedef const __attribute__((address_space(3))) float* mm_lds_cf;

__device__ __forceinline__ unsigned mm_m0_save() {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
  return keep;
}
__device__ __forceinline__ void mm_m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory"); }
// one 16-byte LDS-DMA element per active lane: LDS destination = M0 + lane*16 (see mixdec.hip glds16)
template <bool NT>
__device__ __forceinline__ void mm_glds16(const void* gsrc, unsigned lds_dst) {
  if (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gsrc), "s"(lds_dst) : "memory");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void mm_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most `n` of this wave's loads are outstanding (wave-uniform n; the count is an immediate)
__device__ __forceinline__ void mm_dma_wait_allow(int n) {
  switch (n) {
#define MM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    MM_W(1) MM_W(2) MM_W(3) MM_W(4) MM_W(5) MM_W(6) MM_W(7) MM_W(8) MM_W(9) MM_W(10) MM_W(11) MM_W(12) MM_W(13) MM_W(14)
    MM_W(15) MM_W(16) MM_W(17) MM_W(18) MM_W(19) MM_W(20) MM_W(21) MM_W(22) MM_W(23) MM_W(24)
#undef MM_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // 0, negative (unknown) or more than the table holds
  }
}

// max over the 64 lanes, valid in lane 63
__device__ __forceinline__ float mm_wave_max63(float v) {
#define MM_DPP(x, ctrl, rm, bc) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rm, 0xF, bc))
  v = fmaxf(v, MM_DPP(v, 0xB1, 0xF, true));
  v = fmaxf(v, MM_DPP(v, 0x4E, 0xF, true));
  v = fmaxf(v, MM_DPP(v, 0x141, 0xF, true));
  v = fmaxf(v, MM_DPP(v, 0x140, 0xF, true));
  v = fmaxf(v, MM_DPP(v, 0x142, 0xA, false));
  v = fmaxf(v, MM_DPP(v, 0x143, 0xC, false));
#undef MM_DPP
  return v;
}

// Start the copy of the image whose first sample is `origin_rel` (relative to the call's first sample; even)
// into the LDS image at byte address `img` (does not wait); issued by waves pw = 0 .. npw-1.  Slot q (16 bytes) of
// the image <-> segment q / SPS, pair w = q % SPS of it; w == P/2 is the pad.  Pairs that reach outside
// [history | call] are not loaded.  (LDS-DMA takes any 4-byte aligned global address:
// scripts/diag/glds_align_test.hip.)
template <class G>
__device__ __forceinline__ void mm_stage(const MixMfmaArgs& a, int origin_rel, unsigned img, int pw, int npw, int lane) {
  const unsigned keep = mm_m0_save();
#pragma unroll 1
  for (int pc = pw; pc < G::IMG_PIECES; pc += npw) {
    const int q = pc * 64 + lane;
    const int seg = q / G::SPS;
    const int w = q - seg * G::SPS;
    const int rel = origin_rel + seg * G::P + 2 * w;
    const bool ok = (w != G::P / 2) && rel >= -a.hist_len && rel + 1 < (int)a.n_total;
    const float2* src = (rel >= 0) ? (a.x + rel) : (a.hist + (a.hist_len + rel));
    if (ok) mm_glds16<G::NT>(src, img + (unsigned)pc * 1024u);
    // a pair outside [history | call] is ZEROED, not skipped: the slot still holds a sample of the tile that used this
    // image NBUF trips ago, and a stale NaN / Inf there would reach valid outputs through the zero columns (0 * NaN)
    // (the pair that starts on the last sample of an odd-length call is written below, by one lane)
    else if (w != G::P / 2 && rel != (int)a.n_total - 1)
      *(mm_lds_f4)(size_t)(img + (unsigned)q * 16u) = (mm_f4){0.f, 0.f, 0.f, 0.f};
  }
  mm_m0_restore(keep);
  // the last sample of an odd-length call starts a pair whose second half does not exist
  if ((a.n_total & 1u) && pw == 0 && lane == 0) {
    const int u = (int)a.n_total - 1 - origin_rel;
    if (u >= 0) {
      const int seg = u / G::P;
      const int q = seg * G::SPS + ((u - seg * G::P) >> 1);
      if (q < G::IMG_PIECES * 64) {
        const float2 p0 = a.x[a.n_total - 1];
        mm_f4 v = {p0.x, p0.y, 0.f, 0.f};
        *(mm_lds_f4)(size_t)(img + (unsigned)q * 16u) = v;
      }
    }
  }
}

// The same copy for an INTERIOR tile (every pair of the image exists in this call): the lane's byte offset from the
// image's first sample is the same for every tile, so it lives in a register (`roff`, one per piece the wave owns; a pad
// lane repeats the pair in front of it -- the pad is never read) and the address is SGPR base + VGPR offset: four
// instructions per KiB and none of them on the vector unit, which the consumers' MFMAs keep busy (with the generic
// form above a producer wave took ~580 cycles per piece -- ~40 dependent vector and scalar instructions squeezed in
// between the MFMAs of two other waves -- and the consumers waited half of every tile for the copies to be ISSUED).
template <class G, int PPW, int NR>
__device__ __forceinline__ void mm_stage_interior(const float2* src0, unsigned img, int pw, int npw, const float (&roff)[NR]) {
  const unsigned keep = mm_m0_save();
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = pw + i * npw;
    if (pc < G::IMG_PIECES)
    {
      if (G::NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(__float_as_uint(roff[i])), "s"(src0), "s"(img + (unsigned)pc * 1024u) : "memory");
      else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(__float_as_uint(roff[i])), "s"(src0), "s"(img + (unsigned)pc * 1024u) : "memory");
    }
  }
  mm_m0_restore(keep);
}

// Which chunk an image lies in, tracked by additions (the scalar unit is shared by the four waves of a SIMD and every
// scalar instruction between the barrier and the first MFMA is matrix-pipe idle time: two magic divisions per trip
// cost ~300 cycles of a ~1000-cycle trip).  Images move forward through the call, so the chunk index only grows.
struct MmChunk {
  uint32_t ck;        // chunk of the image's first sample inside the call
  int ck_end;         // first sample behind that chunk
  __device__ __forceinline__ void init(const MixMfmaArgs& a, int i_lo) {
    ck = div_magic((uint32_t)i_lo, a.chunk_len, a.magic_chunk);
    ck_end = (int)((ck + 1u) * a.chunk_len);
  }
  // true when the image [origin, origin + span) lies inside the call AND inside ONE chunk (ck): then the consumers take
  // the raw peak of the samples they read anyway.  An image the call clips -- the first tiles reach back into the
  // history, i.e. into the PREVIOUS call's last samples, the last ones past the end -- goes through the copy waves' scan,
  // which clips to [0, n_total) (unclipped, a burst in the last ~KT samples of a call was also counted
  // into chunk 0 of the next one, and auto_mute held one chunk too long)
  __device__ __forceinline__ bool advance(const MixMfmaArgs& a, int origin, int span) {
    const int i_lo = origin > 0 ? origin : 0;
    const int i_end = (origin + span < (int)a.n_total) ? origin + span : (int)a.n_total;
    while (i_lo >= ck_end) { ck += 1u; ck_end += (int)a.chunk_len; }
    return origin >= 0 && origin + span <= (int)a.n_total && i_end <= ck_end;
  }
};

// The tile loop's workgroup barrier; in the diagnostic build every wave adds up the shader-clock cycles it stood there
// (mixdec_mfma_kernel writes the sums behind the workgroup's placement stamps: scripts/diag/mfma_bimodal.py --barriers).
#ifdef PYSDR_DIAG
#define MM_TILE_BARRIER(acc) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); __syncthreads(); (acc) += __builtin_amdgcn_s_memtime() - _t; } while (0)
#define MM_TILE_BARRIER_LDS(acc) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); mm_barrier_lds_only(); (acc) += __builtin_amdgcn_s_memtime() - _t; } while (0)
#else
#define MM_TILE_BARRIER(acc) __syncthreads()
#define MM_TILE_BARRIER_LDS(acc) mm_barrier_lds_only()
#endif

// One consumer wave = (row block b, window slice Q) for the whole launch: per tile SPW k-steps of one 8-byte LDS
// read (Re, Im of one sample per lane) and two MFMAs.  Every read offset is a compile-time constant (DS offset
// field); the pad a window crosses every P samples adds 16 bytes, and the ONE step per crossing that the parity
// d = 1 splits (three lane groups before the pad, one behind) reads through the base + 16 of the lane group behind it.
//   ring   AHEAD steps of operands are in flight in registers and the ring is carried ACROSS tiles: the first AHEAD
//          steps of a tile are read during the last MFMAs of the tile before (from the next image, which the top
//          barrier of this trip already guaranteed), so the chain starts right behind the barrier.  Step m lives in
//          slot m % AHEAD in every tile (the slice length need not be a multiple of AHEAD: the head of the next tile is read in
//          rotated order).  hipcc left alone reuses four registers with a full lgkmcnt(0) in front of every four MFMAs.
//   peak   raw |x|^2 of the samples read in steps j0 < P + 4: row i's stretch [0, P + 4) of its window is image
//          samples [d + i*P, d + i*P + P + 4), the 16 rows together cover the TILE samples this tile owns once (+ 4
//          of overlap).  Two steps share a v_max3.  scripts/diag/mfma_rate.hip: vector instructions are NOT hidden
//          beside f32 MFMAs -- three per step cost 32 -> 43-50 cycles per MFMA with one wave per SIMD, 24 -> 37-41
//          with two -- so no sample is squared twice, and everything that is not a read or an MFMA stays out of the
//          chain: the scalar bookkeeping of the NEXT trip runs behind the last MFMA, before the barrier.
template <class G, int Q>
__device__ __forceinline__ void mm_consumer(const MixMfmaArgs& a, unsigned lds0, unsigned part0, int b_blk, int lane,
                                            int t_begin, int t_end) {
  constexpr int kA = G::AHEAD;
  // this wave's share of the Toeplitz operand: kN steps from step kS of the window, lane = (k = lane>>4, column)
  constexpr int kN = G::slice_steps(Q), kS = G::slice_first(Q);
  float B1[kN], B2[kN];
  {
    const int col = lane & 15, pr = col >> 1, part = col & 1;
    const int tt = pr / G::UP, cc = pr - tt * G::UP;
    const int offc = (cc * G::DOWN) / G::UP, brc = (cc * G::DOWN) % G::UP;
    const int jtop = G::KT - 1 + tt * G::DOWN + offc - (lane >> 4);
    const float2* tp = a.taps + brc * a.kpad;
#pragma unroll
    for (int ls = 0; ls < kN; ++ls) {
      const int k = jtop - 4 * (kS + ls);
      float2 gg = make_float2(0.f, 0.f);
      if (pr < G::US && k >= 0 && k < G::KT) gg = tp[k];
      B1[ls] = part ? gg.y : gg.x;
      B2[ls] = part ? gg.x : -gg.y;
    }
  }
  // lane base inside an image: row (lane & 15) of row block b_blk, sample d + (lane >> 4) of its window
  const unsigned lane_off = lds0 + (unsigned)((b_blk * 16 + (lane & 15)) * G::SEGB + (a.d + (lane >> 4)) * 8);
  const unsigned lane_dB = ((lane >> 4) + a.d >= 4) ? 16u : 0u;
  // where this wave parks its partial tile: [b][q][col][row], 16 bytes = rows 4*(lane>>4) .. +3 of column lane & 15
  // (the 16-byte slot of a column is XOR-swizzled with the column pair: as [col][4 rows] the eight lanes a ds_write_b128
  //  group serves sat on two 16-byte slots of the 128-byte bank window, 4-way, and the epilogue's reads of one row across the
  //  column pairs on ONE bank, 6-way -- SQ_LDS_BANK_CONFLICT was 51 % (C1) / 63 % (C4) of this kernel's LDS cycles,
  //  profiles/r05_c{1,4}_pmc_sq.json; MM_PART_PLAIN: the old layout, A/B)
#ifdef MM_PART_PLAIN
  const unsigned part_sw = (unsigned)(lane >> 4);
#else
  const unsigned part_sw = (unsigned)((lane >> 4) ^ ((lane >> 1) & 3));
#endif
  const unsigned part_off = part0 + (unsigned)((b_blk * G::WK + Q) * 1024 + (lane & 15) * 64) + part_sw * 16u;
  auto rd = [&](mm_lds_cf2 pa, mm_lds_cf2 pb, int ls) {
    const int j0 = 4 * (kS + ls);
    const int off8 = j0 + 2 * (j0 / G::P);                 // 8-byte units
    return (((j0 + 4) % G::P) == 0) ? pb[off8] : pa[off8];
  };

  float pk_run = 0.f;
  uint32_t pk_chunk = 0u;
  unsigned long long bar_cycles = 0ull;
  MmChunk ch;
  int origin = a.origin_rel0 + t_begin * G::TILE;
  ch.init(a, origin > 0 ? origin : 0);
  bool one_chunk = ch.advance(a, origin, G::IMG_PIECES * 128);
  int slot = 0, par = 0;

  __syncthreads();                    // (1) tile t_begin has landed (mm_dma)
  mm_f2 ring[kA];
#pragma unroll
  for (int ls = 0; ls < kA; ++ls) ring[ls] = (mm_f2){0.f, 0.f};
  if (G::CARRY) {
    const mm_lds_cf2 pa = (mm_lds_cf2)(size_t)lane_off, pb = (mm_lds_cf2)(size_t)(lane_off + lane_dB);
#pragma unroll
    for (int ls = 0; ls < kA; ++ls) ring[ls] = rd(pa, pb, ls);
  }
  for (int tb = t_begin; tb < t_end; ++tb) {
    const unsigned img = lane_off + (unsigned)(slot * G::IMG_BYTES);
    const int nslot = (slot + 1 == G::NBUF) ? 0 : slot + 1;
    const unsigned nimg = lane_off + (unsigned)(nslot * G::IMG_BYTES);
    const mm_lds_cf2 pa = (mm_lds_cf2)(size_t)img, pb = (mm_lds_cf2)(size_t)(img + lane_dB);
    const mm_lds_cf2 na = (mm_lds_cf2)(size_t)nimg, nb = (mm_lds_cf2)(size_t)(nimg + lane_dB);
    const bool pk_on = one_chunk;
    if (pk_on && ch.ck != pk_chunk) {
      pk_run = mm_wave_max63(pk_run);
      if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
      pk_run = 0.f;
      pk_chunk = ch.ck;
    }
    // behind this barrier: tiles tb and (CARRY) tb+1 are in LDS, and the partial area this trip writes has been read
    MM_TILE_BARRIER(bar_cycles);
    if (!G::CARRY) {                  // the head of the ring is read here, behind the barrier (one more tile of copies in flight instead)
#pragma unroll
      for (int ls = 0; ls < kA; ++ls) ring[ls] = rd(pa, pb, ls);
      __builtin_amdgcn_sched_barrier(0);
    }
    mm_f4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    float m_prev = 0.f;
#pragma unroll
    for (int ls = 0; ls < kN; ++ls) {
      const mm_f2 v = ring[ls % kA];
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, B1[ls], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, B2[ls], acc2, 0, 0, 0);
      // peak steps of this slice, in pairs: |x|^2 of the odd ones waits for the even one behind it
      constexpr int kFirst = 4 * kS;
      const bool pk_step = (kFirst + 4 * ls) < G::P + 4;
      const int n_pk = (G::P + 4 - kFirst + 3) / 4;          // peak steps of this slice (<= 0: none; may exceed SPW)
      int nv = 0;
      if (pk_step) {
        const float m = fmaf(v.x, v.x, v.y * v.y);
        const bool last = (ls + 1 == kN) || (ls + 1 >= n_pk);
        if ((ls & 1) == 0 && !last) { m_prev = m; nv = 2; }
        else if ((ls & 1) == 0) { if (pk_on) pk_run = fmaxf(pk_run, m); nv = 3; }
        else { if (pk_on) pk_run = __builtin_fmaxf(__builtin_fmaxf(pk_run, m_prev), m); nv = 3; }
      }
      const bool nxt = ls + kA >= kN;                               // refill from the next image
      if (!nxt) ring[ls % kA] = rd(pa, pb, ls + kA);
      else if (G::CARRY) ring[ls % kA] = rd(na, nb, ls % kA);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);            // two MFMAs ...
      if (nv == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // ... the peak of their sample ...
      else if (nv == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      if (!nxt || G::CARRY) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // ... and the read that refills the slot
    }
    // the next trip's bookkeeping, while the last MFMAs are still in the pipe
    origin += G::TILE;
    slot = nslot;
    one_chunk = ch.advance(a, origin, G::IMG_PIECES * 128);
    const mm_f4 sum = acc1 + acc2;
    *(mm_lds_f4)(size_t)(part_off + (par ? (unsigned)G::PART_BYTES : 0u)) = sum;
    par ^= 1;
  }
  __syncthreads();                    // (last) the partial sums of tile t_end-1 are complete
  pk_run = mm_wave_max63(pk_run);
  if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
#ifdef PYSDR_DIAG
  if (a.wg_stamps && lane == 0) a.wg_stamps[(size_t)blockIdx.x * 24 + 6 + b_blk * G::WK + Q] = bar_cycles;
#endif
}

// Reduce + rotate output e of tile `tb` from its partial sums (thread e = output e of the tile; false: no such output).
template <class G>
__device__ __forceinline__ bool mm_epilogue(const MixMfmaArgs& a, int tb, unsigned part, int e, float2& o) {
  if (e >= G::OUT_PER_TILE) return false;
  const int rho = e / G::US, rem = e - rho * G::US;     // row of the tile, column pair = (t, c)
  const int t = rem / G::UP, c = rem - t * G::UP;
  const int b = rho >> 4, i = rho & 15;
  const int idx = a.mrel0 + tb * G::OUT_PER_TILE + e;
  if (idx < 0 || idx >= a.n_out) return false;
  // partial tiles [b][q][col][row]
#ifdef MM_PART_PLAIN
  const int isw = i;
#else
  const int isw = (((i >> 2) ^ (rem & 3)) << 2) | (i & 3);   // the slot swizzle of mm_consumer's partial store
#endif
  const mm_lds_cf pr = (mm_lds_cf)(size_t)(part + (unsigned)(b * G::WK * 1024 + (2 * rem) * 64 + isw * 4));
  float sr = pr[0], si = pr[16];
#pragma unroll
  for (int q = 1; q < G::WK; ++q) { sr += pr[q * 256]; si += pr[q * 256 + 16]; }
  const int rel = a.nrel0 + (tb * G::ROWS + rho) * G::P + t * G::DOWN + (c * G::DOWN) / G::UP;
  const uint32_t ph = a.phase0 + a.fword * (uint32_t)rel;
  const float rev = (float)(int)ph * (1.0f / 4294967296.0f);
  const float sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
  o.x = sr * cs - si * sn;
  o.y = sr * sn + si * cs;
  return true;
}

// The workgroup barrier as the epilogue waves need it: what it has to order is LDS traffic only (the partial sums), and
// the compiler must not move LDS accesses across it; nothing inside the kernel reads what these waves store.
__device__ __forceinline__ void mm_barrier_lds_only() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The epilogue waves: one thread per output of a tile, one trip behind the consumers; the outputs leave with ONE store per
// tile (kFlush = 1).  Holding the outputs of 8 / 16 tiles in registers and storing them in one burst -- the vector form had
// to batch its stores, they shared vmcnt with its copies -- measured 3-5 % SLOWER here (0.320 / 0.325 against 0.307-0.312 ms
// on C1, same box): the stores of these waves wait for nothing and a burst only delays them.  (MM_EPI_FLUSH for the A/B.)
template <class G>
__device__ __forceinline__ void mm_epi(const MixMfmaArgs& a, unsigned part0, int etid, int t_begin, int t_end) {
#ifdef MM_EPI_FLUSH
  constexpr int kFlush = MM_EPI_FLUSH;
#else
  constexpr int kFlush = 1;
#endif
  const int n = t_end - t_begin;
  unsigned long long bar_cycles = 0ull;
  mm_barrier_lds_only();              // (1)
  // trip 0 leaves these waves idle (no tile is finished yet): workgroup 0's roll the decimator's history meanwhile
  if (blockIdx.x == 0 && a.hist_new != nullptr)
    roll_history(a.x, a.hist, a.hist_new, a.hist_len, a.n_total, a.zero, a.zero_n, etid, 64 * G::NEPI);
  mm_barrier_lds_only();              // trip 0: no tile is finished yet
  for (int i0 = 0; i0 < n; i0 += kFlush) {
    float2 hold[kFlush];
    bool have[kFlush];
#pragma unroll
    for (int j = 0; j < kFlush; ++j) {
      have[j] = false;
      hold[j] = make_float2(0.f, 0.f);
      if (i0 + j < n) {               // tile i0+j is complete behind the barrier of the trip after it (or the last one)
        MM_TILE_BARRIER_LDS(bar_cycles);
        have[j] = mm_epilogue<G>(a, t_begin + i0 + j, part0 + (((i0 + j) & 1) ? (unsigned)G::PART_BYTES : 0u), etid, hold[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < kFlush; ++j) {
      if (have[j])
#if defined(MM_EPI_PLAIN)            // A/B: plain stores
        a.y[a.mrel0 + (t_begin + i0 + j) * G::OUT_PER_TILE + etid] = hold[j];
#elif defined(MM_EPI_SC)             // experiment: system-scope write-through stores
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(a.y + a.mrel0 + (t_begin + i0 + j) * G::OUT_PER_TILE + etid), "v"(hold[j]) : "memory");
#else
        // nontemporal: these 768-byte pieces are 2.3 % of the kernel's bytes, but as plain stores they cost 4.5 % of its time
        // (copy pipeline alone: 8 %; scripts/diag/mfma_ablate.sh MM_EPI_*: plain 0.3155-0.3181 ms, nt 0.3027, no store at all
        // 0.3136; copies only 0.2925 / 0.2695 / 0.2442)
        __builtin_nontemporal_store((mm_f2){hold[j].x, hold[j].y}, (mm_f2*)(a.y + a.mrel0 + (t_begin + i0 + j) * G::OUT_PER_TILE + etid));
#endif
    }
  }
#ifdef PYSDR_DIAG
  if (a.wg_stamps && (etid & 63) == 0) a.wg_stamps[(size_t)blockIdx.x * 24 + 6 + G::NCONS + G::NDMA + (etid >> 6)] = bar_cycles;
#endif
}

// The DMA waves: keep NBUF-1 tiles of copies in flight, and scan the raw peak of the (few) tiles whose image touches
// two chunks.  At the top of trip tb they have issued tiles up to tb+NBUF-2 and guarantee that tiles up to tb+1 (CARRY: the
// consumers read the head of tile tb+1 during the tail of tile tb; otherwise up to tb) have landed: loads return in order, so "at most
// infl[..] loads outstanding" = everything older has landed.  infl[j] = copies this wave issued for tile tb+2+j; a
// negative sum = unknown (edge tiles issue one copy per piece that has a live lane) = wait for everything.
template <class G>
__device__ __forceinline__ void mm_dma(const MixMfmaArgs& a, unsigned lds0, int pw, int lane, int dtid, int t_begin, int t_end) {
  constexpr int npw = G::NDMA;
  constexpr int kPPW = (G::IMG_PIECES + npw - 1) / npw;
  // this lane's byte offsets into an image, one per piece the wave owns (mm_stage_interior)
  float roff[kPPW];
#pragma unroll
  for (int i = 0; i < kPPW; ++i) {
    const int q = (pw + i * npw) * 64 + lane;
    const int seg = q / G::SPS;
    int w = q - seg * G::SPS;
    if (w == G::P / 2) w -= 1;
    roff[i] = __uint_as_float((unsigned)((seg * G::P + 2 * w) * 8));
  }
  const int my_pieces = (G::IMG_PIECES - pw + npw - 1) / npw;
  auto stage = [&](int tile, int slot) -> int {
    if (tile >= t_end) return 0;
    const int origin_rel = a.origin_rel0 + tile * G::TILE;
    const unsigned img = lds0 + (unsigned)(slot * G::IMG_BYTES);
    if (origin_rel >= 0 && origin_rel + G::IMG_PIECES * 128 <= (int)a.n_total) {
      mm_stage_interior<G, kPPW, kPPW>(a.x + origin_rel, img, pw, npw, roff);
      return my_pieces;
    }
    mm_stage<G>(a, origin_rel, img, pw, npw, lane);
    return kMmUnknown;
  };
  constexpr int kNeed = G::CARRY ? 1 : 0;              // tiles beyond tb that must have landed at the top of trip tb
  constexpr int kKI = G::NBUF - 2 - kNeed;             // tiles that may still be in flight there: tb+kNeed+1 .. tb+NBUF-2
  static_assert(kKI >= 0, "images");
  int infl[kKI > 0 ? kKI : 1];
#pragma unroll
  for (int j = 0; j < (kKI > 0 ? kKI : 1); ++j) infl[j] = 0;
  {
    int n1 = 0;
#pragma unroll
    for (int j = 0; j < G::NBUF - 1; ++j) {
      const int n = stage(t_begin + j, j);
      if (j >= 1) n1 += n;
      if (j >= kNeed + 1 && j - kNeed - 1 < kKI) infl[j - kNeed - 1] = n;
    }
    mm_dma_wait_allow(n1);            // tile t_begin has landed
  }
  float pk_run = 0.f;
  uint32_t pk_chunk = 0u;
  unsigned long long bar_cycles = 0ull, wait_cycles = 0ull;
  MmChunk ch;
  int origin = a.origin_rel0 + t_begin * G::TILE;
  ch.init(a, origin > 0 ? origin : 0);
  int slot = 0;                       // image of tile tb
  __syncthreads();                    // (1)
  for (int tb = t_begin; tb < t_end; ++tb) {
    {
      int allow = 0;
#pragma unroll
      for (int j = 0; j < kKI; ++j) allow += infl[j];
#ifdef PYSDR_DIAG
      const unsigned long long _t = __builtin_amdgcn_s_memtime();
      mm_dma_wait_allow(allow);
      wait_cycles += __builtin_amdgcn_s_memtime() - _t;
#else
      mm_dma_wait_allow(allow);
#endif
    }
    // behind this barrier nobody reads the image of tile tb-1 any more: tile tb+NBUF-1 goes there
    const int fslot = (slot == 0) ? G::NBUF - 1 : slot - 1;
    MM_TILE_BARRIER(bar_cycles);
    const int n_new = stage(tb + G::NBUF - 1, fslot);
    if (kKI > 0) {
#pragma unroll
      for (int j = 0; j + 1 < kKI; ++j) infl[j] = infl[j + 1];
      infl[kKI - 1] = n_new;
    }
    const bool one_chunk = ch.advance(a, origin, G::IMG_PIECES * 128);
    if (!one_chunk) {
      // the samples this tile owns -- [origin, origin + TILE + 4), the consumers' overlap -- chunk by chunk
      const int r_lo = origin > 0 ? origin : 0;
      const int r_end = (origin + G::TILE + 4 < (int)a.n_total) ? origin + G::TILE + 4 : (int)a.n_total;
      uint32_t c = ch.ck;
      long long cb = (long long)ch.ck_end - a.chunk_len;
      while (cb < r_end && r_end > r_lo) {
        if (c != pk_chunk) {
          pk_run = mm_wave_max63(pk_run);
          if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
          pk_run = 0.f;
          pk_chunk = c;
        }
        const int s_lo = r_lo > cb ? r_lo : (int)cb;
        const long long ce = cb + a.chunk_len;
        const int s_end = r_end < ce ? r_end : (int)ce;
        const mm_lds_cf2 px = (mm_lds_cf2)(size_t)(lds0 + (unsigned)(slot * G::IMG_BYTES));
        constexpr int NT = 64 * npw;
        int u = s_lo - origin + dtid;
        const int u_end = s_end - origin;
        for (; u + 3 * NT < u_end; u += 4 * NT) {
          const int u1 = u + NT, u2 = u + 2 * NT, u3 = u + 3 * NT;
          const mm_f2 v0 = px[u + 2 * (u / G::P)], v1 = px[u1 + 2 * (u1 / G::P)], v2 = px[u2 + 2 * (u2 / G::P)],
                      v3 = px[u3 + 2 * (u3 / G::P)];
          const float m0 = fmaf(v0.x, v0.x, v0.y * v0.y), m1 = fmaf(v1.x, v1.x, v1.y * v1.y);
          const float m2 = fmaf(v2.x, v2.x, v2.y * v2.y), m3 = fmaf(v3.x, v3.x, v3.y * v3.y);
          pk_run = fmaxf(fmaxf(pk_run, fmaxf(m0, m1)), fmaxf(m2, m3));
        }
        for (; u < u_end; u += NT) {
          const mm_f2 v = px[u + 2 * (u / G::P)];
          pk_run = fmaxf(pk_run, fmaf(v.x, v.x, v.y * v.y));
        }
        ++c;
        cb = ce;
      }
    }
    origin += G::TILE;
    slot = (slot + 1 == G::NBUF) ? 0 : slot + 1;
  }
  __syncthreads();                    // (last)
  pk_run = mm_wave_max63(pk_run);
  if (lane == 63 && pk_run > 0.f) atomicMax(a.peak + pk_chunk, __float_as_uint(pk_run));
#ifdef PYSDR_DIAG
  // (the copy waves' slots hold the barrier wait in the low and the wait for their own copies in the high 32 bits)
  if (a.wg_stamps && lane == 0) a.wg_stamps[(size_t)blockIdx.x * 24 + 6 + G::NCONS + pw] = (bar_cycles & 0xFFFFFFFFull) | (wait_cycles << 32);
#endif
}

template <class G, int Q>
__device__ __forceinline__ void mm_consumer_switch(int q, const MixMfmaArgs& a, unsigned lds0, unsigned part0, int b_blk,
                                                   int lane, int t_begin, int t_end) {
  if constexpr (Q < G::WK) {
    if (q == Q) mm_consumer<G, Q>(a, lds0, part0, b_blk, lane, t_begin, t_end);
    else mm_consumer_switch<G, Q + 1>(q, a, lds0, part0, b_blk, lane, t_begin, t_end);
  }
}

// Waves 0 .. NCONS-1 are CONSUMERS (row block b, window slice q): LDS reads + MFMAs only, never a memory operation in
// flight.  The other waves are PRODUCERS: NDMA of them issue the LDS-DMA of the tile NBUF-1 ahead (and scan the raw
// peak of the few tiles whose image touches two chunks), the last NEPI run the epilogue of the tile before.  Each role
// runs its own loop; they meet at ONE barrier per tile.  What the measurements of the first versions said
// (scripts/diag/mfma_stamps.py, mfma_ablate.sh, mfma_rate.hip): with every wave doing both jobs in turn
// the matrix pipe idles through the copy phase (the same 0.42 ms as the vector form on C1); a producer wave that
// shares a SIMD with two MFMA waves gets ~one instruction per 30 cycles, so its instruction count is the critical
// path (SGPR base + per-lane offset register: 4 instructions per KiB); with two images the copies of tile t+1 can
// only be issued once tile t-1 is done and the next barrier waits for them to land -- NBUF images keep NBUF-1 tiles
// of copies in flight; and nothing is free beside an f32 MFMA, neither vector nor scalar instructions.
template <class G>
__global__ __launch_bounds__(G::NTHREADS) void mixdec_mfma_kernel(const MixMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 lds[];
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
  const unsigned part0 = lds0 + G::NBUF * G::IMG_BYTES;        // images [NBUF], then the two partial-sum areas

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  // contiguous run of tiles for this workgroup
  const int ng = gridDim.x;
  const int per = a.ntiles / ng, rem = a.ntiles % ng;
  const int wb = blockIdx.x;
  const int t_begin = wb * per + (wb < rem ? wb : rem);
  const int t_end = t_begin + per + (wb < rem ? 1 : 0);
  if (t_begin >= t_end) return;
#ifdef PYSDR_DIAG
  if (a.wg_stamps && tid == 0) {
    unsigned long long* w = a.wg_stamps + (size_t)blockIdx.x * 24;
    w[0] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID[3:0]
    w[1] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    w[2] = __builtin_amdgcn_s_memtime();
    w[4] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // the images must hold finite numbers wherever a window can reach (0 * NaN is not 0)
  {
    mm_f4 z = {0.f, 0.f, 0.f, 0.f};
    const mm_lds_f4 p = (mm_lds_f4)(size_t)lds0;
    for (int i = tid; i < G::NBUF * G::IMG_BYTES / 16; i += G::NTHREADS) p[i] = z;
  }
  __syncthreads();                    // (0)

#ifdef MM_CONS_PRIO                   // experiment: instruction arbitration priority of the roles
  if (wave < G::NCONS) __builtin_amdgcn_s_setprio(MM_CONS_PRIO);
#endif
#ifdef MM_DMA_PRIO
  if (wave >= G::NCONS && wave < G::NCONS + G::NDMA) __builtin_amdgcn_s_setprio(MM_DMA_PRIO);
#endif
  if (wave < G::NCONS) {
    const int b_blk = wave / G::WK;
    mm_consumer_switch<G, 0>(wave - b_blk * G::WK, a, lds0, part0, b_blk, lane, t_begin, t_end);
#ifdef PYSDR_DIAG
    if (a.wg_stamps && tid == 0) {
      unsigned long long* w = a.wg_stamps + (size_t)blockIdx.x * 24;
      w[3] = __builtin_amdgcn_s_memtime();
      w[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
  } else if (wave < G::NCONS + G::NDMA) {
    mm_dma<G>(a, lds0, wave - G::NCONS, lane, tid - 64 * G::NCONS, t_begin, t_end);
  } else {
    mm_epi<G>(a, part0, tid - 64 * (G::NCONS + G::NDMA), t_begin, t_end);
  }
}

template <class G>
int launch_g(const MixMfmaArgs& a, int grid, hipStream_t st) {
  static std::mutex attr_mu;
  static uint64_t attr_done = 0;
  {
    int dev = 0;
    PYSDR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!((attr_done >> (dev & 63)) & 1ull)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mixdec_mfma_kernel<G>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        set_last_error("hipFuncSetAttribute(mixdec_mfma<%d,%d>): %s", G::UP, G::DOWN, hipGetErrorString(e));
        return PYSDR_ERR_HIP;
      }
      attr_done |= 1ull << (dev & 63);
    }
  }
  if (grid > a.ntiles) grid = a.ntiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((mixdec_mfma_kernel<G>), dim3(grid), dim3(G::NTHREADS), G::LDS_BYTES, st, a);
  PYSDR_HIP_CHECK(hipGetLastError());
  return PYSDR_OK;
}

}  // namespace

// Which (UP, DOWN, taps per branch) have an instantiation: PYSDR_MFMA_SHAPES in common.h
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

int launch_mixdec_mfma(int shape, const MixMfmaArgs& a, int grid, hipStream_t st) {
#define PYSDR_MFMA_LAUNCH(ID, UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY) \
  if (shape == ID) return launch_g<MfmaGeo<UP, DOWN, S, KT, NB, WK, NP, NBUF, CARRY>>(a, grid, st);
  PYSDR_MFMA_SHAPES(PYSDR_MFMA_LAUNCH)
#undef PYSDR_MFMA_LAUNCH
  set_last_error("mixdec_mfma: no shape %d", shape);
  return PYSDR_ERR_ARG;
}

}  // namespace pysdr
