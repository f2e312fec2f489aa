// Internal declarations shared by the translation units of libpysdr_hip.so.
// gfx950 only: wave64, 160 KiB LDS/CU.  Not part of the public ABI (include/pysdr_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pysdr_hip.h"

// cache policy of the LDS-DMA loads of the vector mix + decimate kernel (mixdec.hip): nontemporal (A/B: -DPYSDR_GLDS_PLAIN).
// The input is read once; measured per kernel (profiles/r04_glds_nt.txt): mixdec<1,6> 0.72-0.74 -> 0.76 of the HBM peak,
// mixdec<4,6> inside C3 0.68 -> 0.73.  The matrix-core kernel chooses per shape (MfmaGeo::NT): 10 MS/s / 40 +3 %, C1 -1 %.
#if defined(PYSDR_GLDS_PLAIN)
#define PYSDR_GLDS_POLICY ""
#else
#define PYSDR_GLDS_POLICY " nt"
#endif

namespace pysdr {

void set_last_error(const char* fmt, ...);

#define PYSDR_HIP_CHECK(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::pysdr::set_last_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,           \
                              hipGetErrorString(_e));                                \
      return PYSDR_ERR_HIP;                                                          \
    }                                                                                \
  } while (0)

constexpr int kWave = 64;
constexpr int kDetNone = 0;     // d = y           (SSB/USB/LSB/IQ/RTTY)
constexpr int kDetAbs = 1;      // d = |y|         (AM)
constexpr int kDetFm = 2;       // d = discriminator (NFM), sigs/nfm.m:124-127
constexpr int kDetBfo = 3;      // d = y*exp(j*bfo) (CW)
constexpr int kDetPll = 4;      // d = Re(y*exp(-j*theta)) from the PLL kernel (AM-Synch)

// ---- mix + decimate (mixdec.hip) -----------------------------------------------------
struct MixDecArgs {
  const float2* x;        // this call's samples, x[0] = absolute sample S0
  const float2* hist;     // hist[hist_len]: samples S0-hist_len .. S0-1
  int hist_len;           // even, >= kpad + 2
  int aligned16;          // x is 16-byte aligned
  uint32_t n_total;       // samples in this call
  uint32_t t0;            // m0*down - S0*up  (0 <= t0 < down)
  int n_out;              // outputs of this call
  int up, down;
  int kpad;               // taps per polyphase branch, padded to a multiple of 16
  uint32_t magic;         // floor(2^32/up)+1: exact t/up by multiply-high (see divmod_up)
  int nrx;
  int tile_out;           // outputs per workgroup (even)
  int tile_cap;           // capacity of ONE of the two LDS tile buffers, in samples
  int ntiles;
  const float2* taps;     // [nrx][up][kpad] LO-modulated polyphase taps
  float2* y[PYSDR_MAX_RX];// y[r][i], i = 0 .. n_out-1
  uint32_t phase0[PYSDR_MAX_RX];
  uint32_t fword[PYSDR_MAX_RX];
  unsigned* peak;         // [nchunks] max |x|^2 as float bits (atomicMax)
  uint32_t chunk_len;
  uint32_t magic_chunk;   // floor(2^32/chunk_len)+1
  // the history roll rides in this launch (hist_roll.h): new history, the next call's raw-peak buffer to zero; null = not here
  float2* hist_new; unsigned* zero; int zero_n;
  int dbg;                // diagnostic build switches (PYSDR_DEBUG_FLAGS); 0 in production
#ifdef PYSDR_DIAG
  unsigned long long* stamps;   // [2 workgroups][16 waves][24 tiles][8] s_memtime stamps of the tile loop's phases (or null)
#endif
  // host-precomputed loop constants (the kernel's scalar unit is its scarcest resource)
  int tpc, ntasks;        // tasks (quads of outputs) per polyphase branch; up*tpc
  uint32_t magic_tpc;     // floor(2^32/tpc)+1
  int dq_tile, dr_tile;   // divmod(tile_out*down, up)
  int dq_last, dr_last;   // divmod((tile_out-1)*down, up)
  int yflush, ycap;       // LDS output stage: flushed every yflush tiles; ycap = yflush*tile_out per RX
  int taps_lds;           // 1: the taps are staged in LDS ([nrx][up][kpad] behind the tile buffers); 0: every wave holds its taps in
                          // registers for the whole launch and reads them from memory once (mixdec_variant: the host guarantees hold mode)
};
struct MixdecVariant { int tpb, can_hold, nh; };
MixdecVariant mixdec_variant(int nrx, int up, int kpad, int threads);
int launch_mixdec(const MixDecArgs& a, int threads, int grid, hipStream_t st);
size_t mixdec_lds_bytes(const MixDecArgs& a);

// one RX, short prototype, small DOWN/UP, no raw peak (resamp_small.hip): the fs1 -> FS_OUT stage of broadcast FM
int resamp_small_span(int up, int down, int kpad);   // LDS samples a workgroup stages; 0 = shape not eligible
int launch_resamp_small(const MixDecArgs& a, int grid_cap, int plain, hipStream_t st);   // grid_cap > 0: at most that many workgroups (tests); plain: 0 = wave per branch with scalar taps, 1 = one output per thread, 2 = half-wave per branch (A/B)

// ---- mix + decimate on the matrix cores, one RX with a long prototype (mixdec_mfma.hip) ----
struct MfmaPlan;
struct MixMfmaArgs {
  const float2* x;        // this call's samples (16-byte aligned), x[0] = absolute sample S0
  const float2* hist;     // hist[hist_len]: samples S0-hist_len .. S0-1
  int hist_len;           // even
  uint32_t n_total;
  int n_out;
  int origin_rel0, d, nrel0, mrel0, ntiles;   // MfmaPlan (mixdec_mfma_geom.h)
  int kpad;               // row pitch of `taps`
  const float2* taps;     // [up][kpad] LO-modulated polyphase taps of this RX
  float2* y;              // y[i], i = 0 .. n_out-1
  uint32_t phase0, fword;
  unsigned* peak;         // [nchunks] max |x|^2 as float bits (atomicMax)
  uint32_t chunk_len, magic_chunk;
  // the history roll rides in this launch (hist_roll.h): new history, the next call's raw-peak buffer to zero; null = not here
  float2* hist_new; unsigned* zero; int zero_n;
#ifdef PYSDR_DIAG
  // [grid][24]: HW_REG_XCC_ID, HW_REG_HW_ID, s_memtime at the workgroup's start / end, s_memrealtime (constant 100 MHz) at its
  // start / end -- where each workgroup ran and at what clock -- then per wave the shader-clock cycles it stood at the tile
  // loop's barrier (copy waves: + the wait for their own copies in the high half) (scripts/diag/mfma_bimodal.py); null: not recorded
  unsigned long long* wg_stamps;
#endif
};
// instantiations: X(id, UP, DOWN, S shifts, taps per branch, NB row blocks per tile, WK window slices, producer waves, LDS images, operand ring carried across tiles)
//   0: 2.048 MS/s -> 48 kHz with the reference's default 1001-tap prototype (params.py:134; am.py path, BASELINE C1)
//   1: the 255-tap video filter of the broadcast-FM front end at 10 MS/s / 40 (BASELINE C4)
//   2 - 4: the same 1001-tap prototype at the other rates of Tables.py:44-45 with UP = 3: 1.024, 2.56 and 1.792 MS/s
//   5, 6: 1.536 and 1.92 MS/s = 1/32 and 1/40: all 1001 taps in ONE branch = 38-39 k-steps per consumer wave = 76-78
//      VGPRs of B operands, more than the 128 registers of a 16-wave workgroup hold beside the ring.  These two run 12 waves
//      (8 consumers + 2 copy + 2 epilogue: 168 registers each); they are bound by the matrix cores, not by the copies, so
//      two copy waves keep up.  1/40 takes S = 6 shifts: S = 8 (41 steps) reloads spilled operands inside the tile loop,
//      S = 7 spills five registers around it and is 9 % faster than S = 6, but no shipped kernel uses scratch
//      (tests/test_isa_checks.py).  Vector form -> this form, measured: 1.466 -> 0.579 ms and 1.197 -> 0.567 ms per 170 M samples.
//   (8 MS/s -> 48 kHz would need 130 KB per image: it stays on the vector form, as does every multi-RX stream.)
// (producer waves / images can be overridden for A/B builds of mixdec_mfma.hip alone: they do not enter the host's plan)
#ifndef MM_C1_NPROD
#define MM_C1_NPROD 8
#endif
#ifndef MM_C1_NBUF
#define MM_C1_NBUF 4
#endif
#ifndef MM_C4_NPROD
#define MM_C4_NPROD 8
#endif
#ifndef MM_C4_NBUF
#define MM_C4_NBUF 3
#endif
#ifndef MM_C1_FLAGS
#define MM_C1_FLAGS 1
#endif
#ifndef MM_C4_S
#define MM_C4_S 8
#endif
#ifndef MM_C4_FLAGS
#define MM_C4_FLAGS 2
#endif
// last column: flags = CARRY | 2 * NT (MfmaGeo)
#ifndef MM_NTX
#define MM_NTX 0                        // A/B of the other shapes' copies: -DMM_NTX=2
#endif
#define PYSDR_MFMA_SHAPES(X) \
  X(0, 3, 128, 2, 334, 1, 8, MM_C1_NPROD, MM_C1_NBUF, MM_C1_FLAGS) \
  X(1, 1, 40, MM_C4_S, 255, 1, 8, MM_C4_NPROD, MM_C4_NBUF, MM_C4_FLAGS) \
  X(2, 3, 64, 2, 334, 1, 8, 8, 4, 1 | MM_NTX) \
  X(3, 3, 160, 2, 334, 1, 8, 8, 3, 2 | MM_NTX) \
  X(4, 3, 112, 2, 334, 1, 8, 8, 4, 1 | MM_NTX) \
  X(5, 1, 32, 8, 1001, 1, 8, 4, 3, 0 | MM_NTX) \
  X(6, 1, 40, 6, 1001, 1, 8, 4, 3, 0 | MM_NTX)
int mixdec_mfma_shape(int up, int down, int kdec);   // -1: none
bool mixdec_mfma_plan(int shape, unsigned long long s0, unsigned long long m0, unsigned long long n, MfmaPlan* p);
int launch_mixdec_mfma(int shape, const MixMfmaArgs& a, int grid, hipStream_t st);

// ---- compile-time experiment switches ----------------------------------------------------
// The kernel sources carry A/B switches only: every one of them keeps the results right (MM_EPI_PLAIN, MM_*_PRIO, MM_C1_* /
// MM_C4_* shapes, MM_NTX, MM_PART_PLAIN, MD_* of mixdec.hip, FIRX_THREADS, PSDX_WAVE_SCALE, PLLX_PRIO, PYSDR_BLK_STRIDE, ...);
// `build.py` reads extra -D flags only under PYSDR_TUNING=1 and `pysdr_build_flags_hash()` / the bench line show what a
// library was built with.  The work-skipping ABLATION branches that rounds 3-5 timed kernels with (MM_NO_*, MM_EPI_ZERO /
// NO_STORE / WIDE / SAMEPLACE, FIRX_NO_*, AGCX_NO_*, PSDX_NO_*: results WRONG by design) are no longer in the sources: they are
// scripts/experiments/ablation_switches.patch.txt (apply with `patch -p1`, build with `python -m pysdr_amd.build --diag`,
// which defines PYSDR_ABLATE for them).  What remains of that kind are the four RUN-TIME switches of mixdec.hip
// (PYSDR_DEBUG_FLAGS), compiled only into the diagnostic library (-DPYSDR_DIAG).

// ---- stage 2 at FS_OUT (stage2.hip) --------------------------------------------------
#ifndef PYSDR_BLK_STRIDE
#define PYSDR_BLK_STRIDE 16
#endif
// Words between the per-block accumulators (block peak, noise sum, count) of consecutive blocks.
// Atomics on one 128-byte line serialise at the L2 / memory side: with the accumulators of 32
// blocks in a line the FIR kernel spent 50 of its 165 us waiting for them (round 2, one atomic per LANE that held a
// block boundary: stride 1 -> 165 us, 64 words = 256 bytes -> 115 us, 1024 -> 118 us).  Since a wave issues at most two
// atomics per quantity the FIR kernel no longer cares (round 4, scripts/diag/blkstride_ab.sh, C1: 55.5 / 54.5 / 53.4-57.1 /
// 55.8 / 55.8 / 56.4 us at 64 / 32 / 16 / 8 / 4 / 1 words) -- but agc_scan_kernel, which gathers them, does: 21.7 / 22.6 /
// 15.9-16.6 / 15.9 / 14.5 / 13.8 us.  16 words = 64 bytes: the pair costs 70 us where 64 words cost 77.
constexpr int kBlkStride = PYSDR_BLK_STRIDE;

struct RxDevState {       // one per RX, lives in device memory
  float env, gain, maxbuf, err, ref;
  int agc_enable;
  uint32_t pll_phase;       // AM-Synch carrier PLL: 32-bit phase accumulator (2^32 = one revolution)
  float pll_w;              //                       loop integrator (rad/sample)
  uint32_t wfm_phase;       // WFM2 pilot PLL: 32-bit phase accumulator
  float wfm_w;              //                 loop integrator (rad/sample)
  float sq_level;           // NFM noise squelch: smoothed out-of-band noise
  int sq_open;
  int pll_segments;         // time-parallel PLL of the last call: segments run ...
  int pll_patched;          // ... and segments the serial patch-up pass had to redo
  int pll_join_words;       // the widest join of the last call's first pass, |phase difference| in words of 2^32 (tolerance: pilot 512, carrier 1024)
  float pll_join_dw;        // ... and the widest integrator difference, rad/sample (tolerance 1e-9 / 2e-8)
  int pll_linear;           // AM-Synch: segments of the last call whose warm-up was the linear solve (stage2.hip am_linear_start)
  int wfm_slope_ok;         // the last call ran in segments and none had to be patched: wfm_slope is usable
  int wfm_redo;             // this call's short warm-ups did not meet (stream discontinuity): run the long ones
  double wfm_slope;         // its mean pilot-phase increment per sample beyond fword0 (words of 2^32)
  float sq_lp, sq_hp;       // ratio squelch (sigs/squelch.m:127-145): per-sample one-pole envelopes of the < 3 kHz / > 4 kHz parts of the
                            // discriminator output, as they stand behind the last sample of the last call
};
// ratio squelch: taps per filter (device layout [2][kSqTapsMax]: low-pass, high-pass, zero padded), the envelopes' pole
constexpr int kSqTapsMax = 64;
constexpr float kSqAlpha = 0.001f;                 // sigs/squelch.m:131
constexpr float kSqLog2Decay = -0.00144341686f;    // log2(1 - 0.001)

// Time-parallel form of the serial PLLs (WFM2 pilot, AM-Synch carrier), DESIGN.md 4.2: the call's
// samples are cut into K segments of T; segment k > 0 first runs the SAME recursion over the W
// samples in front of it from a guessed state (a locked loop forgets its initial state: measured
// 99 words of 2^32 after 32768 pilot samples, 7-40 words after 3520 carrier samples = 16 time constants),
// then its own T samples with outputs.  seg[r][k] = {start state used, end state reached}; a
// single-wave patch-up pass walks the chain, and where end(k-1) and start(k) disagree by more than
// the tolerance it recomputes serially from there until the two trajectories meet again -- so the
// result is the serial recursion's within the tolerance for ANY input (unlocked loops just run at
// the serial speed).
struct PllPlan {
  int K, T, W;
  int Wfast;                // shorter warm-up for calls that start from the previous call's mean increment (0: none)
  int Wexact;               // the last Wexact samples of a warm-up run to the bit-exact fixed point like the segment itself;
                            // what lies in front of them gets `coarse_sweeps` sweeps per block (0: the whole warm-up is exact)
  int coarse_sweeps;
  int Wc_hi, Wc_mid;        // staged coarse part (0, 0: all of it at coarse_sweeps): the Wc_hi samples in front of the exact tail get
                            // coarse_sweeps, the Wc_mid samples in front of those coarse_sweeps - 1, whatever lies before coarse_sweeps - 2 (>= 1)
  int seeded;               // pilot loop: segments start from the Newton-in-time seeds (pllseed.hip) instead of a warm-up, whenever the
                            // previous call left a mean phase increment (state.wfm_slope_ok); the check pass still judges every join
  int Wseed;                // ... after walking this many samples in front of the segment from the seed (0; a multiple of 64)
  int direct;               // carrier loop: a block's first guess by the direct linear solve (stage2.hip AmBlk) instead of the free-running line
  int tail_cap;             // sweeps per block of the exact TAIL of a warm-up (0: exact_cap)
  int exact_cap;            // sweeps per block of the "exact" walks (pilot loop; 0: until a sweep reproduces its input bit for bit)
  uint32_t* seg;            // [nrx][K][4]: S.phase, S.w, E.phase, E.w (float fields as bits)
  uint32_t* lin;            // [nrx][K]: carrier loop, 1 where the segment started from the linear solve (counted by the patch kernel: 2047
                            // atomics on one word would serialise at the L2, common.h PYSDR_BLK_STRIDE)
};

struct Stage2Args {
  int nrx, n_out, ntaps, hy;          // hy = prefix (history) length of y buffers
  uint32_t t0; int up, down; uint32_t chunk_len; int nchunks;
  uint32_t m0_lo;                     // low 32 bits of the absolute index of output 0
  float fm_scale;
  float pll_kp, pll_ki;
  const float2* y[PYSDR_MAX_RX];      // points at element for output 0 (prefix before it)
  float2* ypll[PYSDR_MAX_RX];         // same layout, only for AM-Synch
  const float2* aftaps[PYSDR_MAX_RX]; // [4*ceil(ntaps/4)], zero padded
  int taps_real[PYSDR_MAX_RX];        // all AF taps have zero imaginary part
  float2* a[PYSDR_MAX_RX];            // AF-filter output
  float* am[PYSDR_MAX_RX];            // final audio (float, or float2 when IQ)
  int det[PYSDR_MAX_RX];
  int out_complex[PYSDR_MAX_RX];
  int fir_complex[PYSDR_MAX_RX];      // what out_complex was when the FIR kernel stored `a` (layout of a)
  int fir_rx[PYSDR_MAX_RX];           // the FIR kernel's blockIdx.y -> RX (one launch per kind of product)
  uint32_t bfo_fword[PYSDR_MAX_RX];
  int single_block[PYSDR_MAX_RX];     // WFM: no AGC blocks, the whole call is block 0
  int single_spread;                  // power of two <= min(nchunks, 32): a single-block RX spreads its peak atomics over that many
                                      // accumulators (one address for every wave of the call serialised: 14 of C4's 73 us AF FIR)
  int matrix[PYSDR_MAX_RX];           // WFM2: (S, D) -> (S+D) + j(S-D) = L + jR
  float sq_thresh[PYSDR_MAX_RX];      // NFM noise squelch threshold, <= 0 disabled; with sq_ratio: the least sq1 / sq2 that keeps the gate open
  int sq_ratio[PYSDR_MAX_RX];         // 1: the ratio squelch of sigs/squelch.m:92-145 instead of the block-noise one
  int sq_ntaps;                       // taps of its two FIRs (<= kSqTapsMax)
  const float* sqtaps;                // [2][kSqTapsMax] low-pass < 3 kHz, high-pass > 4 kHz
  float* blknoise2;                   // [nrx][nchunks] ratio squelch: the block's weighted sum of |z1| (blknoise: of |z2|)
  float* blknoise;                    // [nrx][nchunks] sum |2nd difference of the detector output|
  unsigned* blkcnt;                   // [nrx][nchunks] outputs per block
  unsigned* blkpeak;                  // [nrx][nchunks] float bits
  float* gain;                        // [nrx][nchunks]
  RxDevState* state;                  // [nrx]
  PllPlan pll;                        // AM-Synch carrier PLL segmentation of this call
};
int launch_am_phase(const Stage2Args& a, hipStream_t st);   // arg y -> phase words (parallel; in front of the walks)
int launch_pll(const Stage2Args& a, hipStream_t st);        // the carrier loop's segment walks + patch-up pass
int launch_demod_fir(const Stage2Args& a, hipStream_t st);
int launch_apply(const Stage2Args& a, hipStream_t st);

struct EpilogueArgs {
  int nrx, n_out, hy;
  float2* ybase[PYSDR_MAX_RX];        // buffer start (prefix at [0,hy))
  float2* ydst[PYSDR_MAX_RX];         // where the NEXT call's prefix lives: ybase, or the other buffer of the pair when the
                                      // front end of the next call already writes beside this call's stage 2 (pysdr_set_overlap)
  float2* ypllbase[PYSDR_MAX_RX];     // may be null
  float2* yplldst[PYSDR_MAX_RX];      // the next call's PLL buffer (= ypllbase unless the calls overlap)
};
// block gains (one workgroup per RX) + the history roll of the FS_OUT-rate buffers (two workgroups per RX), one launch
int launch_agc_scan(const Stage2Args& a, const EpilogueArgs& e, hipStream_t st);
// new history = last hist_len samples of [old history | x[0..n)]
// (+ zeroes `zero[0 .. zero_n)`: the raw-peak buffer of the next call)
int launch_hist_roll(const float2* x, const float2* hist_old, float2* hist_new, int hist_len,
                     uint32_t n_total, unsigned* zero, int zero_n, hipStream_t st);

// ---- broadcast FM (WFM / WFM2) at the IF rate fs1 (stage2.hip) -------------------------
struct WfmArgs {
  int nrx, n1;                        // IF-rate samples of this call
  float scale;                        // fs1 / (2*pi*75 kHz)
  float kp, ki, norm, rad2word;       // pilot PLL constants
  uint32_t fword0;                    // 19 kHz at fs1
  const float2* y1[PYSDR_MAX_RX];     // IF IQ, element 0 = first new sample (1-sample prefix)
  float2* y1base[PYSDR_MAX_RX];       // buffer start
  float2* y1dst[PYSDR_MAX_RX];        // buffer start of the NEXT call's IF buffer (= y1base unless the calls overlap): gets the prefix
  float2* w[PYSDR_MAX_RX];            // out: mpx*(1 + 2j*sin(2*theta)) (WFM: imag 0)
  int stereo[PYSDR_MAX_RX];
  double* seed[PYSDR_MAX_RX];         // scan buffers of the Newton-in-time seeds (pll_seed_doubles(m1max) doubles; may be null)
  float* mnT[PYSDR_MAX_RX];           // mpx * norm in the seed kernels' order (pll_seed_index: sample j of lane l of wave v at
                                      // (v * kSeedRun + j) * 64 + l), written by the discriminator kernel; null: no seeds
  RxDevState* state;
  PllPlan pll;                        // pilot PLL segmentation of this call
  int pll_pass;                       // 0: first pass over the segments, 1: the redo pass (only if state.wfm_redo)
};
int launch_wfm_disc(const WfmArgs& a, hipStream_t st);      // polar discriminator + the IF buffer's 1-sample history
bool wfm_any_stereo(const WfmArgs& a);
int launch_wfm_pll(const WfmArgs& a, hipStream_t st);       // the pilot loop's (seeds,) segment walks, check and patch-up passes
size_t pll_seed_doubles(int n1max);                         // pllseed.hip: doubles of scan buffer per RX
// A lane of the seed kernels owns kSeedRun = 64 consecutive samples (every segment length is a multiple of 64: a segment
// starts a lane), a wave 64 lanes = a tile of 4096 samples.  Where sample i of a call sits in mnT: the lanes' j-th samples lie
// side by side, so that the kernels read them coalesced WITHOUT staging through LDS (which cannot be had beside the
// persistent front end of the next call: even 8 KB per workgroup waited for it to end).
constexpr int kSeedRun = 64, kSeedTile = 64 * kSeedRun;
inline size_t pll_seed_mnt_floats(int n1max) { return (((size_t)n1max + kSeedTile - 1) / kSeedTile) * kSeedTile; }
constexpr size_t pll_seed_index(int i) {     // (constexpr: usable from host and device code alike)
  return ((size_t)(i / kSeedTile) * kSeedRun + (size_t)(i % kSeedRun)) * 64 + (size_t)((i / kSeedRun) % 64);
}
int launch_wfm_seed(const WfmArgs& a, hipStream_t st);      // segment start states by two Newton passes over the whole call

// ---- misc kernels (misc.hip) ---------------------------------------------------------
int launch_quad_mixer(const float2* x, float2* y, size_t n, uint32_t phase0, uint32_t fword,
                      hipStream_t st);
int launch_fir_real(const float* xx, const float* h, int nt, float* y, int n, hipStream_t st);
int launch_psd_pre(const float2* x, size_t hop, int nframes, int chunk, int nfft,
                   const float* win, float2* work, int is_complex, hipStream_t st);
int launch_psd_post(const float2* work, int nframes, int nfft, int half, int db, float* out,
                    hipStream_t st);

// fused 32768 -> 65536 PSD path (psdfft.hip): two kernels, `work` = nframes x 65536 complex
// packed: the intermediate as block-scaled 24-bit fixed point (6 bytes per complex) instead of float2
int launch_psd64k(const float2* x, size_t hop, int nframes, const float* win, float2* work,
                  float* out, int db, hipStream_t st, int packed);

}  // namespace pysdr
