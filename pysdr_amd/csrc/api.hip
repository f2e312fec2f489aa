// C ABI of libpysdr_hip.so (include/pysdr_hip.h): context management, the per-call
// launch sequence of the receiver hot path, spectrum (rocFFT), device-memory helpers
// and the RCCL broadcast.  Host-side only; kernels live in mixdec/stage2/misc.hip.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <algorithm>
#include <cstdarg>
#include <map>

#include "common.h"
#include "mixdec_mfma_geom.h"

namespace pysdr {

static thread_local char g_err[2048] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace pysdr

using namespace pysdr;

namespace {

constexpr double kTwo32 = 4294967296.0;
constexpr float kNfmFullScaleDev = 5000.0f;
constexpr double kPllBwHz = 50.0, kPllZeta = 0.7071;
constexpr double kPllZetaPlan = 0.7071;
constexpr double kWfmPllBwHz = 30.0;

constexpr int kPllSegMax = 8192;

// Segmentation of a serial PLL over n samples (PllPlan, common.h).  W = warm-up in samples = `taus`
// time constants 1/(zeta*wn) of the loop; calls shorter than three warm-ups stay one segment.
PllPlan plan_pll(int n, double fs, double bw_hz, double taus, double taus_fast, int t_min, int k_max, uint32_t* seg) {
  PllPlan p;
  const double tau = fs / (kPllZetaPlan * 2.0 * M_PI * bw_hz);
  p.W = ((int)std::ceil(taus * tau) + 63) & ~63;
  p.Wfast = taus_fast > 0 ? (((int)std::ceil(taus_fast * tau) + 63) & ~63) : 0;
  p.Wexact = 0;
  p.Wc_hi = p.Wc_mid = 0;
  p.tail_cap = 0;
  p.coarse_sweeps = 0;
  p.exact_cap = 0;
  p.seeded = 0;
  p.Wseed = 0;
  p.direct = 0;
  if (n < 3 * p.W || k_max <= 1) {
    p.K = 1;
    p.T = (std::max(n, 64) + 63) & ~63;
  } else {
    p.T = std::max(t_min, (((n + k_max - 1) / k_max) + 63) & ~63);
    p.K = (n + p.T - 1) / p.T;
  }
  p.seg = seg;
  p.lin = seg + (size_t)PYSDR_MAX_RX * kPllSegMax * 4;
  return p;
}

inline bool mode_has_agc(int m) {
  return m == PYSDR_AM || m == PYSDR_AM_SYNCH || m == PYSDR_SSB || m == PYSDR_USB ||
         m == PYSDR_LSB || m == PYSDR_CW || m == PYSDR_RTTY;
}
inline int mode_detector(int m) {
  switch (m) {
    case PYSDR_AM: return kDetAbs;
    case PYSDR_AM_SYNCH: return kDetPll;
    case PYSDR_NFM: return kDetFm;
    case PYSDR_CW: return kDetBfo;
    default: return kDetNone;
  }
}

// One polyphase decimator instance (mixdec kernel): geometry, raw-sample history (double
// buffered), LO-modulated taps on the device, absolute input counter.
struct Decim {
  int up = 1, down = 1, ntaps = 0, kdec = 0, kpad = 0, hist_len = 0, max_rx = 0;
  float2* d_hist[2] = {nullptr, nullptr};
  int hist_cur = 0;
  float2* d_taps = nullptr;            // [max_rx][up][kpad]
  std::vector<float2> h_taps;
  unsigned long long s_abs = 0;        // absolute index of the next input sample
  size_t per() const { return (size_t)up * kpad; }
};

struct RxHost {
  int mode = PYSDR_AM;
  double lo_freq = 0;
  uint32_t fword = 0, phase = 0;
  std::vector<double> h;        // prototype, ntaps_dec
  std::vector<double> af;       // complex AF taps, 2*ntaps_af
  double bfo = 0;
  uint32_t bfo_fword = 0;
  bool taps_dirty = true, af_dirty = true, agc_dirty = true;
  unsigned reset_pending = 3;
  int agc_enable = 1;
  float agc_ref = 0.5f;
  float sq_thresh = 0.f;
  float sq_ratio = 0.f;         // ratio squelch armed (> 0: the least sq1 / sq2 that keeps the gate open); takes precedence
  float2* d_y = nullptr;        // [hy + mmax]
  float2* d_y_alt = nullptr;    // the second buffer of the pair, allocated when the context overlaps its calls (pysdr_set_overlap)
  float2* d_ypll = nullptr;     // [hy + mmax], allocated on first AM-Synch use
  float2* d_ypll_alt = nullptr; // its pair (pysdr_set_overlap)
  float2* d_a = nullptr;        // [mmax]
  float* d_am = nullptr;        // [2*mmax]
  float2* d_aftaps = nullptr;   // [ntaps_af rounded up to 4], zero padded
  int taps_real = 0;
  // broadcast FM (WFM / WFM2)
  std::vector<double> wfm_video;   // pre-detection filter at SRATE (ntaps_dec)
  std::vector<double> wfm_resamp;  // fs1 -> FS_OUT prototype (up2 * taps per phase)
  bool wfm_dirty = true;
  Decim wfm_audio;                 // per-RX resampler fs1 -> FS_OUT
  float2* d_y1 = nullptr;          // [2 + m1max] IF-rate IQ (1-sample history in slot 1)
  float2* d_y1_alt = nullptr;      // its pair (pysdr_set_overlap)
  float2* d_w = nullptr;           // [m1max] composite * (1 + 2j sin 2theta)
  float2* d_w_alt = nullptr;       // its pair (pysdr_set_overlap)
  double* d_seed = nullptr;        // WFM2: scan buffers of the pilot loop's Newton-in-time seeds (pllseed.hip), allocated on first stereo use
  float* d_mnt[2] = {nullptr, nullptr};   // ... and mpx * norm in the seed kernels' order, written by the discriminator: a pair like d_w
};

// What one pysdr_process_batch call uses of the receivers' host-side state.  Taken under
// c->mu by apply_pending, so that a setter running on another thread (gui.py:1713,1938)
// between two lines of the launch sequence can only affect the NEXT call: the taps on the
// device, the NCO word, the detector kind and the buffers it needs always belong together.
struct RxSnap {
  int mode = PYSDR_AM;
  uint32_t fword = 0, phase = 0, bfo_fword = 0;
  float sq_thresh = 0.f, sq_ratio = 0.f;
  int taps_real = 0;
  float2 *d_y = nullptr, *d_ypll = nullptr, *d_a = nullptr, *d_y1 = nullptr, *d_w = nullptr;   // d_y / d_ypll / d_y1 / d_w: THIS call's buffer of each pair
  float2 *d_y_next = nullptr, *d_ypll_next = nullptr, *d_y1_next = nullptr;   // the next call's (the same one unless the calls overlap): gets the history prefix
  double* d_seed = nullptr;
  float* d_mnt = nullptr;
  float* d_am = nullptr;
  float2* d_aftaps = nullptr;
};
struct CallSnap {
  int nrx = 0, nwfm = 0;
  bool use2 = false;             // this call's tail is deferred (pysdr_set_overlap)
  RxSnap rx[PYSDR_MAX_RX];
};

// The tail of a call (T in pysdr_ctx): everything it needs that the front part of the call worked out.
struct TailJob {
  bool valid = false;
  CallSnap snap;
  bool wfm = false;
  bool wait_pll = false;         // the call's loop walks run on stream2: the tail waits for ev_pll[par]
  int par = 0, nchunks = 0, n1 = 0;
  size_t chunk_len = 0, n = 0;
  unsigned long long s0 = 0;
  Stage2Args s;                  // narrow-band: complete.  Broadcast FM: the output counts come from the audio resampler, which is part of the tail
  hipEvent_t ev_end = nullptr;   // the profile's last mark of that call (nullptr: not profiled)
};

}  // namespace

struct pysdr_ctx {
  pysdr_cfg cfg;
  int nrx = 0;
  RxHost rx[PYSDR_MAX_RX];
  std::mutex mu;
  // Everything that queues work on the context's streams or moves the deferred tail -- process / process_batch, fetch, sync,
  // the state getters (they queue the deferred tail: flush_tail), get_elapsed_ms, set_overlap, a spectrum ordering itself
  // against the context, the ingest ring's submit / collect, the broadcast -- runs under this lock: the reference reads the
  // AGC fields from its GUI / watchdog thread (watchdog.py:298-302) while the RX thread is inside demod_data, and a getter
  // that ran the tail beside pysdr_process_batch's `c->tail = job` could run it twice or from a half-copied job.
  // Recursive: pysdr_process is process_batch + fetch.  The setters keep to `mu` (they only mark work as pending).
  std::recursive_mutex run_mu;
  hipStream_t stream = nullptr;
  // pysdr_set_overlap: a call is three groups of launches --
  //   F  the front end (mix + decimate; + the short parallel kernels in front of a serial loop: arg y for the carrier
  //      loop, the discriminator for broadcast FM),
  //   P  the segment walks of a serial loop (AM-Synch carrier PLL, WFM2 pilot PLL): a few thousand single-wave chains,
  //      latency bound, no LDS, a third to a half of such a call's time,
  //   T  the tail (audio resampler of broadcast FM, detector + AF FIR, AGC, output): LDS-tiled and throughput bound --
  // and in the single-stream form they run F P T, F P T, ...  Overlapped, P(k) goes on `stream2` and `stream` runs
  //   F(k)  [wait P(k-1)]  T(k-1)  F(k+1)  [wait P(k)]  T(k) ...
  // i.e. the TAIL OF A CALL IS DEFERRED until the front end of the next one has been queued (or until anything asks
  // for its results: flush_tail), so that the walks of call k run beside T(k-1) and F(k+1).  What F writes and T reads a
  // call later -- the FS_OUT-rate IQ y, the IF-rate IQ y1, the loops' buffers -- alternates between two buffers (`par`);
  // each history prefix is rolled into the OTHER buffer.  Two events order it: ev_front (F(k) done -> P(k) may start) and
  // ev_pll[par] (P(k) done -> T(k) may start); everything else is stream order.
  // Round 5's first form put P AND T on stream2; measured (profiles/r05_overlap_*.txt): the AF FIR cannot start while the
  // persistent workgroups of a front end hold their CUs' LDS, so T took turns with F whatever the streams said, and a P
  // that shares its SIMDs with matrix-core waves runs 2.2-2.5x longer (a vector instruction waits for the MFMA in
  // front of it) -- P + T behind one another on stream2 was then LONGER than F, and the front end idled.
  // WHICH calls: those with a serial loop in them (overlap = 1); overlap = 2 defers every call's tail (tests: nothing
  // runs on stream2 then, the buffers and the deferral are exercised all the same).  A change of form between two calls
  // flushes and drains.
  hipStream_t stream2 = nullptr;
  int overlap = 0;               // pysdr_set_overlap: 0 off, 1 the calls with a serial loop, 2 every call
  int overlap_env = -1;          // PYSDR_OVERLAP=0/1/2 (under PYSDR_TUNING): pysdr_set_overlap is overruled (A/B runs, the test suite in every form)
  bool use2 = false;             // the form of the LAST call
  int par = 0;                   // buffer of each pair the NEXT call's front end writes
  int last_par = 0;              // ... and the one the last finished call wrote (pysdr_fetch reads its IQ)
  hipEvent_t ev_pll[2] = {nullptr, nullptr};
  int n_ingest = 0;              // ingest rings on this context (they run it single-stream)
  int tail_first = -1;           // PYSDR_OVERLAP_ORDER=0/1 (A/B): the deferred tail behind / in front of the next call's walks; -1: by mode
  TailJob tail;                  // the deferred tail of the last call (tail.valid)
  int hy = 0, mmax = 0;
  size_t cap_samples = 0;
  Decim main;                    // SRATE -> FS_OUT (UP/DOWN) for the narrow-band modes
  Decim wfm_front;               // SRATE -> fs1 = SRATE/d1 for WFM/WFM2
  int d1 = 0, up2 = 0, down2 = 0, m1max = 0;
  int last_wfm = 0;              // the last call ran the broadcast-FM pipeline
  float2* d_stage = nullptr;
  size_t stage_cap = 0;
  unsigned* d_peak = nullptr;    // [max_chunks] raw-chunk peaks of the last call = d_peak2[peak_cur]
  unsigned* d_peak2[2] = {nullptr, nullptr};   // two buffers: the history-roll kernel of a call zeroes the OTHER one for the next
  int peak_cur = 0;              //   call (a memset per call was one more kernel + 4 us on the stream)
  bool peak_clean[2] = {true, true};   // that buffer is all zero (a call that failed before its history roll leaves the other dirty)
#ifdef PYSDR_DIAG
  unsigned long long* d_stamps = nullptr;   // mixdec phase stamps (PYSDR_DEBUG_FLAGS & 256)
  unsigned long long* d_mm_stamps = nullptr;   // mixdec_mfma: per-workgroup placement and clocks (PYSDR_DEBUG_FLAGS & 512), [1024][24]
#endif
  unsigned* d_peak_scratch = nullptr;  // [1] sink for decimators whose raw peak is not wanted
  unsigned* d_blkpeak = nullptr; // [MAX_RX][max_chunks]
  float* d_gain = nullptr;       // [MAX_RX][max_chunks]
  float* d_blknoise = nullptr;   // [MAX_RX][max_chunks]
  float* d_blknoise2 = nullptr;  // [MAX_RX][max_chunks] ratio squelch: the low-pass envelope's block sums (allocated when first armed)
  float* d_sqtaps = nullptr;     // [2][kSqTapsMax] its two FIRs
  int sq_ntaps = 0;
  unsigned* d_blkcnt = nullptr;  // [MAX_RX][max_chunks]
  RxDevState* d_state = nullptr; // [MAX_RX]
  uint32_t* d_pllseg = nullptr;  // [MAX_RX][kPllSegMax][4] start/end states of the time-parallel PLLs + [MAX_RX][kPllSegMax] flags (PllPlan::lin)
  // last call
  int last_nout = 0, last_nchunks = 0, last_nrx = 0;
  size_t last_chunk_len = 0;
  unsigned long long last_s0 = 0;
  int last_complex[PYSDR_MAX_RX] = {0};
  // tuning / profiling
  int tile_bytes = 0, threads = 1024;  // per LDS buffer (two per workgroup); 0 = as large as fits
  int wgs_per_cu = 1, num_cus = 256;
  int grid_override = 0;               // PYSDR_MIXDEC_GRID: workgroups of the mix+decimate launches (tests: many tiles per workgroup in a small call)
  int resamp_plain = 0;                // PYSDR_RESAMP_PLAIN: the audio resampler of broadcast FM 0 = a wave per branch, taps in scalar registers (resamp_wave_kernel),
                                       // 1 = one output per thread (resamp_small_kernel), 2 = a half-wave per branch (resamp_branch_kernel) (A/B)
  int mfma_enable = 1;                 // long single-RX prototypes on the matrix cores (mixdec_mfma.hip); 0: VALU form (A/B)
  int dbg_flags = 0, yflush_cap = 0;      // tuning / diagnostic switches, read from the environment once
  // carrier-PLL segmentation (PYSDR_AM_PLL = "taus,taus_exact,coarse_sweeps,kmax,tmin" overrides for A/B runs): warm-up of 16
  // time constants from the block mean of the signal's own phase (joins 7-40 words of 2^32 against a tolerance of 1024 in the
  // NumPy model of the sweeps, scripts/experiments/am_pll_sweeps.py; 14 leave up to 730 on a noisy carrier 40 Hz off tune),
  // the first 11 of them at 4 sweeps per block (5: joins 5 -> 6 words on the bench's carrier, 58 -> 316 in the model's noisy
  // one, segment kernel 105 -> 97 us; 3: the noisy model leaves 2700), the last 5 to the fixed point (8-10 sweeps)
  double am_taus = 16.0, am_taus_exact = 5.0;
  int am_coarse_sweeps = 4, am_kmax = 2048, am_tmin = 512;
  int am_seeded = 1, am_wseed = 0;     // PYSDR_AM_SEED=on[,walked]: warm-ups by the linear solve where the window allows (stage2.hip am_linear_start)
  int am_direct = 1;                   // PYSDR_AM_DIRECT=0: blocks start from the free-running line (A/B)
  int am_phase_on_2 = 1;               // PYSDR_AM_PHASE_STREAM=0: arg y on the front stream in the overlapped form too (A/B)
  int pll_kmax = 0;                       // pysdr_set_pll_segments: 0 = default, 1 = serial
  // pilot-PLL segmentation (PYSDR_WFM_PLL = "taus,taus_fast,taus_exact,coarse_sweeps,kmax,tmin,exact_cap" overrides for A/B runs)
  // measured on MI355X (bench.py --workload c4, scripts/diag/pll_sweep.sh; front end ms per 2048 chunks):
  //   exact warm-ups 1.70 | 3 coarse sweeps + 6 / 5 / 4 tau exact 1.56 / 1.57 / 1.54 | 4 sweeps 1.61 | 2 sweeps: every join
  //   misses (180 ms of serial patching) | 3 sweeps + 3 tau exact: 115 joins miss | fast warm-up 11 / 9 tau: the check pass
  //   redoes the call (1.85 / 2.4) | 4096 / 3072 / 1024 segments with exact warm-ups 1.96 / 1.78 / 1.71 (2048: 1.70)
  // round 4, after the sweeps went from 35 to 22 vector instructions (stage2.hip wfm_pll_walk; scripts/diag/c4_kt.sh, us per
  // segment-kernel launch, same box): 1024 / 1280 / 1536 / 1792 / 2048 segments 388 / 365 / 349-351 / 379 / 370 -- the walk
  // is now as much the latency of one segment's chain as the SIMDs' issue rate, and fewer, longer segments walk less warm-up
  // judged by the widest join they leave (pysdr_pll_join_margin, tolerance 512 words; scripts/diag/c4_pll4.sh): exact tail 5 / 4 tau
  // 107 / 122 words (C4 step 1.170 / 1.160 ms); two sweeps over the first half of the coarse part 280; exact tail at 4 sweeps 422;
  // both: joins miss; warm-up of 12 / 14 tau instead of 13: 283 / 104
  double wfm_taus = 20.0, wfm_taus_fast = 13.0, wfm_taus_exact = 4.0;
  int wfm_coarse_sweeps = 3, wfm_kmax = 1536, wfm_tmin = 2048;
  double wfm_taus_hi = 0.0, wfm_taus_mid = 0.0;   // staged coarse warm-up (PllPlan::Wc_hi / Wc_mid) in time constants; 0, 0: one stage
  int wfm_tail_cap = 0;                // sweeps per block of the exact tail of a warm-up (0: wfm_exact_cap)
  int wfm_exact_cap = 5;               // sweeps per block of the pilot loop's exact walks (0: to the bit-stable fixed point)
  // Round 5: the segments of the pilot loop start from Newton-in-time seeds (pllseed.hip; two linearised passes over the call by
  // parallel scans of affine maps: within ~30 words of 2^32 of the exact walk) instead of a 13-tau warm-up, whenever the previous
  // call left a mean phase increment; PYSDR_WFM_SEED="0" switches it off, "1,n" walks n samples from the seed first (A/B)
  int wfm_seeded = 1, wfm_wseed = 0;
  int profile = 0;
  static constexpr int kSlots = 64;       // ring of per-call event sets (profiling)
  hipEvent_t ev[kSlots][4] = {};
  hipEvent_t ev_front = nullptr;          // the input of the last call has been consumed (front end done)
  hipEvent_t front_marker = nullptr;      // the event that marks it for the LAST call (ev_front, the profile's, or none)
  int front_wanted = 0;                   // a spectrum has ordered itself behind the front end: keep recording ev_front
  unsigned long long ncalls = 0;
  // RCCL
  void* rccl_lib = nullptr;
  ncclComm_t comm = nullptr;
};

struct pysdr_spectrum {
  int device = 0, chunk = 0, nfft = 0, max_frames = 0;
  hipStream_t stream = nullptr;
  float* d_win = nullptr;
  float2* d_work = nullptr;   // [work_frames][nfft], grown on demand
  size_t work_frames = 0;
  float2* d_in = nullptr;     // [chunk] staging for host frames
  float* d_out = nullptr;     // [nfft] staging for host frames
  void* d_fftwork = nullptr;
  size_t fftwork_bytes = 0;
  std::map<int, rocfft_plan> plans;
  rocfft_execution_info info = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipEvent_t ev_order = nullptr;
  bool force_rocfft = false;  // PYSDR_PSD_ROCFFT: rocFFT even for the 32768 -> 65536 size
  int group = 0;              // frames per launch pair of the four-step path (PYSDR_PSD_GROUP); 0 = 480 with the 24-bit intermediate, 448 with float2
  bool ran = false;           // spectrum_run has recorded ev[1] at least once
  int packed = 1;             // four-step intermediate as block-scaled 24-bit fixed point (psdfft.hip; PYSDR_PSD_PACKED=0: float2)
  // PYSDR_PSD_STREAMS=2: the groups alternate between two streams, each with its own half-size intermediate
  // (2 x group/2 frames = the same Infinity Cache footprint), so that the columns of one group run beside
  // the rows of the other and the kernel boundaries of one stream hide behind the other's kernels
  static constexpr int kMaxStreams = 4;
  int nstreams = 2;
  hipStream_t xstream[kMaxStreams] = {};     // [0] unused (= stream)
  float2* xwork[kMaxStreams] = {};           // [0] unused (= d_work)
  size_t xwork_frames = 0;
  hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {};
};

// N4: ingest ring.  Pinned host chunk buffers the device reads directly over PCIe (async H2D on
// its own stream into two device staging buffers) and pinned per-slot result buffers.
struct pysdr_ingest {
  pysdr_ctx* c = nullptr;
  int nslots = 0;
  int chunks_per_slot = 1;
  size_t cap = 0;                         // samples per slot buffer (chunks_per_slot chunks)
  int ocap = 0;                           // outputs per RX per chunk
  hipStream_t copy_stream = nullptr;
  std::vector<float2*> h_in;              // [nslots] pinned
  float2* d_in[2] = {nullptr, nullptr};
  hipEvent_t ev_free[2] = {nullptr, nullptr};   // the kernels that read d_in[i] have finished
  bool used[2] = {false, false};
  std::vector<hipEvent_t> ev_copied, ev_done;   // [nslots]
  std::vector<float*> h_am, h_iq;         // [nslots*MAX_RX] pinned
  std::vector<float*> h_peak;             // [nslots] pinned
  std::vector<int> n_out, in_flight, n_chunks;
  std::vector<int> nrx;                   // [nslots] sub-receivers of the slot's submit (pysdr_rx_add may run before its collect)
  std::vector<std::vector<int>> chunk_nout;     // [nslots][chunks of the slot's last submit]
  std::vector<int> cx;                    // [nslots*MAX_RX]
  unsigned long long seq = 0;
  bool counted = false;                   // this ring is in its context's n_ingest
};

namespace {

std::mutex g_rocfft_mu;
int g_rocfft_users = 0;

int use_device(int dev) {
  hipError_t e = hipSetDevice(dev);
  if (e != hipSuccess) {
    set_last_error("hipSetDevice(%d): %s", dev, hipGetErrorString(e));
    return PYSDR_ERR_NO_DEVICE;
  }
  return PYSDR_OK;
}

// The A/B and tuning switches of INTEGRATION.md ("tuning environment") are read ONLY when PYSDR_TUNING=1 is set as
// well: a drop-in library must not change kernels because of a variable that happens to be in the environment.
const char* tuning_env(const char* name) {
  const char* on = getenv("PYSDR_TUNING");
  if (!on || atoi(on) <= 0) return nullptr;
  return getenv(name);
}

// g[p][k] = h[p + up*k] * exp(-j*w*k), w = 2*pi*fword/2^32
void build_taps(const Decim& d, const double* h, int nt, uint32_t fword, float2* out) {
  for (int p = 0; p < d.up; ++p) {
    for (int k = 0; k < d.kpad; ++k) {
      const int j = p + d.up * k;
      float2 g = make_float2(0.f, 0.f);
      if (j < nt) {
        const uint32_t ph = (uint32_t)((uint64_t)fword * (uint64_t)k);   // mod 2^32
        const double ang = -2.0 * M_PI * ((double)(int32_t)ph / kTwo32);
        g.x = (float)(h[j] * std::cos(ang));
        g.y = (float)(h[j] * std::sin(ang));
      }
      out[(size_t)p * d.kpad + k] = g;
    }
  }
}

// All initial fills go to the context's own stream: it is a non-blocking stream, i.e. NOT ordered
// against the null stream a plain hipMemset runs on, and the first tap upload follows on it.
int decim_init(Decim& d, int up, int down, int ntaps, int max_rx, hipStream_t st) {
  d.up = up; d.down = down; d.ntaps = ntaps; d.max_rx = max_rx;
  d.kdec = (ntaps + up - 1) / up;
  d.kpad = (d.kdec + 15) / 16 * 16;
  d.hist_len = d.kpad + 2;
  d.h_taps.assign((size_t)max_rx * d.per(), make_float2(0.f, 0.f));
  for (int i = 0; i < 2; ++i) {
    PYSDR_HIP_CHECK(hipMalloc(&d.d_hist[i], d.hist_len * sizeof(float2)));
    PYSDR_HIP_CHECK(hipMemsetAsync(d.d_hist[i], 0, d.hist_len * sizeof(float2), st));
  }
  PYSDR_HIP_CHECK(hipMalloc(&d.d_taps, (size_t)max_rx * d.per() * sizeof(float2)));
  PYSDR_HIP_CHECK(hipMemsetAsync(d.d_taps, 0, (size_t)max_rx * d.per() * sizeof(float2), st));
  return PYSDR_OK;
}

void decim_free(Decim& d) {
  for (int i = 0; i < 2; ++i) if (d.d_hist[i]) { (void)hipFree(d.d_hist[i]); d.d_hist[i] = nullptr; }
  if (d.d_taps) { (void)hipFree(d.d_taps); d.d_taps = nullptr; }
}

int decim_upload_taps(pysdr_ctx* c, Decim& d, int slot, const double* h, int nt, uint32_t fword) {
  float2* dst = d.h_taps.data() + (size_t)slot * d.per();
  build_taps(d, h, nt, fword, dst);
  PYSDR_HIP_CHECK(hipMemcpyAsync(d.d_taps + (size_t)slot * d.per(), dst, d.per() * sizeof(float2),
                                 hipMemcpyHostToDevice, c->stream));
  return PYSDR_OK;
}

// The divisor of SRATE whose quotient is closest to 250 kHz
int wfm_if_decim(double srate) {
  const long long sr = (long long)std::llround(srate);
  long long best = 1;
  double bd = -1.0;
  const long long lim = std::max<long long>(2, sr / 100000);
  for (long long d = 1; d <= lim; ++d) {
    if (sr % d) continue;
    const double err = std::fabs((double)sr / (double)d - 250e3);
    if (bd < 0 || err < bd) { best = d; bd = err; }
  }
  return (int)best;
}

inline bool is_wfm(int m) { return m == PYSDR_WFM || m == PYSDR_WFM2; }

// One launch of the fused mix+decimate kernel on `nrx` receivers that share `d`.
struct DecimResult { int n_out; uint32_t t0; unsigned long long m0; };
int decim_run(pysdr_ctx* c, Decim& d, const float2* d_x, size_t n, int nrx, float2* const* y,
              const uint32_t* phase0, const uint32_t* fword, unsigned* peak, size_t chunk_len,
              int y_cap, DecimResult* res, hipStream_t st) {
  const int up = d.up, down = d.down;
  const unsigned long long s0 = d.s_abs, s1 = s0 + n;
  const unsigned long long m0 = (s0 * up + down - 1) / down, m1 = (s1 * up + down - 1) / down;
  const int n_out = (int)(m1 - m0);
  if (n_out > y_cap) { set_last_error("decimator: n_out %d > capacity %d", n_out, y_cap); return PYSDR_ERR_STATE; }
  if ((double)n * up + down >= 2147483647.0) { set_last_error("decimator: call too long for 31-bit indices"); return PYSDR_ERR_ARG; }
  // One RX with a long prototype at a rate that has an instantiation: the matrix-core form (mixdec_mfma.hip).  The
  // choice depends on the decimator's shape only -- never on the call -- so every call of a stream sums in the
  // same order (batch == chunk by chunk bit for bit), whatever the alignment of the caller's device pointer (the
  // LDS-DMA takes any 4-byte aligned source: scripts/diag/glds_align_test.hip).
  const int mshape = (nrx == 1 && c->mfma_enable) ? mixdec_mfma_shape(up, down, d.kdec) : -1;
  MfmaPlan plan;
  if (mshape >= 0 && !mixdec_mfma_plan(mshape, s0, m0, n, &plan)) {
    // (falling through to the vector form here would sum THIS call in another order than its neighbours: an error instead)
    set_last_error("decimator: call of %zu samples too long for the matrix-core form (2^30 samples per call)", n);
    return PYSDR_ERR_ARG;
  }
  if (mshape >= 0) {
    MixMfmaArgs b;
    memset(&b, 0, sizeof(b));
    b.x = d_x;
    b.hist = d.d_hist[d.hist_cur];
    b.hist_len = d.hist_len;
    b.n_total = (uint32_t)n;
    b.n_out = n_out;
    b.origin_rel0 = plan.origin_rel0; b.d = plan.d; b.nrel0 = plan.nrel0; b.mrel0 = plan.mrel0; b.ntiles = plan.ntiles;
    b.kpad = d.kpad;
    b.taps = d.d_taps;
    b.y = y[0];
    b.phase0 = phase0[0]; b.fword = fword[0];
    b.peak = peak ? peak : c->d_peak_scratch;
    b.chunk_len = peak ? (uint32_t)chunk_len : (uint32_t)std::max<size_t>(n, 1);
    b.magic_chunk = (b.chunk_len == 1) ? 0u : (uint32_t)(4294967296ULL / (unsigned long long)b.chunk_len) + 1u;
    // the history roll rides in the launch (hist_roll.h): the epilogue waves of workgroup 0 during their idle first trip
    b.hist_new = d.d_hist[d.hist_cur ^ 1];
    b.zero = peak ? c->d_peak2[c->peak_cur ^ 1] : nullptr;
    b.zero_n = peak ? c->cfg.max_chunks : 0;
#ifdef PYSDR_DIAG
    if ((c->dbg_flags & 512) && !c->d_mm_stamps) {
      PYSDR_HIP_CHECK(hipMalloc(&c->d_mm_stamps, 1024 * 24 * sizeof(unsigned long long)));
      PYSDR_HIP_CHECK(hipMemset(c->d_mm_stamps, 0, 1024 * 24 * sizeof(unsigned long long)));
    }
    b.wg_stamps = (c->dbg_flags & 512) ? c->d_mm_stamps : nullptr;
#endif
    int rc = launch_mixdec_mfma(mshape, b, c->grid_override > 0 ? c->grid_override : c->num_cus, st);
    if (rc) return rc;
    if (peak) c->peak_clean[c->peak_cur ^ 1] = true;
    d.hist_cur ^= 1;
    d.s_abs = s1;
    if (res) { res->n_out = n_out; res->t0 = (uint32_t)(m0 * down - s0 * up); res->m0 = m0; }
    return PYSDR_OK;
  }
  MixDecArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_x;
  a.hist = d.d_hist[d.hist_cur];
  a.hist_len = d.hist_len;
  a.aligned16 = ((reinterpret_cast<uintptr_t>(d_x) & 15u) == 0) ? 1 : 0;
  a.n_total = (uint32_t)n;
  a.t0 = (uint32_t)(m0 * down - s0 * up);
  a.n_out = n_out;
  a.up = up; a.down = down;
  a.kpad = d.kpad;
  a.magic = (up == 1) ? 0u : (uint32_t)(4294967296ULL / (unsigned)up) + 1u;
  a.nrx = nrx;
  const int ratio = (down + up - 1) / up;
  // Which instantiation runs this shape (mixdec.hip md_dispatch): its thread count, and whether its waves can hold their taps
  // in registers.  The long-prototype multi-RX shapes (768 threads) then read the taps from memory once per launch and the
  // LDS holds tiles and the output stage only; the BASELINE shapes (1024 threads) keep their taps in LDS as before.
  const MixdecVariant var = mixdec_variant(nrx, up, d.kpad, c->threads);
  const int threads = std::min(c->threads, var.tpb);
  const int hold_step = (threads / 64) / (up * var.nh);           // waves per (branch, RX group); 0: not enough waves to hold
  bool taps_lds = !(var.tpb != 1024 && var.can_hold && hold_step >= 1);
  // two tile buffers + the taps must fit the LDS share of one workgroup
  const int wgs = c->wgs_per_cu;
  const long lds_share = (160L * 1024) / wgs - 512;
  // tile_bytes == 0: the largest tile the LDS share allows (fewer tiles = less per-tile
  // scalar work, the kernel's scarcest resource).  What is left over holds the output stage
  // (yflush tiles of outputs per RX); if that is less than 4 tiles' worth the tile shrinks.
  const long slack = d.kpad + 2L * ratio + 8 + 128;   // halo, ownership overhang, whole 64-pair DMA pieces
  long cap = 0, tile_out = 0, yflush = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t taps_bytes = taps_lds ? (size_t)nrx * up * d.kpad * sizeof(float2) : 0;
    long reserve = 0;
    for (int pass = 0; pass < 2; ++pass) {
      cap = c->tile_bytes > 0 ? c->tile_bytes / (long)sizeof(float2) : (1L << 30);
      if (2 * cap * (long)sizeof(float2) + (long)taps_bytes + reserve > lds_share)
        cap = (lds_share - (long)taps_bytes - reserve) / (2 * (long)sizeof(float2));
      tile_out = ((cap - slack) * up) / down;
      // whole quads of every polyphase branch -- and, where the waves hold their taps, the same number of quads for every
      // wave of a (branch, RX group): a tile of 60 outputs at 3/500 would give one wave in four a second task
      const long quantum = 4L * up * ((!taps_lds && tile_out >= 4L * up * hold_step) ? hold_step : 1);
      if (tile_out >= quantum) tile_out -= tile_out % quantum;
      tile_out &= ~1L;
      if (tile_out < 2) {
        tile_out = 2;
        cap = slack + (2L * down + up - 1) / up + 2;
      } else {
        cap = std::min(cap, slack + (tile_out * down + up - 1) / up + 2);   // no more LDS than the tile needs
      }
      cap = (cap + 1) & ~1L;
      const long per_tile = (long)nrx * tile_out * (long)sizeof(float2);
      const long left = lds_share - (long)taps_bytes - 2 * cap * (long)sizeof(float2);
      yflush = left > 0 ? std::min(16L, left / per_tile) : 0;
      if (yflush >= 4 || pass == 1) break;
      reserve = 4 * per_tile;
    }
    // hold mode needs every tile to start on the same polyphase branch; a tile too small for that reads its taps from LDS
    if (taps_lds || (tile_out % up) == 0) break;
    taps_lds = true;
  }
  if (yflush < 1) {
    set_last_error("decimator: filter (%d taps, %d rx, up %d) does not fit LDS", d.ntaps, nrx, up);
    return PYSDR_ERR_ARG;
  }
  a.taps_lds = taps_lds ? 1 : 0;
  a.tile_out = (int)tile_out;
  a.tile_cap = (int)cap;
  if (c->yflush_cap > 0 && c->yflush_cap < yflush) yflush = c->yflush_cap;
  a.yflush = (int)yflush;
  a.ycap = (int)(yflush * tile_out);
  a.tpc = (int)((((tile_out + up - 1) / up) + 3) >> 2);
  a.ntasks = up * a.tpc;
  a.magic_tpc = (a.tpc == 1) ? 0u : (uint32_t)(4294967296ULL / (unsigned)a.tpc) + 1u;
  a.dq_tile = (int)((tile_out * down) / up);
  a.dr_tile = (int)((tile_out * down) % up);
  a.dq_last = (int)(((tile_out - 1) * down) / up);
  a.dr_last = (int)(((tile_out - 1) * down) % up);
  a.ntiles = n_out > 0 ? (n_out + a.tile_out - 1) / a.tile_out : 1;
  a.taps = d.d_taps;
  for (int r = 0; r < nrx; ++r) { a.y[r] = y[r]; a.phase0[r] = phase0[r]; a.fword[r] = fword[r]; }
  a.peak = peak ? peak : c->d_peak_scratch;
  a.chunk_len = peak ? (uint32_t)chunk_len : (uint32_t)std::max<size_t>(n, 1);
  a.magic_chunk = (a.chunk_len == 1) ? 0u : (uint32_t)(4294967296ULL / (unsigned long long)a.chunk_len) + 1u;
  a.dbg = c->dbg_flags;
#ifdef PYSDR_DIAG
  if ((c->dbg_flags & 256) && !c->d_stamps) {
    PYSDR_HIP_CHECK(hipMalloc(&c->d_stamps, 2 * 16 * 24 * 8 * sizeof(unsigned long long)));
    PYSDR_HIP_CHECK(hipMemset(c->d_stamps, 0, 2 * 16 * 24 * 8 * sizeof(unsigned long long)));
  }
  a.stamps = (c->dbg_flags & 256) ? c->d_stamps : nullptr;
#endif
  // one RX, no raw peak wanted, short prototype and a small DOWN/UP (the fs1 -> FS_OUT stage of broadcast FM): one thread
  // per output (resamp_small.hip).  Decided by the decimator's shape and its call site only, never by the call.
  const bool small = (nrx == 1 && !peak && resamp_small_span(up, down, d.kpad) > 0);
  // the history roll rides in the launch (hist_roll.h: workgroup 0, while its first tile's copies are in flight)
  a.hist_new = d.d_hist[d.hist_cur ^ 1];
  a.zero = peak ? c->d_peak2[c->peak_cur ^ 1] : nullptr;
  a.zero_n = peak ? c->cfg.max_chunks : 0;
  int rc = small ? launch_resamp_small(a, c->grid_override, c->resamp_plain, st)
                 : launch_mixdec(a, c->threads, c->grid_override > 0 ? c->grid_override : c->num_cus * wgs, st);   // (the REQUESTED thread count: it selects the instantiation, which then clamps it to its own)
  if (rc) return rc;
  if (small && n_out <= 0) {          // the resampler starts no kernel for a call without outputs: the roll on its own
    rc = launch_hist_roll(d_x, d.d_hist[d.hist_cur], d.d_hist[d.hist_cur ^ 1], d.hist_len, (uint32_t)n, a.zero, a.zero_n, st);
    if (rc) return rc;
  }
  if (peak) c->peak_clean[c->peak_cur ^ 1] = true;
  d.hist_cur ^= 1;
  d.s_abs = s1;
  if (res) { res->n_out = n_out; res->t0 = a.t0; res->m0 = m0; }
  return PYSDR_OK;
}

int wfm_setup(pysdr_ctx* c) {
  if (c->d1) return PYSDR_OK;
  c->d1 = wfm_if_decim(c->cfg.srate);
  const double fs1 = c->cfg.srate / c->d1;
  const long long a = (long long)std::llround(std::floor(c->cfg.srate * c->cfg.up / c->cfg.down));
  const long long b = (long long)std::llround(fs1);
  long long g = std::__gcd(a, b);
  c->up2 = (int)(a / g);
  c->down2 = (int)(b / g);
  c->m1max = (int)(c->cap_samples / (size_t)c->d1) + 4;
  return decim_init(c->wfm_front, 1, c->d1, c->cfg.ntaps_dec, PYSDR_MAX_RX, c->stream);
}

int flush_tail(pysdr_ctx* c);

int apply_pending(pysdr_ctx* c, CallSnap* snap) {
  std::lock_guard<std::mutex> lk(c->mu);
  snap->nrx = c->nrx;
  snap->nwfm = 0;
  for (int r = 0; r < c->nrx; ++r) snap->nwfm += is_wfm(c->rx[r].mode) ? 1 : 0;
  if (snap->nwfm != 0 && snap->nwfm != c->nrx) {
    // the reference's mode is global (P.MODE) and the rate-reduction order differs for
    // broadcast FM (receiver.py:718-719): one context runs one pipeline
    set_last_error("pysdr_process_batch: WFM/WFM2 cannot be mixed with narrow-band modes in one context");
    return PYSDR_ERR_STATE;
  }
  // Overlapping calls: a deferred tail belongs to the PREVIOUS call and must see the taps / AGC settings / buffers of that
  // call, so whatever a setter left to do (taps, resets, buffers) and any change of form first flushes it and drains both
  // streams -- it is rare, and everything below may then go on `stream` as in the single-stream form (drained again at
  // the end, so that the loop walks on stream2 see it).
  bool drained = false;
  snap->use2 = c->overlap >= 2;
  for (int r = 0; r < c->nrx && c->overlap == 1; ++r)
    snap->use2 |= (c->rx[r].mode == PYSDR_AM_SYNCH || c->rx[r].mode == PYSDR_WFM2);
  // the second buffer of every pair exists while the calls overlap AND afterwards for as long as the current buffer is the
  // second one (par only moves in overlapped calls; a single-stream call stays on whichever buffer is current)
  const bool need_alt = snap->use2 || c->par != 0;
  bool dirty = snap->use2 != c->use2;
  for (int r = 0; r < c->nrx && (need_alt || c->tail.valid); ++r) {
    const RxHost& x = c->rx[r];
    dirty |= x.taps_dirty || x.af_dirty || x.agc_dirty || x.reset_pending != 0 || (is_wfm(x.mode) && x.wfm_dirty) ||
             (x.mode == PYSDR_AM_SYNCH && (x.d_ypll == nullptr || (need_alt && x.d_ypll_alt == nullptr))) ||
             (x.mode == PYSDR_WFM2 && x.d_seed == nullptr && c->wfm_seeded) ||
             (need_alt && (x.d_y_alt == nullptr || (x.d_y1 != nullptr && x.d_y1_alt == nullptr)));
  }
  if (dirty) {
    int rc = flush_tail(c);
    if (rc) return rc;
    PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->stream2) PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream2));
    if (snap->use2 && !c->stream2) {
      // The second stream exists from the first call that uses it, not from pysdr_set_overlap: the runtime maps streams
      // onto a handful of hardware queues, and an IDLE fifth stream beside a spectrum's two made the PSD's streams share
      // one (measured, same box: C3's PSD call 2.39 -> 2.95 ms, at a low priority 3.6 ms; scripts/diag/ab_r04.sh).
      // LOW priority: where both have work the dispatcher places the front end's workgroups first (they are persistent
      // over a static share of the tiles: one that starts late ends late, and the launch with it).
      int lo = 0, hi = 0;
      PYSDR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      const char* pe = tuning_env("PYSDR_OVERLAP_PRIO");
      const int prio = (pe && *pe) ? atoi(pe) : lo;
      PYSDR_HIP_CHECK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio));
      for (int i = 0; i < 2; ++i) PYSDR_HIP_CHECK(hipEventCreateWithFlags(&c->ev_pll[i], hipEventDisableTiming));
    }
    c->use2 = snap->use2;
    drained = true;
  }
  for (int r = 0; r < c->nrx; ++r) {
    RxHost& x = c->rx[r];
    if (x.taps_dirty) {
      int rc = decim_upload_taps(c, c->main, r, x.h.data(), c->cfg.ntaps_dec, x.fword);
      if (rc) return rc;
      x.taps_dirty = false;
      x.wfm_dirty = true;             // the LO moved: the WFM front taps carry it too
    }
    if (is_wfm(x.mode) && x.wfm_dirty) {
      if (x.wfm_video.empty() || x.wfm_resamp.empty()) {
        set_last_error("rx %d is in WFM mode but pysdr_set_wfm_taps was never called", r);
        return PYSDR_ERR_STATE;
      }
      int rc = wfm_setup(c);
      if (rc) return rc;
      if (!x.d_y1) {
        PYSDR_HIP_CHECK(hipMalloc(&x.d_y1, ((size_t)c->m1max + 2) * sizeof(float2)));
        PYSDR_HIP_CHECK(hipMemsetAsync(x.d_y1, 0, ((size_t)c->m1max + 2) * sizeof(float2), c->stream));
        drained = drained || need_alt;        // (the pair's second buffer follows below)
        PYSDR_HIP_CHECK(hipMalloc(&x.d_w, (size_t)c->m1max * sizeof(float2)));
        rc = decim_init(x.wfm_audio, c->up2, c->down2, (int)x.wfm_resamp.size(), 1, c->stream);
        if (rc) return rc;
      }
      rc = decim_upload_taps(c, c->wfm_front, r, x.wfm_video.data(), c->cfg.ntaps_dec, x.fword);
      if (rc) return rc;
      rc = decim_upload_taps(c, x.wfm_audio, 0, x.wfm_resamp.data(), (int)x.wfm_resamp.size(), 0u);
      if (rc) return rc;
      x.wfm_dirty = false;
    }
    if (x.af_dirty) {
      std::vector<float2> t((c->cfg.ntaps_af + 7) & ~7, make_float2(0.f, 0.f));
      x.taps_real = 1;
      for (int k = 0; k < c->cfg.ntaps_af; ++k) {
        t[k] = make_float2((float)x.af[2 * k], (float)x.af[2 * k + 1]);
        if (t[k].y != 0.f) x.taps_real = 0;
      }
      PYSDR_HIP_CHECK(hipMemcpyAsync(x.d_aftaps, t.data(), t.size() * sizeof(float2),
                                     hipMemcpyHostToDevice, c->stream));
      PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));   // t goes out of scope
      x.af_dirty = false;
    }
    if (x.reset_pending) {
      RxDevState st;
      PYSDR_HIP_CHECK(hipMemcpyAsync(&st, c->d_state + r, sizeof(st), hipMemcpyDeviceToHost, c->stream));
      PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (x.reset_pending & 1u) { st.env = 0.f; st.gain = 1.f; st.maxbuf = 0.f; st.err = 0.f; st.sq_level = 0.f; st.sq_open = 1; st.sq_lp = 0.f; st.sq_hp = 0.f; }
      if (x.reset_pending & 2u) { st.pll_phase = 0u; st.pll_w = 0.f; st.wfm_phase = 0u; st.wfm_w = 0.f; st.wfm_slope_ok = 0; }
      st.ref = x.agc_ref; st.agc_enable = x.agc_enable;
      PYSDR_HIP_CHECK(hipMemcpyAsync(c->d_state + r, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
      PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
      x.reset_pending = 0; x.agc_dirty = false;
    }
    if (x.agc_dirty) {
      PYSDR_HIP_CHECK(hipMemcpyAsync(&(c->d_state + r)->ref, &x.agc_ref, sizeof(float), hipMemcpyHostToDevice, c->stream));
      PYSDR_HIP_CHECK(hipMemcpyAsync(&(c->d_state + r)->agc_enable, &x.agc_enable, sizeof(int), hipMemcpyHostToDevice, c->stream));
      PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
      x.agc_dirty = false;
    }
    if (x.mode == PYSDR_WFM2 && x.d_seed == nullptr && c->wfm_seeded && c->m1max > 0) {
      PYSDR_HIP_CHECK(hipMalloc(&x.d_seed, pll_seed_doubles(c->m1max) * sizeof(double)));
      for (int i = 0; i < 2; ++i) {
        PYSDR_HIP_CHECK(hipMalloc(&x.d_mnt[i], pll_seed_mnt_floats(c->m1max) * sizeof(float)));
        PYSDR_HIP_CHECK(hipMemsetAsync(x.d_mnt[i], 0, pll_seed_mnt_floats(c->m1max) * sizeof(float), c->stream));
      }
    }
    if (x.mode == PYSDR_AM_SYNCH && x.d_ypll == nullptr) {
      const size_t n = (size_t)c->hy + c->mmax;
      PYSDR_HIP_CHECK(hipMalloc(&x.d_ypll, n * sizeof(float2)));
      PYSDR_HIP_CHECK(hipMemsetAsync(x.d_ypll, 0, n * sizeof(float2), c->stream));
    }
    // (a second buffer that comes into being while it is the CURRENT one -- a mode first used after overlapped calls left
    //  par at 1 -- inherits the history prefix of the first, which is where that mode's last use left it)
    if (need_alt && x.d_ypll != nullptr && x.d_ypll_alt == nullptr) {
      const size_t ny = (size_t)c->hy + c->mmax;
      PYSDR_HIP_CHECK(hipMalloc(&x.d_ypll_alt, ny * sizeof(float2)));
      PYSDR_HIP_CHECK(hipMemsetAsync(x.d_ypll_alt, 0, ny * sizeof(float2), c->stream));
      if (c->par) PYSDR_HIP_CHECK(hipMemcpyAsync(x.d_ypll_alt, x.d_ypll, (size_t)c->hy * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    }
    if (need_alt && x.d_y_alt == nullptr) {
      const size_t ny = (size_t)c->hy + c->mmax;
      PYSDR_HIP_CHECK(hipMalloc(&x.d_y_alt, ny * sizeof(float2)));
      PYSDR_HIP_CHECK(hipMemsetAsync(x.d_y_alt, 0, ny * sizeof(float2), c->stream));
      if (c->par) PYSDR_HIP_CHECK(hipMemcpyAsync(x.d_y_alt, x.d_y, (size_t)c->hy * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    }
    if (need_alt && x.d_y1 != nullptr && x.d_y1_alt == nullptr) {
      PYSDR_HIP_CHECK(hipMalloc(&x.d_y1_alt, ((size_t)c->m1max + 2) * sizeof(float2)));
      PYSDR_HIP_CHECK(hipMemsetAsync(x.d_y1_alt, 0, ((size_t)c->m1max + 2) * sizeof(float2), c->stream));
      if (c->par) PYSDR_HIP_CHECK(hipMemcpyAsync(x.d_y1_alt, x.d_y1, 2 * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
      PYSDR_HIP_CHECK(hipMalloc(&x.d_w_alt, (size_t)c->m1max * sizeof(float2)));
    }
    RxSnap& q = snap->rx[r];
    q.mode = x.mode; q.fword = x.fword; q.phase = x.phase; q.bfo_fword = x.bfo_fword;
    q.sq_thresh = x.sq_thresh; q.sq_ratio = x.sq_ratio; q.taps_real = x.taps_real;
    q.d_a = x.d_a; q.d_am = x.d_am; q.d_aftaps = x.d_aftaps;
    q.d_seed = x.d_seed;
    q.d_mnt = x.d_mnt[c->par];
    // this call's buffer of each pair and the next call's (the other one when the calls overlap)
    const int p = c->par, pn = snap->use2 ? (p ^ 1) : p;
    q.d_y = p ? x.d_y_alt : x.d_y;
    q.d_y_next = pn ? x.d_y_alt : x.d_y;
    q.d_ypll = p ? x.d_ypll_alt : x.d_ypll;
    q.d_ypll_next = pn ? x.d_ypll_alt : x.d_ypll;
    q.d_y1 = p ? x.d_y1_alt : x.d_y1;
    q.d_y1_next = pn ? x.d_y1_alt : x.d_y1;
    q.d_w = p ? x.d_w_alt : x.d_w;
  }
  if (drained) PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  return PYSDR_OK;
}

// Stage2Args of a call from its snapshot (everything but the carrier-loop plan)
void fill_stage2(pysdr_ctx* c, const CallSnap& snap, bool wfm, int nchunks, size_t chunk_len, const DecimResult& res,
                 Stage2Args* out) {
  Stage2Args& s = *out;
  memset(&s, 0, sizeof(s));
  const int up = c->cfg.up, down = c->cfg.down, nrx = snap.nrx;
  s.nrx = nrx; s.n_out = res.n_out; s.ntaps = c->cfg.ntaps_af; s.hy = c->hy;
  s.t0 = res.t0; s.up = up; s.down = down; s.chunk_len = (uint32_t)chunk_len; s.nchunks = nchunks;
  s.m0_lo = (uint32_t)(res.m0 & 0xFFFFFFFFull);
  const double fs_out = std::floor(c->cfg.srate * up / down);
  s.fm_scale = (float)(fs_out / (2.0 * M_PI * kNfmFullScaleDev));
  {
    const double wn = 2.0 * M_PI * kPllBwHz / fs_out;
    s.pll_kp = (float)(2.0 * kPllZeta * wn);
    s.pll_ki = (float)(wn * wn);
  }
  for (int r = 0; r < nrx; ++r) {
    const RxSnap& x = snap.rx[r];
    s.y[r] = x.d_y + c->hy;
    s.ypll[r] = x.d_ypll ? x.d_ypll + c->hy : nullptr;
    s.aftaps[r] = x.d_aftaps;
    s.taps_real[r] = x.taps_real;
    s.a[r] = x.d_a;
    s.am[r] = x.d_am;
    s.det[r] = mode_detector(x.mode);
    s.out_complex[r] = (x.mode == PYSDR_IQ || wfm) ? 1 : 0;
    s.fir_complex[r] = s.out_complex[r];
    s.single_block[r] = wfm ? 1 : 0;
    s.single_spread = 1;
    while (s.single_spread * 2 <= std::min(nchunks, 32)) s.single_spread *= 2;
    s.matrix[r] = (x.mode == PYSDR_WFM2) ? 1 : 0;
    s.bfo_fword[r] = x.bfo_fword;
    s.sq_ratio[r] = (x.sq_ratio > 0.f && c->d_sqtaps != nullptr) ? 1 : 0;
    s.sq_thresh[r] = s.sq_ratio[r] ? x.sq_ratio : x.sq_thresh;
  }
  s.blkpeak = c->d_blkpeak; s.gain = c->d_gain; s.state = c->d_state;
  s.blknoise = c->d_blknoise; s.blkcnt = c->d_blkcnt;
  s.blknoise2 = c->d_blknoise2; s.sqtaps = c->d_sqtaps; s.sq_ntaps = c->sq_ntaps;
}

// T: the tail of a call on `stream` -- for broadcast FM the audio resamplers first, then detector + AF FIR, the block
// gains (+ the history rolls into the next call's buffers) and the output stage.  Called at the end of the call itself
// (single-stream form) or from the NEXT call / a flush (deferred).
int run_tail(pysdr_ctx* c, TailJob& j) {
  j.valid = false;
  const CallSnap& snap = j.snap;
  const int nrx = snap.nrx;
  if (j.wait_pll) PYSDR_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_pll[j.par], 0));
  Stage2Args& s = j.s;
  int rc;
  if (j.wfm) {
    DecimResult res;
    const uint32_t zero = 0u;
    for (int r = 0; r < nrx; ++r) {
      float2* y1 = snap.rx[r].d_y + c->hy;
      rc = decim_run(c, c->rx[r].wfm_audio, snap.rx[r].d_w, (size_t)j.n1, 1, &y1, &zero, &zero, nullptr, 0,
                     c->mmax, &res, c->stream);
      if (rc) return rc;
    }
    fill_stage2(c, snap, true, j.nchunks, j.chunk_len, res, &s);
  }
  const int n_out = s.n_out;
  rc = launch_demod_fir(s, c->stream); if (rc) return rc;
  // the gains, and beside them (same launch) the history roll of the FS_OUT-rate buffers: the AF FIR was their last reader
  EpilogueArgs e;
  memset(&e, 0, sizeof(e));
  e.nrx = nrx; e.n_out = n_out; e.hy = c->hy;
  for (int r = 0; r < nrx; ++r) {
    e.ybase[r] = snap.rx[r].d_y;
    e.ydst[r] = snap.rx[r].d_y_next;
    e.ypllbase[r] = (s.det[r] == kDetPll) ? snap.rx[r].d_ypll : nullptr;
    e.yplldst[r] = (s.det[r] == kDetPll) ? snap.rx[r].d_ypll_next : nullptr;
  }
  rc = launch_agc_scan(s, e, c->stream); if (rc) return rc;
  // WFM (mono) emits the real part of the complex pipeline
  for (int r = 0; r < nrx; ++r) if (snap.rx[r].mode == PYSDR_WFM) s.out_complex[r] = 0;
  rc = launch_apply(s, c->stream); if (rc) return rc;
  if (j.ev_end) PYSDR_HIP_CHECK(hipEventRecord(j.ev_end, c->stream));
  for (int r = 0; r < nrx; ++r) c->last_complex[r] = (snap.rx[r].mode == PYSDR_IQ || snap.rx[r].mode == PYSDR_WFM2) ? 1 : 0;
  c->last_par = j.par;
  c->last_nrx = nrx;
  c->last_nout = n_out; c->last_nchunks = j.nchunks;
  c->last_chunk_len = j.chunk_len; c->last_s0 = j.s0; c->last_wfm = j.wfm ? 1 : 0;
  return PYSDR_OK;
}

// Whatever asks for a call's results (fetch, sync, the state getters, a setter's pending work, a spectrum that orders
// itself behind the whole demodulation) first queues its deferred tail.
int flush_tail(pysdr_ctx* c) {
  if (!c->tail.valid) return PYSDR_OK;
  return run_tail(c, c->tail);
}

}  // namespace

extern "C" {

const char* pysdr_strerror(int status) {
  switch (status) {
    case PYSDR_OK: return "ok";
    case PYSDR_ERR_ARG: return "bad argument";
    case PYSDR_ERR_NO_DEVICE: return "no HIP device";
    case PYSDR_ERR_HIP: return "HIP runtime error";
    case PYSDR_ERR_FFT: return "rocFFT error";
    case PYSDR_ERR_STATE: return "bad state / capacity exceeded";
    case PYSDR_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
  }
}

const char* pysdr_last_error(void) { return g_err; }

int pysdr_version(void) { return 100; }

int pysdr_device_count(int* n) {
  if (!n) return PYSDR_ERR_ARG;
  int k = 0;
  hipError_t e = hipGetDeviceCount(&k);
  if (e != hipSuccess) {
    *n = 0;
    set_last_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return PYSDR_ERR_NO_DEVICE;
  }
  *n = k;
  return PYSDR_OK;
}

uint32_t pysdr_freq_word(double f_hz, double fs_hz, double* f_actual) {
  const double w = std::nearbyint(f_hz / fs_hz * kTwo32);
  long long wi = (long long)w;
  long long ws = ((wi + (1LL << 31)) % (1LL << 32) + (1LL << 32)) % (1LL << 32) - (1LL << 31);
  if (f_actual) *f_actual = (double)ws * fs_hz / kTwo32;
  return (uint32_t)(ws & 0xFFFFFFFFLL);
}

int pysdr_create(const pysdr_cfg* cfg, pysdr_ctx** out) {
  if (!cfg || !out) return PYSDR_ERR_ARG;
  if (cfg->up < 1 || cfg->down < 1 || cfg->in_chunk < 1 || cfg->max_chunks < 1 ||
      cfg->ntaps_dec < 1 || cfg->ntaps_af < 1 || cfg->ntaps_af > 2048 || cfg->srate <= 0) {
    set_last_error("pysdr_create: invalid cfg");
    return PYSDR_ERR_ARG;
  }
  if (cfg->max_chunks > 16384) {
    // agc_scan_kernel holds two words per AGC block of a call (+ 1/16 padding) in LDS: 160 KB end at ~19 k blocks
    set_last_error("pysdr_create: max_chunks %d > 16384 (the AGC scan keeps a call's blocks in LDS)", cfg->max_chunks);
    return PYSDR_ERR_ARG;
  }
  int rc = use_device(cfg->device);
  if (rc) return rc;
  pysdr_ctx* c = new pysdr_ctx();
  c->cfg = *cfg;
  // FIR history (taps padded to 8) + discriminator, rounded up to 16 outputs so that output 0 of a
  // call sits on a 128-byte line: the mix+decimate kernel's 512-byte wave stores then cover whole
  // lines (partial lines are written through as masked writes = read-modify-write at the DRAM)
  c->hy = ((((cfg->ntaps_af + 7) & ~7) + 4 + 1) + 15) & ~15;
  c->cap_samples = (size_t)cfg->max_chunks * (size_t)cfg->in_chunk;
  // tuning / ablation switches (bench.py and DESIGN.md 4.1 use them; all default to off, read only under PYSDR_TUNING=1)
  { const char* e = tuning_env("PYSDR_MIXDEC_WGS"); if (e && atoi(e) > 0) c->wgs_per_cu = atoi(e); }
  { const char* e = tuning_env("PYSDR_MIXDEC_YFLUSH"); if (e && atoi(e) > 0) c->yflush_cap = atoi(e); }
  { const char* e = tuning_env("PYSDR_AM_PLL");
    if (e && *e) {
      double t = 0, tx = 0; int cs = 0, km = 0, tm = 0;
      const int nf = sscanf(e, "%lf,%lf,%d,%d,%d", &t, &tx, &cs, &km, &tm);
      if (nf >= 1 && t > 0) c->am_taus = t;
      if (nf >= 2 && tx >= 0) c->am_taus_exact = tx;
      if (nf >= 3 && cs >= 0) c->am_coarse_sweeps = std::min(cs, 8);
      if (nf >= 4 && km > 0) c->am_kmax = std::min(km, kPllSegMax);
      if (nf >= 5 && tm >= 64) c->am_tmin = (tm + 63) & ~63;
    } }
  { const char* e = tuning_env("PYSDR_AM_SEED");
    if (e && *e) { int on = 1, ws = 0; const int nf = sscanf(e, "%d,%d", &on, &ws); if (nf >= 1) c->am_seeded = on ? 1 : 0; if (nf >= 2 && ws >= 0) c->am_wseed = ws; } }
  { const char* e = tuning_env("PYSDR_AM_DIRECT"); if (e && *e) c->am_direct = atoi(e) ? 1 : 0; }
  { const char* e = tuning_env("PYSDR_AM_PHASE_STREAM"); if (e && *e) c->am_phase_on_2 = atoi(e) ? 1 : 0; }
  { const char* e = tuning_env("PYSDR_MIXDEC_MFMA"); if (e && *e) c->mfma_enable = atoi(e) ? 1 : 0; }
  { const char* e = tuning_env("PYSDR_WFM_SEED");
    if (e && *e) { int on = 1, ws = 0; const int nf = sscanf(e, "%d,%d", &on, &ws); if (nf >= 1) c->wfm_seeded = on ? 1 : 0; if (nf >= 2 && ws >= 0) c->wfm_wseed = ws; } }
  { const char* e = tuning_env("PYSDR_OVERLAP_ORDER"); if (e && *e) c->tail_first = atoi(e); }
  { const char* e = tuning_env("PYSDR_OVERLAP"); if (e && *e) c->overlap_env = std::max(0, std::min(2, atoi(e))); }
  { const char* e = tuning_env("PYSDR_RESAMP_PLAIN"); if (e && *e) c->resamp_plain = atoi(e); }
  { const char* e = tuning_env("PYSDR_MIXDEC_GRID"); if (e && atoi(e) > 0) c->grid_override = atoi(e); }
  { const char* e = tuning_env("PYSDR_WFM_PLL");
    if (e && *e) {
      double a = c->wfm_taus, b = c->wfm_taus_fast, x = c->wfm_taus_exact;
      int sw = c->wfm_coarse_sweeps, km = c->wfm_kmax, tm = c->wfm_tmin, xc = c->wfm_exact_cap;
      double th = c->wfm_taus_hi, tmid = c->wfm_taus_mid;
      int tc = c->wfm_tail_cap;
      const int got = sscanf(e, "%lf,%lf,%lf,%d,%d,%d,%d,%lf,%lf,%d", &a, &b, &x, &sw, &km, &tm, &xc, &th, &tmid, &tc);
      if (got >= 10 && tc >= 0) c->wfm_tail_cap = tc;
      if (got >= 9 && th >= 0 && tmid >= 0) { c->wfm_taus_hi = th; c->wfm_taus_mid = tmid; }
      if (got >= 7 && xc >= 0) c->wfm_exact_cap = xc;
      if (got >= 1 && a > 0) c->wfm_taus = a;
      if (got >= 2 && b >= 0) c->wfm_taus_fast = b;
      if (got >= 3 && x > 0) c->wfm_taus_exact = x;
      if (got >= 4 && sw >= 0) c->wfm_coarse_sweeps = sw;
      if (got >= 5 && km >= 1) c->wfm_kmax = std::min(km, kPllSegMax);
      if (got >= 6 && tm >= 64) c->wfm_tmin = tm;
    } }
#ifdef PYSDR_DIAG
  // work-skipping ablation switches exist only in a diagnostic build (python -m pysdr_amd.build --diag)
  { const char* e = getenv("PYSDR_DEBUG_FLAGS"); c->dbg_flags = e ? atoi(e) : 0; }
#endif
  c->mmax = (int)((c->cap_samples * (size_t)cfg->up) / (size_t)cfg->down) + 4;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0)
      c->num_cus = prop.multiProcessorCount;
  }
  if ((double)c->cap_samples * cfg->up + cfg->down >= 2147483647.0) {
    // the same bound decim_run enforces per call (31-bit index arithmetic in the kernel)
    set_last_error("pysdr_create: max_chunks*in_chunk*up must stay below 2^31");
    delete c;
    return PYSDR_ERR_ARG;
  }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_error("pysdr_create: %s -> %s", #e, hipGetErrorString(_e)); pysdr_destroy(c); return PYSDR_ERR_HIP; } } while (0)
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  rc = decim_init(c->main, cfg->up, cfg->down, cfg->ntaps_dec, PYSDR_MAX_RX, c->stream);
  if (rc) { pysdr_destroy(c); return rc; }
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&c->d_peak2[i], (size_t)cfg->max_chunks * sizeof(unsigned)));
    CK(hipMemsetAsync(c->d_peak2[i], 0, (size_t)cfg->max_chunks * sizeof(unsigned), c->stream));
  }
  c->d_peak = c->d_peak2[0];
  CK(hipMalloc(&c->d_peak_scratch, 64 * sizeof(unsigned)));
  CK(hipMalloc(&c->d_blkpeak, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(unsigned)));
  CK(hipMemsetAsync(c->d_blkpeak, 0, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(unsigned), c->stream));
  CK(hipMalloc(&c->d_gain, (size_t)PYSDR_MAX_RX * cfg->max_chunks * sizeof(float)));
  CK(hipMalloc(&c->d_blknoise, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(float)));
  CK(hipMemsetAsync(c->d_blknoise, 0, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(float), c->stream));
  CK(hipMalloc(&c->d_blkcnt, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(unsigned)));
  CK(hipMemsetAsync(c->d_blkcnt, 0, (size_t)PYSDR_MAX_RX * cfg->max_chunks * kBlkStride * sizeof(unsigned), c->stream));
  CK(hipMalloc(&c->d_state, PYSDR_MAX_RX * sizeof(RxDevState)));
  CK(hipMemsetAsync(c->d_state, 0, PYSDR_MAX_RX * sizeof(RxDevState), c->stream));
  CK(hipMalloc(&c->d_pllseg, (size_t)PYSDR_MAX_RX * kPllSegMax * 5 * sizeof(uint32_t)));
  for (int k = 0; k < pysdr_ctx::kSlots; ++k)
    for (int i = 0; i < 4; ++i) CK(hipEventCreate(&c->ev[k][i]));
  CK(hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming));
#undef CK
  if (c->overlap_env >= 1) { rc = pysdr_set_overlap(c, c->overlap_env); if (rc) { pysdr_destroy(c); return rc; } }
  *out = c;
  return PYSDR_OK;
}

void pysdr_destroy(pysdr_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  pysdr_comm_destroy(c);
  for (int r = 0; r < PYSDR_MAX_RX; ++r) {
    RxHost& x = c->rx[r];
    if (x.d_y) (void)hipFree(x.d_y);
    if (x.d_y_alt) (void)hipFree(x.d_y_alt);
    if (x.d_y1_alt) (void)hipFree(x.d_y1_alt);
    if (x.d_ypll_alt) (void)hipFree(x.d_ypll_alt);
    if (x.d_w_alt) (void)hipFree(x.d_w_alt);
    if (x.d_seed) (void)hipFree(x.d_seed);
    for (int i = 0; i < 2; ++i) if (x.d_mnt[i]) (void)hipFree(x.d_mnt[i]);
    if (x.d_ypll) (void)hipFree(x.d_ypll);
    if (x.d_a) (void)hipFree(x.d_a);
    if (x.d_am) (void)hipFree(x.d_am);
    if (x.d_aftaps) (void)hipFree(x.d_aftaps);
    if (x.d_y1) (void)hipFree(x.d_y1);
    if (x.d_w) (void)hipFree(x.d_w);
    decim_free(x.wfm_audio);
  }
  decim_free(c->main);
  decim_free(c->wfm_front);
  if (c->d_stage) (void)hipFree(c->d_stage);
  for (int i = 0; i < 2; ++i) if (c->d_peak2[i]) (void)hipFree(c->d_peak2[i]);
  if (c->d_peak_scratch) (void)hipFree(c->d_peak_scratch);
  if (c->d_blkpeak) (void)hipFree(c->d_blkpeak);
  if (c->d_gain) (void)hipFree(c->d_gain);
  if (c->d_blknoise) (void)hipFree(c->d_blknoise);
  if (c->d_blknoise2) (void)hipFree(c->d_blknoise2);
  if (c->d_sqtaps) (void)hipFree(c->d_sqtaps);
  if (c->d_blkcnt) (void)hipFree(c->d_blkcnt);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_pllseg) (void)hipFree(c->d_pllseg);
  for (int k = 0; k < pysdr_ctx::kSlots; ++k)
    for (int i = 0; i < 4; ++i) if (c->ev[k][i]) (void)hipEventDestroy(c->ev[k][i]);
  if (c->ev_front) (void)hipEventDestroy(c->ev_front);
  for (int i = 0; i < 2; ++i) if (c->ev_pll[i]) (void)hipEventDestroy(c->ev_pll[i]);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int pysdr_rx_add(pysdr_ctx* c, int mode, double lo_freq, const double* h, const double* af,
                 double bfo, int* irx) {
  if (!c || !h || !af) return PYSDR_ERR_ARG;
  if (mode < 0 || mode > PYSDR_RTTY) return PYSDR_ERR_ARG;
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->nrx >= PYSDR_MAX_RX) {
    set_last_error("pysdr_rx_add: more than %d receivers", PYSDR_MAX_RX);
    return PYSDR_ERR_STATE;
  }
  const int r = c->nrx;
  RxHost& x = c->rx[r];
  x.mode = mode;
  x.lo_freq = lo_freq;
  x.fword = pysdr_freq_word(lo_freq, c->cfg.srate, nullptr);
  x.phase = 0;
  x.h.assign(h, h + c->cfg.ntaps_dec);
  x.af.assign(af, af + 2 * c->cfg.ntaps_af);
  x.bfo = bfo;
  const double fs_out = c->cfg.srate * c->cfg.up / c->cfg.down;
  x.bfo_fword = pysdr_freq_word(bfo, std::floor(fs_out), nullptr);
  x.agc_enable = mode_has_agc(mode);
  x.taps_dirty = x.af_dirty = x.agc_dirty = true;
  x.reset_pending = 3;
  const size_t ny = (size_t)c->hy + c->mmax;
  PYSDR_HIP_CHECK(hipMalloc(&x.d_y, ny * sizeof(float2)));
  PYSDR_HIP_CHECK(hipMemsetAsync(x.d_y, 0, ny * sizeof(float2), c->stream));
  PYSDR_HIP_CHECK(hipMalloc(&x.d_a, (size_t)c->mmax * sizeof(float2)));
  PYSDR_HIP_CHECK(hipMalloc(&x.d_am, (size_t)c->mmax * 2 * sizeof(float)));
  PYSDR_HIP_CHECK(hipMalloc(&x.d_aftaps, (size_t)((c->cfg.ntaps_af + 7) & ~7) * sizeof(float2)));
  c->nrx = r + 1;
  if (irx) *irx = r;
  return PYSDR_OK;
}

int pysdr_set_lo(pysdr_ctx* c, int irx, double f_hz, double* f_actual) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  RxHost& x = c->rx[irx];
  x.lo_freq = f_hz;
  x.fword = pysdr_freq_word(f_hz, c->cfg.srate, f_actual);
  x.taps_dirty = true;
  return PYSDR_OK;
}

int pysdr_set_dec_taps(pysdr_ctx* c, int irx, const double* h, int n) {
  if (!c || !h || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  if (n != c->cfg.ntaps_dec) {
    set_last_error("pysdr_set_dec_taps: got %d taps, context was created for %d", n, c->cfg.ntaps_dec);
    return PYSDR_ERR_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  c->rx[irx].h.assign(h, h + n);
  c->rx[irx].taps_dirty = true;
  return PYSDR_OK;
}

int pysdr_set_mode(pysdr_ctx* c, int irx, int mode, const double* af, int n, double bfo) {
  if (!c || irx < 0 || irx >= c->nrx || mode < 0 || mode > PYSDR_RTTY) return PYSDR_ERR_ARG;
  if (af && n != c->cfg.ntaps_af) {
    set_last_error("pysdr_set_mode: got %d AF taps, context was created for %d", n, c->cfg.ntaps_af);
    return PYSDR_ERR_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  RxHost& x = c->rx[irx];
  x.mode = mode;
  if (af) { x.af.assign(af, af + 2 * n); x.af_dirty = true; }
  x.bfo = bfo;
  const double fs_out = std::floor(c->cfg.srate * c->cfg.up / c->cfg.down);
  x.bfo_fword = pysdr_freq_word(bfo, fs_out, nullptr);
  x.agc_enable = mode_has_agc(mode);
  x.agc_dirty = true;
  return PYSDR_OK;
}

int pysdr_reset(pysdr_ctx* c, int irx, unsigned what) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  c->rx[irx].reset_pending |= (what & 3u);
  return PYSDR_OK;
}

int pysdr_set_agc(pysdr_ctx* c, int irx, int enable, float ref) {
  if (!c || irx < 0 || irx >= c->nrx || !(ref > 0.f)) return PYSDR_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  c->rx[irx].agc_enable = enable ? 1 : 0;
  c->rx[irx].agc_ref = ref;
  c->rx[irx].agc_dirty = true;
  return PYSDR_OK;
}

int pysdr_set_squelch(pysdr_ctx* c, int irx, float thresh) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  c->rx[irx].sq_thresh = thresh > 0.f ? thresh : 0.f;
  return PYSDR_OK;
}

int pysdr_set_squelch_ratio(pysdr_ctx* c, int irx, float min_ratio, const float* lp, const float* hp, int ntaps) {
  if (!c || irx < 0 || irx >= c->nrx || (min_ratio > 0.f && (!lp || !hp || ntaps < 1 || ntaps > kSqTapsMax))) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);     // allocates and uploads on the context's stream
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  if (min_ratio > 0.f) {
    const size_t nb = (size_t)PYSDR_MAX_RX * c->cfg.max_chunks * kBlkStride * sizeof(float);
    if (!c->d_blknoise2) {
      PYSDR_HIP_CHECK(hipMalloc(&c->d_blknoise2, nb));
      PYSDR_HIP_CHECK(hipMemsetAsync(c->d_blknoise2, 0, nb, c->stream));
    }
    if (!c->d_sqtaps) PYSDR_HIP_CHECK(hipMalloc(&c->d_sqtaps, 2 * kSqTapsMax * sizeof(float)));
    float h[2 * kSqTapsMax] = {0.f};
    for (int q = 0; q < ntaps; ++q) { h[q] = lp[q]; h[kSqTapsMax + q] = hp[q]; }
    // (the two FIRs belong to the context -- they depend on FS_OUT only; pageable source: the copy is staged before the call returns)
    PYSDR_HIP_CHECK(hipMemcpyAsync(c->d_sqtaps, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->sq_ntaps = ntaps;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  c->rx[irx].sq_ratio = min_ratio > 0.f ? min_ratio : 0.f;
  return PYSDR_OK;
}

int pysdr_squelch_ratio_get(pysdr_ctx* c, int irx, float* sq_lp, float* sq_hp, int* open) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (sq_lp) *sq_lp = d.sq_lp;
  if (sq_hp) *sq_hp = d.sq_hp;
  if (open) *open = d.sq_open;
  return PYSDR_OK;
}

int pysdr_squelch_get(pysdr_ctx* c, int irx, float* level, int* open) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);                     // (its wait for the loop walks orders `stream` behind stream2 as well)
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (level) *level = d.sq_level;
  if (open) *open = d.sq_open;
  return PYSDR_OK;
}

int pysdr_pll_stats(pysdr_ctx* c, int irx, int* segments, int* patched) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);                     // (its wait for the loop walks orders `stream` behind stream2 as well)
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (segments) *segments = d.pll_segments;
  if (patched) *patched = d.pll_patched;
  return PYSDR_OK;
}

int pysdr_pll_linear_starts(pysdr_ctx* c, int irx, int* n) {
  if (!c || !n || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  *n = d.pll_linear;
  return PYSDR_OK;
}

int pysdr_pll_join_margin(pysdr_ctx* c, int irx, int* max_words, float* max_dw) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);                     // (its wait for the loop walks orders `stream` behind stream2 as well)
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (max_words) *max_words = d.pll_join_words;
  if (max_dw) *max_dw = d.pll_join_dw;
  return PYSDR_OK;
}

int pysdr_set_pll_segments(pysdr_ctx* c, int max_segments) {
  if (!c || max_segments < 0) return PYSDR_ERR_ARG;
  c->pll_kmax = max_segments;
  return PYSDR_OK;
}

int pysdr_agc_get(pysdr_ctx* c, int irx, pysdr_agc_state* st) {
  if (!c || !st || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  RxDevState d;
  rc = flush_tail(c);                     // (its wait for the loop walks orders `stream` behind stream2 as well)
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpyAsync(&d, c->d_state + irx, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  st->agc = d.env; st->gain = d.gain; st->maxbuf = d.maxbuf; st->ref = d.ref; st->err = d.err;
  return PYSDR_OK;
}

#ifdef PYSDR_DIAG
// diagnostic build only: where every workgroup of the last mixdec_mfma launch ran and at what clock, [1024][24] uint64
extern "C" int pysdr_diag_mfma_stamps(pysdr_ctx* c, unsigned long long* host) {
  if (!c || !host || !c->d_mm_stamps) return PYSDR_ERR_ARG;
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  PYSDR_HIP_CHECK(hipMemcpy(host, c->d_mm_stamps, 1024 * 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return PYSDR_OK;
}
// diagnostic build only: the last launch's mixdec phase stamps, [2][16][24][8] uint64
extern "C" int pysdr_diag_stamps(pysdr_ctx* c, unsigned long long* host) {
  if (!c || !host || !c->d_stamps) return PYSDR_ERR_ARG;
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  PYSDR_HIP_CHECK(hipMemcpy(host, c->d_stamps, 2 * 16 * 24 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return PYSDR_OK;
}
#endif

int pysdr_build_flags_hash(void) {
#ifdef PYSDR_EXTRA_FLAGS_HASH
  return (int)(PYSDR_EXTRA_FLAGS_HASH);
#else
  return 0;
#endif
}

int pysdr_get_tuning(pysdr_ctx* c, int32_t out[8]) {
  if (!c || !out) return PYSDR_ERR_ARG;
#ifdef PYSDR_DIAG
  out[0] = 1;
#else
  out[0] = 0;
#endif
  out[1] = c->dbg_flags; out[2] = c->wgs_per_cu; out[3] = c->yflush_cap;
  out[4] = c->tile_bytes; out[5] = c->threads; out[6] = c->num_cus; out[7] = c->mfma_enable;
  return PYSDR_OK;
}

int pysdr_set_profile(pysdr_ctx* c, int enable) {
  if (!c) return PYSDR_ERR_ARG;
  c->profile = enable;
  return PYSDR_OK;
}

int pysdr_set_tile(pysdr_ctx* c, int tile_bytes, int threads) {
  if (!c || (tile_bytes != 0 && (tile_bytes < 4096 || tile_bytes > 150 * 1024)) || threads < 64 || threads > 1024 ||
      (threads & 63))
    return PYSDR_ERR_ARG;
  c->tile_bytes = tile_bytes;
  c->threads = threads;
  return PYSDR_OK;
}

int pysdr_get_elapsed_ms(pysdr_ctx* c, int which, int back, float* ms) {
  if (!c || !ms || which < 0 || which > 3 || back < 0 || back >= pysdr_ctx::kSlots - (which == 3 ? 1 : 0)) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  if ((unsigned long long)back + (which == 3 ? 1u : 0u) >= c->ncalls) { set_last_error("pysdr_get_elapsed_ms: no such call"); return PYSDR_ERR_STATE; }
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  // the end mark of an overlapped context's LAST call is recorded by its deferred tail: queue that first (otherwise the event
  // is unrecorded, or still holds the record of 64 calls ago: ADVICE r5)
  rc = flush_tail(c);
  if (rc) return rc;
  hipEvent_t* ev = c->ev[(c->ncalls - 1 - back) % pysdr_ctx::kSlots];
  PYSDR_HIP_CHECK(hipEventSynchronize(ev[3]));
  hipEvent_t a = ev[0], b = ev[1];
  if (which == 1) { a = ev[1]; b = ev[3]; }
  if (which == 2) { a = ev[0]; b = ev[3]; }
  if (which == 3) {              // start of the previous call -> start of this one: the period of a step
    a = c->ev[(c->ncalls - 2 - back) % pysdr_ctx::kSlots][0];
    b = ev[0];
  }
  PYSDR_HIP_CHECK(hipEventElapsedTime(ms, a, b));
  return PYSDR_OK;
}

int pysdr_sync(pysdr_ctx* c) {
  if (!c) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  rc = flush_tail(c);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (c->stream2) PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream2));
  return PYSDR_OK;
}

int pysdr_set_overlap(pysdr_ctx* c, int enable) {
  if (!c || enable < 0 || enable > 2) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  if (enable && c->n_ingest > 0) {
    set_last_error("pysdr_set_overlap: the context feeds an ingest ring, which queues its result copies behind each call on one stream");
    return PYSDR_ERR_STATE;
  }
  // a deferred tail queued, both streams drained: the switch happens between two calls, whatever is queued
  rc = flush_tail(c);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (c->stream2) PYSDR_HIP_CHECK(hipStreamSynchronize(c->stream2));
  if (c->overlap_env >= 0 && c->n_ingest == 0) enable = c->overlap_env;
  c->overlap = enable;
  return PYSDR_OK;
}

int pysdr_get_overlap(pysdr_ctx* c) { return c ? c->overlap : 0; }
int pysdr_last_call_overlapped(pysdr_ctx* c) { return (c && c->use2) ? 1 : 0; }

int pysdr_wfm_params(double srate, double fs_out, int* d1, int* up2, int* down2) {
  if (srate <= 0 || fs_out <= 0) return PYSDR_ERR_ARG;
  const int d = wfm_if_decim(srate);
  const long long a = (long long)std::llround(fs_out), b = (long long)std::llround(srate / d);
  const long long g = std::__gcd(a, b);
  if (d1) *d1 = d;
  if (up2) *up2 = (int)(a / g);
  if (down2) *down2 = (int)(b / g);
  return PYSDR_OK;
}

int pysdr_set_wfm_taps(pysdr_ctx* c, int irx, const double* video, int nv, const double* resamp, int nr) {
  if (!c || !video || !resamp || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  int d1 = 0, up2 = 0, down2 = 0;
  pysdr_wfm_params(c->cfg.srate, std::floor(c->cfg.srate * c->cfg.up / c->cfg.down), &d1, &up2, &down2);
  if (nv != c->cfg.ntaps_dec || nr < up2 || nr % up2) {
    set_last_error("pysdr_set_wfm_taps: need %d video taps and a multiple of %d resampler taps (got %d, %d)",
                   c->cfg.ntaps_dec, up2, nv, nr);
    return PYSDR_ERR_ARG;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  RxHost& x = c->rx[irx];
  if (!x.wfm_resamp.empty() && (int)x.wfm_resamp.size() != nr) {
    set_last_error("pysdr_set_wfm_taps: resampler length cannot change (%zu -> %d)", x.wfm_resamp.size(), nr);
    return PYSDR_ERR_ARG;
  }
  x.wfm_video.assign(video, video + nv);
  x.wfm_resamp.assign(resamp, resamp + nr);
  x.wfm_dirty = true;
  return PYSDR_OK;
}

int pysdr_process_batch(pysdr_ctx* c, const void* iq, int nchunks, size_t chunk_len, int on_device) {
  if (!c || !iq || nchunks < 1 || chunk_len < 1) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  if (c->nrx < 1) { set_last_error("pysdr_process_batch: no receivers"); return PYSDR_ERR_STATE; }
  const size_t n = (size_t)nchunks * chunk_len;
  if (nchunks > c->cfg.max_chunks || n > c->cap_samples) {
    set_last_error("pysdr_process_batch: %d chunks x %zu samples exceeds capacity (%d chunks, %zu samples)",
                   nchunks, chunk_len, c->cfg.max_chunks, c->cap_samples);
    return PYSDR_ERR_STATE;
  }
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  const float2* d_x = reinterpret_cast<const float2*>(iq);
  if (!on_device) {
    if (c->stage_cap < n) {
      if (c->d_stage) PYSDR_HIP_CHECK(hipFree(c->d_stage));
      c->d_stage = nullptr; c->stage_cap = 0;
      PYSDR_HIP_CHECK(hipMalloc(&c->d_stage, n * sizeof(float2)));
      c->stage_cap = n;
    }
    PYSDR_HIP_CHECK(hipMemcpyAsync(c->d_stage, iq, n * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    d_x = c->d_stage;
  }
  // everything below uses the snapshot taken under the lock, never c->rx[] fields a setter may
  // change (mode / fword / pointers re-read after the lock was released)
  CallSnap snap;
  rc = apply_pending(c, &snap);
  if (rc) return rc;
  const int nrx = snap.nrx;
  const bool wfm = snap.nwfm > 0;
  // F on `stream`; P on SP = stream2 when this call's tail is deferred (the loop walks then run beside the previous call's
  // tail and the next call's front end), else `stream`; T on `stream`, now or a call later (pysdr_ctx, run_tail)
  const bool use2 = snap.use2;
  hipStream_t SP = use2 ? c->stream2 : c->stream;
  const int par = c->par;

  const unsigned long long s0 = wfm ? c->wfm_front.s_abs : c->main.s_abs;
  c->peak_cur ^= 1;                       // zeroed by the previous call's history roll (both are zero at the start)
  c->d_peak = c->d_peak2[c->peak_cur];
  if (!c->peak_clean[c->peak_cur])        // ... unless that call failed on the way: then by hand
    PYSDR_HIP_CHECK(hipMemsetAsync(c->d_peak, 0, (size_t)c->cfg.max_chunks * sizeof(unsigned), c->stream));
  c->peak_clean[c->peak_cur] = false;

  float2* yptr[PYSDR_MAX_RX];
  uint32_t ph[PYSDR_MAX_RX], fw[PYSDR_MAX_RX];
  for (int r = 0; r < nrx; ++r) { ph[r] = snap.rx[r].phase; fw[r] = snap.rx[r].fword; }

  hipEvent_t* ev = c->ev[c->ncalls % pysdr_ctx::kSlots];
  if (c->profile) PYSDR_HIP_CHECK(hipEventRecord(ev[0], c->stream));
  TailJob job;
  job.snap = snap; job.wfm = wfm; job.par = par; job.nchunks = nchunks; job.chunk_len = chunk_len; job.n = n; job.s0 = s0;
  job.ev_end = c->profile ? ev[3] : nullptr;
  bool any_loop = false;                  // this call has segment walks (P)
  WfmArgs w;
  if (!wfm) {
    DecimResult res;
    for (int r = 0; r < nrx; ++r) yptr[r] = snap.rx[r].d_y + c->hy;
    rc = decim_run(c, c->main, d_x, n, nrx, yptr, ph, fw, c->d_peak, chunk_len, c->mmax, &res, c->stream);
    if (rc) return rc;
    c->wfm_front.s_abs = c->main.s_abs;       // both pipelines count the same input stream
    fill_stage2(c, snap, false, nchunks, chunk_len, res, &job.s);
    Stage2Args& s = job.s;
    for (int r = 0; r < nrx; ++r) any_loop |= (s.det[r] == kDetPll);
    any_loop = any_loop && s.n_out > 0;
    if (any_loop) {
      const double fs_out = std::floor(c->cfg.srate * c->cfg.up / c->cfg.down);
      s.pll = plan_pll(s.n_out, fs_out, kPllBwHz, c->am_taus, 0.0, c->am_tmin,
                       c->pll_kmax > 0 ? std::min(c->pll_kmax, c->am_kmax) : c->am_kmax, c->d_pllseg);
      if (c->am_coarse_sweeps > 0 && s.pll.K > 1) {
        const double tau = fs_out / (kPllZetaPlan * 2.0 * M_PI * kPllBwHz);
        s.pll.Wexact = ((int)std::ceil(c->am_taus_exact * tau) + 63) & ~63;
        s.pll.coarse_sweeps = c->am_coarse_sweeps;
      }
      s.pll.seeded = (c->am_seeded && s.pll.K > 1) ? 1 : 0;
      s.pll.direct = c->am_direct;
      s.pll.Wseed = std::min((std::max(0, c->am_wseed) + 63) & ~63, std::max(0, s.pll.W - 64));
      // arg y of every sample: parallel, no LDS.  Single-stream form: part of F.  Overlapped form: in front of the walks on
      // their stream -- 17 us that the front stream does not wait for (profiles/r05_am_phase_stream.txt)
      if (!(use2 && c->am_phase_on_2)) { rc = launch_am_phase(s, c->stream); if (rc) return rc; }
    }
  } else {
    // SRATE -> fs1 (video filter, all RX in one launch) and the discriminator at fs1 are F; the pilot loop is P; each
    // RX's own fs1 -> FS_OUT resampler opens T
    for (int r = 0; r < nrx; ++r) yptr[r] = snap.rx[r].d_y1 + 2;
    DecimResult r1;
    rc = decim_run(c, c->wfm_front, d_x, n, nrx, yptr, ph, fw, c->d_peak, chunk_len, c->m1max, &r1, c->stream);
    if (rc) return rc;
    c->main.s_abs = c->wfm_front.s_abs;
    const int n1 = r1.n_out;
    job.n1 = n1;
    memset(&w, 0, sizeof(w));
    w.nrx = nrx; w.n1 = n1;
    const double fs1 = c->cfg.srate / c->d1;
    w.scale = (float)(fs1 / (2.0 * M_PI * 75e3));
    {
      const double wn = 2.0 * M_PI * kWfmPllBwHz / fs1;
      w.kp = (float)(2.0 * 0.7071 * wn);
      w.ki = (float)(wn * wn);
      w.norm = (float)(2.0 / 0.1);
      w.rad2word = (float)(kTwo32 / (2.0 * M_PI));
      w.fword0 = pysdr_freq_word(19000.0, fs1, nullptr);
    }
    for (int r = 0; r < nrx; ++r) {
      w.y1[r] = snap.rx[r].d_y1 + 2;
      w.y1base[r] = snap.rx[r].d_y1;
      w.y1dst[r] = snap.rx[r].d_y1_next;
      w.w[r] = snap.rx[r].d_w;
      w.stereo[r] = (snap.rx[r].mode == PYSDR_WFM2) ? 1 : 0;
      w.seed[r] = snap.rx[r].d_seed;
      w.mnT[r] = (snap.rx[r].mode == PYSDR_WFM2) ? snap.rx[r].d_mnt : nullptr;
    }
    w.state = c->d_state;
    // measured (scripts/experiments/pll_warmup.py), words of 2^32 left of a wrong start state: from
    // the call's initial state free-running, 60-270 after 32768 samples = 17.5 tau (one segment in
    // 200 beyond the 512-word tolerance: 20 tau); from the previous call's MEAN increment (the loop
    // follows a crystal, so its phase is a straight line plus a bounded wobble) 54 after 13 tau
    w.pll = plan_pll(n1, fs1, kWfmPllBwHz, c->wfm_taus, c->wfm_taus_fast, c->wfm_tmin,
                     c->pll_kmax > 0 ? std::min(c->pll_kmax, c->wfm_kmax) : c->wfm_kmax, c->d_pllseg);
    w.pll.exact_cap = c->wfm_exact_cap;
    w.pll.tail_cap = c->wfm_tail_cap;
    w.pll.seeded = (c->wfm_seeded && w.pll.K > 1) ? 1 : 0;
    w.pll.Wseed = (std::max(0, c->wfm_wseed) + 63) & ~63;
    if (c->wfm_coarse_sweeps > 0 && w.pll.K > 1) {
      const double tau = fs1 / (kPllZetaPlan * 2.0 * M_PI * kWfmPllBwHz);
      w.pll.Wexact = ((int)std::ceil(c->wfm_taus_exact * tau) + 63) & ~63;
      w.pll.coarse_sweeps = c->wfm_coarse_sweeps;
      if (c->wfm_taus_hi > 0 || c->wfm_taus_mid > 0) {
        w.pll.Wc_hi = ((int)std::ceil(c->wfm_taus_hi * tau) + 63) & ~63;
        w.pll.Wc_mid = ((int)std::ceil(c->wfm_taus_mid * tau) + 63) & ~63;
      }
    }
    rc = launch_wfm_disc(w, c->stream);
    if (rc) return rc;
    any_loop = wfm_any_stereo(w);
  }
  // "the input of this call has been consumed": a spectrum that orders itself behind the front end waits for it
  // (pysdr_spectrum_order, direction 2), and so do the loop walks of an overlapped call.  An event record costs ~5.5 us
  // of the stream's timeline on this runtime (the kernel behind it starts that much later: scripts/diag/timeline.sh), so
  // ONE event serves the profile and both orderings, and none is recorded when nobody asked for any.
  // (broadcast FM, single-stream form: the profile's mark sits behind the pilot loop as it always has -- "front" there
  //  is the whole FM front end)
  const bool loop_on_2 = use2 && any_loop;
  c->front_marker = nullptr;
  // WHERE the previous call's tail goes: behind the start of this call's walks (P(k) beside T(k-1) and F(k+1)) or in FRONT
  // of the mark they wait for (P(k) starts behind T(k-1) and shares the machine with F(k+1) only).  Measured, one box,
  // GS/s (profiles/r05_overlap_order_ab.txt): broadcast FM single-stream 377, behind 372-382, in front 432-433 -- the AF FIR
  // and the audio resampler ran 3.5-5x longer beside the pilot loop's waves (58 -> 320 us, 36 -> 125: the two share the vector
  // pipe, the walks' DPP scans and v_cos hold it), which put the tail on the critical path; AM-Synch single-stream 370,
  // behind 426, in front 415-416 (its tail is 85 us against walks of 100).  So: in front for broadcast FM, behind otherwise.
  if (loop_on_2 && (c->tail_first >= 0 ? c->tail_first != 0 : wfm)) { rc = flush_tail(c); if (rc) return rc; }
  if (wfm && !loop_on_2 && any_loop) { rc = launch_wfm_pll(w, c->stream); if (rc) return rc; }
  if (c->profile) { PYSDR_HIP_CHECK(hipEventRecord(ev[1], c->stream)); c->front_marker = ev[1]; }
  else if (c->front_wanted || loop_on_2) { PYSDR_HIP_CHECK(hipEventRecord(c->ev_front, c->stream)); c->front_marker = c->ev_front; }
  // P
  if (loop_on_2) {
    PYSDR_HIP_CHECK(hipStreamWaitEvent(SP, c->front_marker, 0));
    if (!wfm && c->am_phase_on_2) { rc = launch_am_phase(job.s, SP); if (rc) return rc; }
    rc = wfm ? launch_wfm_pll(w, SP) : launch_pll(job.s, SP);
    if (rc) return rc;
    PYSDR_HIP_CHECK(hipEventRecord(c->ev_pll[par], SP));
    job.wait_pll = true;
  } else if (!wfm && any_loop) {
    rc = launch_pll(job.s, c->stream);
    if (rc) return rc;
  }
  c->ncalls++;
  {
    // the NCO phase belongs to the process thread; the lock only orders it against pysdr_rx_add
    std::lock_guard<std::mutex> lk(c->mu);
    for (int r = 0; r < nrx; ++r) c->rx[r].phase = snap.rx[r].phase + snap.rx[r].fword * (uint32_t)n;
  }
  // T: the previous call's if it was deferred, then this call's -- now, or left for the next call / a flush
  rc = flush_tail(c);
  if (rc) return rc;
  if (use2) {
    c->tail = job;
    c->tail.valid = true;
    c->par = par ^ 1;
    return PYSDR_OK;
  }
  return run_tail(c, job);
}

// outputs of the last call per chunk: those whose newest input sample falls into chunk k (a
// two-level cascade for WFM); host arithmetic only
static void last_chunk_counts(pysdr_ctx* c, int* chunk_nout) {
  auto first_out = [&](unsigned long long s) -> unsigned long long {
    if (!c->last_wfm) {
      const unsigned long long up = c->cfg.up, down = c->cfg.down;
      return (s * up + down - 1) / down;
    }
    const unsigned long long m1 = (s + c->d1 - 1) / (unsigned long long)c->d1;
    return (m1 * (unsigned long long)c->up2 + c->down2 - 1) / (unsigned long long)c->down2;
  };
  for (int k = 0; k < c->last_nchunks; ++k) {
    const unsigned long long a0 = c->last_s0 + (unsigned long long)k * c->last_chunk_len;
    chunk_nout[k] = (int)(first_out(a0 + c->last_chunk_len) - first_out(a0));
  }
}

int pysdr_fetch(pysdr_ctx* c, int irx, float* am, float* iq, int cap, int* n_out,
                int* am_is_complex, int* chunk_nout, float* peaks) {
  if (!c || irx < 0 || irx >= c->nrx) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  rc = flush_tail(c);                     // a deferred tail (pysdr_set_overlap) is queued now: its results are what is asked for
  if (rc) return rc;
  const int n = c->last_nout;
  if ((am || iq) && cap < n) { set_last_error("pysdr_fetch: cap %d < n_out %d", cap, n); return PYSDR_ERR_ARG; }
  const int cx = c->last_complex[irx];
  // The epilogue rolled the last hy outputs into the prefix but left [hy, hy+n) intact.
  hipStream_t S2 = c->stream;
  const float2* yb = c->last_par ? c->rx[irx].d_y_alt : c->rx[irx].d_y;
  if (am && n > 0)
    PYSDR_HIP_CHECK(hipMemcpyAsync(am, c->rx[irx].d_am, (size_t)n * (cx ? 2 : 1) * sizeof(float), hipMemcpyDeviceToHost, S2));
  if (iq && n > 0)
    PYSDR_HIP_CHECK(hipMemcpyAsync(iq, yb + c->hy, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, S2));
  if (peaks && c->last_nchunks > 0)
    PYSDR_HIP_CHECK(hipMemcpyAsync(peaks, c->d_peak, (size_t)c->last_nchunks * sizeof(float), hipMemcpyDeviceToHost, S2));
  PYSDR_HIP_CHECK(hipStreamSynchronize(S2));
  if (n_out) *n_out = n;
  if (am_is_complex) *am_is_complex = cx;
  if (chunk_nout) last_chunk_counts(c, chunk_nout);
  return PYSDR_OK;
}

int pysdr_process(pysdr_ctx* c, const float* iq, size_t n, pysdr_out* outs) {
  if (!c || !iq || !outs || n < 1) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = pysdr_process_batch(c, iq, 1, n, 0);
  if (rc) return rc;
  rc = flush_tail(c);                     // (an overlapped context deferred the call's tail: its results are wanted now)
  if (rc) return rc;
  for (int r = 0; r < c->last_nrx; ++r) {
    int nout = 0, cx = 0;
    float pk = 0.f;
    rc = pysdr_fetch(c, r, outs[r].am, outs[r].iq, outs[r].cap, &nout, &cx, nullptr, &pk);
    if (rc) return rc;
    outs[r].n_out = nout;
    outs[r].am_is_complex = cx;
    outs[r].peak_in = pk;
  }
  return PYSDR_OK;
}

// ---------------------------------------------------------------- quad mixer
int pysdr_quad_mixer(int device, const float* x, float* y, size_t n, uint32_t phase0,
                     uint32_t fword, uint32_t* phase_out) {
  if (!x || !y) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  if (phase_out) *phase_out = phase0 + fword * (uint32_t)n;
  if (n == 0) return PYSDR_OK;
  float2 *dx = nullptr, *dy = nullptr;
  PYSDR_HIP_CHECK(hipMalloc(&dx, n * sizeof(float2)));
  hipError_t e = hipMalloc(&dy, n * sizeof(float2));
  if (e != hipSuccess) { (void)hipFree(dx); set_last_error("hipMalloc: %s", hipGetErrorString(e)); return PYSDR_ERR_HIP; }
  rc = PYSDR_OK;
  e = hipMemcpy(dx, x, n * sizeof(float2), hipMemcpyHostToDevice);
  if (e == hipSuccess) rc = launch_quad_mixer(dx, dy, n, phase0, fword, nullptr);
  if (e == hipSuccess && rc == PYSDR_OK) e = hipMemcpy(y, dy, n * sizeof(float2), hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  if (e != hipSuccess) { set_last_error("quad_mixer: %s", hipGetErrorString(e)); return PYSDR_ERR_HIP; }
  return rc;
}

// ---------------------------------------------------------------- convolver
int pysdr_fir_real(int device, const float* xx, const float* h, int ntaps, float* y, size_t n) {
  if (!xx || !h || !y || ntaps < 1) return PYSDR_ERR_ARG;
  if (n == 0) return PYSDR_OK;
  int rc = use_device(device);
  if (rc) return rc;
  const size_t nin = n + (size_t)ntaps - 1;
  float *dx = nullptr, *dh = nullptr, *dy = nullptr;
  hipError_t e = hipMalloc(&dx, nin * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&dh, (size_t)ntaps * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&dy, n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(dx, xx, nin * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dh, h, (size_t)ntaps * sizeof(float), hipMemcpyHostToDevice);
  rc = PYSDR_OK;
  if (e == hipSuccess) rc = launch_fir_real(dx, dh, ntaps, dy, (int)n, nullptr);
  if (e == hipSuccess && rc == PYSDR_OK) e = hipMemcpy(y, dy, n * sizeof(float), hipMemcpyDeviceToHost);
  if (dx) (void)hipFree(dx);
  if (dh) (void)hipFree(dh);
  if (dy) (void)hipFree(dy);
  if (e != hipSuccess) { set_last_error("pysdr_fir_real: %s", hipGetErrorString(e)); return PYSDR_ERR_HIP; }
  return rc;
}

// ---------------------------------------------------------------- spectrum
static int get_plan(pysdr_spectrum* sp, int batch, rocfft_plan* out) {
  auto it = sp->plans.find(batch);
  if (it != sp->plans.end()) { *out = it->second; return PYSDR_OK; }
  rocfft_plan plan = nullptr;
  size_t len[1] = {(size_t)sp->nfft};
  rocfft_status s = rocfft_plan_create(&plan, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                       rocfft_precision_single, 1, len, (size_t)batch, nullptr);
  if (s != rocfft_status_success) { set_last_error("rocfft_plan_create(nfft=%d,batch=%d) = %d", sp->nfft, batch, (int)s); return PYSDR_ERR_FFT; }
  size_t wbs = 0;
  rocfft_plan_get_work_buffer_size(plan, &wbs);
  if (wbs > sp->fftwork_bytes) {
    if (sp->d_fftwork) (void)hipFree(sp->d_fftwork);
    sp->d_fftwork = nullptr; sp->fftwork_bytes = 0;
    PYSDR_HIP_CHECK(hipMalloc(&sp->d_fftwork, wbs));
    sp->fftwork_bytes = wbs;
  }
  if (sp->fftwork_bytes) rocfft_execution_info_set_work_buffer(sp->info, sp->d_fftwork, sp->fftwork_bytes);
  sp->plans[batch] = plan;
  *out = plan;
  return PYSDR_OK;
}

int pysdr_spectrum_create(int device, int chunk_size, int nfft, int max_frames, const float* window,
                          pysdr_spectrum** out) {
  if (!out || !window || chunk_size < 1 || nfft < chunk_size || max_frames < 1) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lk(g_rocfft_mu);
    if (g_rocfft_users++ == 0) rocfft_setup();
  }
  pysdr_spectrum* sp = new pysdr_spectrum();
  sp->device = device; sp->chunk = chunk_size; sp->nfft = nfft; sp->max_frames = max_frames;
  sp->force_rocfft = tuning_env("PYSDR_PSD_ROCFFT") != nullptr;
  { const char* e = tuning_env("PYSDR_PSD_GROUP"); if (e && atoi(e) > 0) sp->group = atoi(e); }
  { const char* e = tuning_env("PYSDR_PSD_PACKED"); if (e && *e) sp->packed = atoi(e) ? 1 : 0; }
  { const char* e = tuning_env("PYSDR_PSD_STREAMS"); if (e && atoi(e) >= 1 && atoi(e) <= pysdr_spectrum::kMaxStreams) sp->nstreams = atoi(e); }
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_error("pysdr_spectrum_create: %s -> %s", #e, hipGetErrorString(_e)); pysdr_spectrum_destroy(sp); return PYSDR_ERR_HIP; } } while (0)
  CK(hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking));
  CK(hipMalloc(&sp->d_win, (size_t)chunk_size * sizeof(float)));
  CK(hipMemcpy(sp->d_win, window, (size_t)chunk_size * sizeof(float), hipMemcpyHostToDevice));
  CK(hipMalloc(&sp->d_in, (size_t)chunk_size * sizeof(float2)));
  CK(hipMalloc(&sp->d_out, (size_t)nfft * sizeof(float)));
  CK(hipEventCreate(&sp->ev[0]));
  CK(hipEventCreate(&sp->ev[1]));
#undef CK
  if (rocfft_execution_info_create(&sp->info) != rocfft_status_success ||
      rocfft_execution_info_set_stream(sp->info, sp->stream) != rocfft_status_success) {
    set_last_error("rocfft_execution_info_create failed");
    pysdr_spectrum_destroy(sp);
    return PYSDR_ERR_FFT;
  }
  *out = sp;
  return PYSDR_OK;
}

void pysdr_spectrum_destroy(pysdr_spectrum* sp) {
  if (!sp) return;
  (void)hipSetDevice(sp->device);
  if (sp->stream) (void)hipStreamSynchronize(sp->stream);
  for (auto& kv : sp->plans) rocfft_plan_destroy(kv.second);
  if (sp->info) rocfft_execution_info_destroy(sp->info);
  if (sp->d_win) (void)hipFree(sp->d_win);
  if (sp->d_work) (void)hipFree(sp->d_work);
  for (int i = 1; i < pysdr_spectrum::kMaxStreams; ++i) {
    if (sp->xstream[i]) { (void)hipStreamSynchronize(sp->xstream[i]); (void)hipStreamDestroy(sp->xstream[i]); }
    if (sp->xwork[i]) (void)hipFree(sp->xwork[i]);
    if (sp->ev_join[i]) (void)hipEventDestroy(sp->ev_join[i]);
  }
  if (sp->ev_fork) (void)hipEventDestroy(sp->ev_fork);
  if (sp->d_in) (void)hipFree(sp->d_in);
  if (sp->d_out) (void)hipFree(sp->d_out);
  if (sp->d_fftwork) (void)hipFree(sp->d_fftwork);
  for (int i = 0; i < 2; ++i) if (sp->ev[i]) (void)hipEventDestroy(sp->ev[i]);
  if (sp->ev_order) (void)hipEventDestroy(sp->ev_order);
  if (sp->stream) (void)hipStreamDestroy(sp->stream);
  delete sp;
  std::lock_guard<std::mutex> lk(g_rocfft_mu);
  if (--g_rocfft_users == 0) rocfft_cleanup();
}

// the transform's work area: `frames` frames of nfft complex (the 64k path only ever needs one
// group of frames, the rocFFT path the whole batch)
static int ensure_work(pysdr_spectrum* sp, size_t frames) {
  if (sp->work_frames >= frames) return PYSDR_OK;
  PYSDR_HIP_CHECK(hipStreamSynchronize(sp->stream));
  if (sp->d_work) PYSDR_HIP_CHECK(hipFree(sp->d_work));
  sp->d_work = nullptr; sp->work_frames = 0;
  PYSDR_HIP_CHECK(hipMalloc(&sp->d_work, frames * (size_t)sp->nfft * sizeof(float2)));
  sp->work_frames = frames;
  return PYSDR_OK;
}

static int spectrum_run(pysdr_spectrum* sp, const float2* d_x, size_t hop, int nframes, int is_complex,
                        int db, float* d_out) {
  int rc;
  if (is_complex && sp->chunk == 32768 && sp->nfft == 65536 && !sp->force_rocfft) {
    // the RF-waterfall size: fused four-step transform, in groups of 448 frames: the 224 MB
    // of intermediate of one group then stays in the 256 MB Infinity Cache between the two
    // kernels (the input and the PSD stream past it with non-temporal accesses).  Measured per
    // 10666 frames: 3.45 ms for groups of 4096, 3.02 ms at 320 and 2.96 ms at 448-512 with the
    // streaming hints, 3.6 ms at 576 (no longer fits); below 128 frames launch gaps dominate.
    // Running the rows of group g beside the columns of group g+1 on a second stream with two FULL groups
    // of intermediate is slower (3.9 ms, round 1: the two working sets evict each other); with HALF a group per
    // stream it is 5 % faster (round 3, below).
    // (round 4: the 24-bit intermediate is 384 KB per frame and the mix + decimate kernel beside it now copies
    //  nontemporal: 320 / 384 / 448 / 512 / 576 / 640 frames = C3 step 3.04 / 2.97 / 2.96-2.97 / 2.93 / 2.94 / 3.26 ms,
    //  scripts/diag/psd_group_sweep.sh; on a second box 448 / 512 / 576 = 2.99-3.01 / 2.95-2.99 / 3.05: the cliff moves from box
    //  to box, so the default sits between the old 448 and the best 512)
    int group = sp->group > 0 ? sp->group : (sp->packed ? 480 : 448);
    if (sp->nstreams > 1 && nframes > group) {
      // The groups are dealt out over `nstreams` streams, each with its own intermediate of group / nstreams
      // frames (together the same Infinity Cache footprint as one group): the columns of one sub-group run
      // beside the rows of another, and the kernel boundaries of one stream hide behind the other's kernels.
      // Measured on C3 (PSD ms per 10666 frames): 1 stream 2.77-2.78, 2 streams x 224 frames 2.63-2.67.
      // (Round 1 tried two streams with FULL 448-frame groups each: 448 MB of intermediate, slower.)
      const int ns = sp->nstreams;
      const int part = (group + ns - 1) / ns;
      rc = ensure_work(sp, (size_t)part);
      if (rc) return rc;
      if (sp->xwork_frames < (size_t)part) {
        for (int i = 1; i < ns; ++i) {
          if (!sp->xstream[i]) {
            PYSDR_HIP_CHECK(hipStreamCreateWithFlags(&sp->xstream[i], hipStreamNonBlocking));
            PYSDR_HIP_CHECK(hipEventCreateWithFlags(&sp->ev_join[i], hipEventDisableTiming));
          }
          PYSDR_HIP_CHECK(hipStreamSynchronize(sp->xstream[i]));
          if (sp->xwork[i]) PYSDR_HIP_CHECK(hipFree(sp->xwork[i]));
          sp->xwork[i] = nullptr;
          PYSDR_HIP_CHECK(hipMalloc(&sp->xwork[i], (size_t)part * sp->nfft * sizeof(float2)));
        }
        if (!sp->ev_fork) PYSDR_HIP_CHECK(hipEventCreateWithFlags(&sp->ev_fork, hipEventDisableTiming));
        sp->xwork_frames = (size_t)part;
      }
      PYSDR_HIP_CHECK(hipEventRecord(sp->ev[0], sp->stream));
      // the side streams start behind whatever sp->stream waited for: they wait for the call's start event itself (a
      // second event for the fork was 5.5 us more on the stream)
      for (int i = 1; i < ns; ++i) PYSDR_HIP_CHECK(hipStreamWaitEvent(sp->xstream[i], sp->ev[0], 0));
      int k = 0;
      for (int f0 = 0; f0 < nframes; f0 += part, ++k) {
        const int nf = (nframes - f0 < part) ? nframes - f0 : part;
        const int w = k % ns;
        rc = launch_psd64k(d_x + (size_t)f0 * hop, hop, nf, sp->d_win, w ? sp->xwork[w] : sp->d_work,
                           d_out + (size_t)f0 * sp->nfft, db, w ? sp->xstream[w] : sp->stream, sp->packed);
        if (rc) break;             // a failed launch still joins the side streams below: what was forked keeps writing
      }                            // d_out / xwork until it is done, and the caller reacts to the error right away
      for (int i = 1; i < ns; ++i) {
        PYSDR_HIP_CHECK(hipEventRecord(sp->ev_join[i], sp->xstream[i]));
        PYSDR_HIP_CHECK(hipStreamWaitEvent(sp->stream, sp->ev_join[i], 0));
      }
      PYSDR_HIP_CHECK(hipEventRecord(sp->ev[1], sp->stream));
      return rc;
    }
    rc = ensure_work(sp, (size_t)std::min(group, nframes));
    if (rc) return rc;
    PYSDR_HIP_CHECK(hipEventRecord(sp->ev[0], sp->stream));
    for (int f0 = 0; f0 < nframes; f0 += group) {
      const int nf = (nframes - f0 < group) ? nframes - f0 : group;
      rc = launch_psd64k(d_x + (size_t)f0 * hop, hop, nf, sp->d_win, sp->d_work,
                         d_out + (size_t)f0 * sp->nfft, db, sp->stream, sp->packed);
      if (rc) return rc;
    }
    PYSDR_HIP_CHECK(hipEventRecord(sp->ev[1], sp->stream));
    return PYSDR_OK;
  }
  rocfft_plan plan;
  rc = get_plan(sp, nframes, &plan);
  if (rc) return rc;
  rc = ensure_work(sp, (size_t)nframes);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventRecord(sp->ev[0], sp->stream));
  rc = launch_psd_pre(d_x, hop, nframes, sp->chunk, sp->nfft, sp->d_win, sp->d_work, is_complex, sp->stream);
  if (rc) return rc;
  void* bufs[1] = {sp->d_work};
  rocfft_status s = rocfft_execute(plan, bufs, nullptr, sp->info);
  if (s != rocfft_status_success) { set_last_error("rocfft_execute = %d", (int)s); return PYSDR_ERR_FFT; }
  rc = launch_psd_post(sp->d_work, nframes, sp->nfft, is_complex ? 0 : 1, db, d_out, sp->stream);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventRecord(sp->ev[1], sp->stream));
  return PYSDR_OK;
}

int pysdr_spectrum_frame(pysdr_spectrum* sp, const float* x, int is_complex, int db, float* psd_out,
                         int* n_out) {
  if (!sp || !x || !psd_out) return PYSDR_ERR_ARG;
  int rc = use_device(sp->device);
  if (rc) return rc;
  const size_t bytes = (size_t)sp->chunk * (is_complex ? sizeof(float2) : sizeof(float));
  PYSDR_HIP_CHECK(hipMemcpyAsync(sp->d_in, x, bytes, hipMemcpyHostToDevice, sp->stream));
  rc = spectrum_run(sp, sp->d_in, 0, 1, is_complex, db, sp->d_out);
  if (!rc) sp->ran = true;
  if (rc) return rc;
  const int nout = is_complex ? sp->nfft : sp->nfft / 2;
  PYSDR_HIP_CHECK(hipMemcpyAsync(psd_out, sp->d_out, (size_t)nout * sizeof(float), hipMemcpyDeviceToHost, sp->stream));
  PYSDR_HIP_CHECK(hipStreamSynchronize(sp->stream));
  if (n_out) *n_out = nout;
  return PYSDR_OK;
}

int pysdr_spectrum_batch(pysdr_spectrum* sp, const void* d_iq, int nframes, size_t hop, void* d_out) {
  if (!sp || !d_iq || !d_out || nframes < 1 || nframes > sp->max_frames) return PYSDR_ERR_ARG;
  int rc = use_device(sp->device);
  if (rc) return rc;
  rc = spectrum_run(sp, reinterpret_cast<const float2*>(d_iq), hop, nframes, 1, 1, reinterpret_cast<float*>(d_out));
  if (!rc) sp->ran = true;
  return rc;
}

int pysdr_spectrum_get_tuning(pysdr_spectrum* sp, int32_t out[4]) {
  if (!sp || !out) return PYSDR_ERR_ARG;
  out[0] = sp->group > 0 ? sp->group : (sp->packed ? 480 : 448); out[1] = sp->force_rocfft ? 1 : 0; out[2] = sp->nstreams; out[3] = sp->packed;
  return PYSDR_OK;
}

int pysdr_spectrum_sync(pysdr_spectrum* sp) {
  if (!sp) return PYSDR_ERR_ARG;
  int rc = use_device(sp->device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipStreamSynchronize(sp->stream));
  return PYSDR_OK;
}

int pysdr_spectrum_elapsed_ms(pysdr_spectrum* sp, float* ms) {
  if (!sp || !ms) return PYSDR_ERR_ARG;
  int rc = use_device(sp->device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventSynchronize(sp->ev[1]));
  PYSDR_HIP_CHECK(hipEventElapsedTime(ms, sp->ev[0], sp->ev[1]));
  return PYSDR_OK;
}

int pysdr_spectrum_order(pysdr_spectrum* sp, pysdr_ctx* c, int direction) {
  if (!sp || !c || direction < 0 || direction > 2 || sp->device != c->cfg.device) return PYSDR_ERR_ARG;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(sp->device);
  if (rc) return rc;
  if (direction == 2) {
    c->front_wanted = 1;                  // from the next call on the front end's end is marked
    if (c->front_marker) {
      PYSDR_HIP_CHECK(hipStreamWaitEvent(sp->stream, c->front_marker, 0));
      return PYSDR_OK;
    }
    direction = 0;                        // this call was not marked: behind everything the context has queued (stricter)
  }
  if (direction == 1 && sp->ran) {        // the context behind the spectrum's last call: its end event is already there
    PYSDR_HIP_CHECK(hipStreamWaitEvent(c->stream, sp->ev[1], 0));
    return PYSDR_OK;
  }
  if (!sp->ev_order) PYSDR_HIP_CHECK(hipEventCreateWithFlags(&sp->ev_order, hipEventDisableTiming));
  if (direction == 0) { rc = flush_tail(c); if (rc) return rc; }   // "behind the whole demodulation" includes a deferred tail
  hipStream_t first = direction == 0 ? c->stream : sp->stream;
  hipStream_t then = direction == 0 ? sp->stream : c->stream;
  PYSDR_HIP_CHECK(hipEventRecord(sp->ev_order, first));
  PYSDR_HIP_CHECK(hipStreamWaitEvent(then, sp->ev_order, 0));
  return PYSDR_OK;
}

// ---------------------------------------------------------------- device memory
int pysdr_dev_alloc(int device, size_t bytes, void** out) {
  if (!out || bytes == 0) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMalloc(out, bytes));
  return PYSDR_OK;
}
int pysdr_dev_free(int device, void* p) {
  if (!p) return PYSDR_OK;
  int rc = use_device(device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipFree(p));
  return PYSDR_OK;
}
int pysdr_dev_upload(int device, void* dst, const void* src_host, size_t bytes) {
  if (!dst || !src_host) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
  return PYSDR_OK;
}
int pysdr_dev_download(int device, void* dst_host, const void* src, size_t bytes) {
  if (!dst_host || !src) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost));
  return PYSDR_OK;
}
int pysdr_dev_copy(int device, void* dst, const void* src, size_t bytes) {
  if (!dst || !src) return PYSDR_ERR_ARG;
  int rc = use_device(device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  return PYSDR_OK;
}

// ---------------------------------------------------------------- ingest ring (N4)
void pysdr_ingest_destroy(pysdr_ingest* g) {
  if (!g) return;
  if (g->c) (void)hipSetDevice(g->c->cfg.device);
  if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
  if (g->c && g->c->stream) (void)hipStreamSynchronize(g->c->stream);
  if (g->c && g->counted) { g->c->n_ingest--; g->counted = false; }
  for (auto p : g->h_in) if (p) (void)hipHostFree(p);
  for (auto p : g->h_am) if (p) (void)hipHostFree(p);
  for (auto p : g->h_iq) if (p) (void)hipHostFree(p);
  for (auto p : g->h_peak) if (p) (void)hipHostFree(p);
  for (int i = 0; i < 2; ++i) {
    if (g->d_in[i]) (void)hipFree(g->d_in[i]);
    if (g->ev_free[i]) (void)hipEventDestroy(g->ev_free[i]);
  }
  for (auto e : g->ev_copied) if (e) (void)hipEventDestroy(e);
  for (auto e : g->ev_done) if (e) (void)hipEventDestroy(e);
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  delete g;
}

int pysdr_ingest_create(pysdr_ctx* c, int nslots, pysdr_ingest** out) {
  return pysdr_ingest_create_batched(c, nslots, 1, out);
}

int pysdr_ingest_create_batched(pysdr_ctx* c, int nslots, int chunks_per_slot, pysdr_ingest** out) {
  if (!c || !out || nslots < 2 || nslots > 64 || chunks_per_slot < 1) return PYSDR_ERR_ARG;
  if (chunks_per_slot > c->cfg.max_chunks) {
    set_last_error("pysdr_ingest_create_batched: %d chunks per slot > the context's max_chunks %d", chunks_per_slot,
                   c->cfg.max_chunks);
    return PYSDR_ERR_ARG;
  }
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  // A ring queues the result copies of a slot behind its call and the next slot's call behind those, all on ONE stream:
  // the context runs single-stream from here on (the ring's rate is the PCIe link's, two orders below the kernels')
  c->n_ingest++;
  if (c->overlap) { rc = pysdr_set_overlap(c, 0); if (rc) { c->n_ingest--; return rc; } }
  pysdr_ingest* g = new pysdr_ingest();
  g->c = c; g->nslots = nslots; g->chunks_per_slot = chunks_per_slot;
  g->counted = true;
  g->cap = (size_t)c->cfg.in_chunk * (size_t)chunks_per_slot;
  g->ocap = (int)((g->cap * (size_t)c->cfg.up) / (size_t)c->cfg.down) + 8;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_last_error("pysdr_ingest_create: %s -> %s", #e, hipGetErrorString(_e)); pysdr_ingest_destroy(g); return PYSDR_ERR_HIP; } } while (0)
  CK(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
  g->h_in.assign(nslots, nullptr);
  g->ev_copied.assign(nslots, nullptr);
  g->ev_done.assign(nslots, nullptr);
  g->h_am.assign((size_t)nslots * PYSDR_MAX_RX, nullptr);
  g->h_iq.assign((size_t)nslots * PYSDR_MAX_RX, nullptr);
  g->h_peak.assign(nslots, nullptr);
  g->n_out.assign(nslots, 0);
  g->in_flight.assign(nslots, 0);
  g->n_chunks.assign(nslots, 0);
  g->nrx.assign(nslots, 0);
  g->chunk_nout.assign(nslots, std::vector<int>());
  g->cx.assign((size_t)nslots * PYSDR_MAX_RX, 0);
  for (int s = 0; s < nslots; ++s) {
    CK(hipHostMalloc(reinterpret_cast<void**>(&g->h_in[s]), g->cap * sizeof(float2), hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void**>(&g->h_peak[s]), (size_t)chunks_per_slot * sizeof(float), hipHostMallocDefault));
    CK(hipEventCreateWithFlags(&g->ev_copied[s], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&g->ev_done[s], hipEventDisableTiming));
    for (int r = 0; r < PYSDR_MAX_RX; ++r) {
      CK(hipHostMalloc(reinterpret_cast<void**>(&g->h_am[(size_t)s * PYSDR_MAX_RX + r]), (size_t)g->ocap * sizeof(float2), hipHostMallocDefault));
      CK(hipHostMalloc(reinterpret_cast<void**>(&g->h_iq[(size_t)s * PYSDR_MAX_RX + r]), (size_t)g->ocap * sizeof(float2), hipHostMallocDefault));
    }
  }
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&g->d_in[i], g->cap * sizeof(float2)));
    CK(hipEventCreateWithFlags(&g->ev_free[i], hipEventDisableTiming));
  }
#undef CK
  *out = g;
  return PYSDR_OK;
}

int pysdr_ingest_buffer(pysdr_ingest* g, int slot, float** iq, size_t* cap_samples) {
  if (!g || !iq || slot < 0 || slot >= g->nslots) return PYSDR_ERR_ARG;
  if (g->in_flight[slot]) { set_last_error("pysdr_ingest_buffer: slot %d is in flight (collect it first)", slot); return PYSDR_ERR_STATE; }
  *iq = reinterpret_cast<float*>(g->h_in[slot]);
  if (cap_samples) *cap_samples = g->cap;
  return PYSDR_OK;
}

int pysdr_ingest_submit(pysdr_ingest* g, int slot, size_t n) {
  if (!g || slot < 0 || slot >= g->nslots || n < 1 || n > g->cap) return PYSDR_ERR_ARG;
  if (g->in_flight[slot]) { set_last_error("pysdr_ingest_submit: slot %d is already in flight", slot); return PYSDR_ERR_STATE; }
  pysdr_ctx* c = g->c;
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  const int b = (int)(g->seq & 1ull);
  // the copy may not overwrite a staging buffer the kernels of two chunks ago still read
  if (g->used[b]) PYSDR_HIP_CHECK(hipStreamWaitEvent(g->copy_stream, g->ev_free[b], 0));
  PYSDR_HIP_CHECK(hipMemcpyAsync(g->d_in[b], g->h_in[slot], n * sizeof(float2), hipMemcpyHostToDevice, g->copy_stream));
  PYSDR_HIP_CHECK(hipEventRecord(g->ev_copied[slot], g->copy_stream));
  PYSDR_HIP_CHECK(hipStreamWaitEvent(c->stream, g->ev_copied[slot], 0));
  // a slot of whole chunks is ONE launch sequence over all of them (same arithmetic as chunk by
  // chunk: pysdr_process_batch); anything else is a single (possibly short) chunk
  const size_t lc = (size_t)c->cfg.in_chunk;
  const int nch = (n > lc && n % lc == 0) ? (int)(n / lc) : 1;
  rc = pysdr_process_batch(c, g->d_in[b], nch, nch > 1 ? lc : n, 1);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventRecord(g->ev_free[b], c->stream));
  g->used[b] = true;
  const int nout = c->last_nout;
  if (nout > g->ocap) { set_last_error("pysdr_ingest_submit: %d outputs > capacity %d", nout, g->ocap); return PYSDR_ERR_STATE; }
  g->n_out[slot] = nout;
  g->nrx[slot] = c->last_nrx;
  for (int r = 0; r < c->last_nrx; ++r) {
    const int cx = c->last_complex[r];
    g->cx[(size_t)slot * PYSDR_MAX_RX + r] = cx;
    if (nout > 0) {
      PYSDR_HIP_CHECK(hipMemcpyAsync(g->h_am[(size_t)slot * PYSDR_MAX_RX + r], c->rx[r].d_am,
                                     (size_t)nout * (cx ? 2 : 1) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
      PYSDR_HIP_CHECK(hipMemcpyAsync(g->h_iq[(size_t)slot * PYSDR_MAX_RX + r], (c->last_par ? c->rx[r].d_y_alt : c->rx[r].d_y) + c->hy,
                                     (size_t)nout * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    }
  }
  PYSDR_HIP_CHECK(hipMemcpyAsync(g->h_peak[slot], c->d_peak, (size_t)nch * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  PYSDR_HIP_CHECK(hipEventRecord(g->ev_done[slot], c->stream));
  g->n_chunks[slot] = nch;
  g->chunk_nout[slot].assign(nch, 0);
  last_chunk_counts(c, g->chunk_nout[slot].data());
  g->in_flight[slot] = 1;
  g->seq += 1;
  return PYSDR_OK;
}

int pysdr_ingest_chunks(pysdr_ingest* g, int slot, int cap, int* nchunks, int* chunk_nout, float* peaks) {
  if (!g || slot < 0 || slot >= g->nslots || !nchunks) return PYSDR_ERR_ARG;
  if (!g->in_flight[slot]) { set_last_error("pysdr_ingest_chunks: slot %d was not submitted", slot); return PYSDR_ERR_STATE; }
  int rc = use_device(g->c->cfg.device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventSynchronize(g->ev_done[slot]));
  const int n = g->n_chunks[slot];
  *nchunks = n;
  if ((chunk_nout || peaks) && cap < n) { set_last_error("pysdr_ingest_chunks: cap %d < %d chunks", cap, n); return PYSDR_ERR_ARG; }
  for (int k = 0; k < n; ++k) {
    if (chunk_nout) chunk_nout[k] = g->chunk_nout[slot][k];
    if (peaks) peaks[k] = g->h_peak[slot][k];
  }
  return PYSDR_OK;
}

int pysdr_ingest_collect(pysdr_ingest* g, int slot, pysdr_out* outs) {
  if (!g || !outs || slot < 0 || slot >= g->nslots) return PYSDR_ERR_ARG;
  if (!g->in_flight[slot]) { set_last_error("pysdr_ingest_collect: slot %d was not submitted", slot); return PYSDR_ERR_STATE; }
  int rc = use_device(g->c->cfg.device);
  if (rc) return rc;
  PYSDR_HIP_CHECK(hipEventSynchronize(g->ev_done[slot]));
  for (int r = 0; r < g->nrx[slot]; ++r) {        // the RX set of THIS slot's submit, not of the latest call
    outs[r].am = g->h_am[(size_t)slot * PYSDR_MAX_RX + r];
    outs[r].iq = g->h_iq[(size_t)slot * PYSDR_MAX_RX + r];
    outs[r].cap = g->ocap;
    outs[r].n_out = g->n_out[slot];
    outs[r].am_is_complex = g->cx[(size_t)slot * PYSDR_MAX_RX + r];
    outs[r].peak_in = *g->h_peak[slot];
  }
  g->in_flight[slot] = 0;
  return PYSDR_OK;
}

// ---------------------------------------------------------------- RCCL (lazy dlopen: librccl is ~0.5 GB)
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return PYSDR_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { set_last_error("dlopen(librccl.so.1): %s", dlerror()); return PYSDR_ERR_RCCL; }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(lib, "ncclBroadcast"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.Broadcast || !g_rccl.CommDestroy) {
    set_last_error("librccl: missing symbols");
    dlclose(lib);
    return PYSDR_ERR_RCCL;
  }
  g_rccl.lib = lib;
  return PYSDR_OK;
}
int rccl_fail(const char* what, ncclResult_t r) {
  set_last_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
  return PYSDR_ERR_RCCL;
}
}  // namespace

int pysdr_comm_unique_id(char id_out[128]) {
  if (!id_out) return PYSDR_ERR_ARG;
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id_out, id.internal, 128);
  return PYSDR_OK;
}

int pysdr_comm_init(pysdr_ctx* c, const char id[128], int rank, int nranks) {
  if (!c || !id || rank < 0 || rank >= nranks) return PYSDR_ERR_ARG;
  int rc = rccl_load();
  if (rc) return rc;
  rc = use_device(c->cfg.device);
  if (rc) return rc;
  if (c->comm) { set_last_error("pysdr_comm_init: already initialised"); return PYSDR_ERR_STATE; }
  ncclUniqueId uid;
  memcpy(uid.internal, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return rccl_fail("ncclCommInitRank", r); }
  return PYSDR_OK;
}

int pysdr_comm_bcast(pysdr_ctx* c, void* d_buf, size_t bytes, int root) {
  if (!c || !d_buf) return PYSDR_ERR_ARG;
  if (!c->comm) { set_last_error("pysdr_comm_bcast: communicator not initialised"); return PYSDR_ERR_STATE; }
  std::lock_guard<std::recursive_mutex> run_lk(c->run_mu);
  int rc = use_device(c->cfg.device);
  if (rc) return rc;
  ncclResult_t r = g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, c->stream);
  if (r != ncclSuccess) return rccl_fail("ncclBroadcast", r);
  return PYSDR_OK;
}

int pysdr_comm_destroy(pysdr_ctx* c) {
  if (!c) return PYSDR_ERR_ARG;
  if (c->comm && g_rccl.CommDestroy) {
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    g_rccl.CommDestroy(c->comm);
  }
  c->comm = nullptr;
  return PYSDR_OK;
}

}  // extern "C"
