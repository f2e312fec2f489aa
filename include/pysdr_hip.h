/*
 * pysdr_hip.h -- C ABI of libpysdr_hip.so: the MI355X (gfx950) implementation of
 * the pySDR receiver hot path (per-sub-receiver LO mix -> rational polyphase
 * resample -> detect -> AF filter -> AGC, plus the PSD FFT).
 *
 * The reference (aa2il/pySDR) is pure Python and has no FFI of its own; its hot
 * path is the Python surface of module `sig_proc` (absent from the tree,
 * reconstructed from call sites in SURVEY.md 2.2).  Every entry point below names
 * the reference call site it stands behind.  The Python facade
 * `pysdr_amd/sig_proc.py` binds these with ctypes.
 *
 * Conventions
 *  - plain C, no C++/torch types; all functions return 0 or a negative
 *    pysdr_status (never throw, never abort) -- the reference convention is
 *    "print + carry on" (receiver.py:603-605).
 *  - caller owns every host buffer; the context owns all device memory.
 *  - pysdr_process*() are called from ONE thread (the RX thread,
 *    receiver.py:684-725).  Setters may be called concurrently from another thread
 *    (the Qt thread, gui.py:1713,1938) and take effect at the next process call.
 *  - complex data are interleaved float pairs (numpy complex64 layout).
 */
#ifndef PYSDR_HIP_H
#define PYSDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYSDR_MAX_RX 8 /* reference MAX_RX = 6 (params.py:33) */

typedef enum {
  PYSDR_OK = 0,
  PYSDR_ERR_ARG = -1,      /* bad argument                                   */
  PYSDR_ERR_NO_DEVICE = -2,/* no HIP device / hipSetDevice failed            */
  PYSDR_ERR_HIP = -3,      /* a HIP runtime call failed (see pysdr_last_error)*/
  PYSDR_ERR_FFT = -4,      /* rocFFT failure                                  */
  PYSDR_ERR_STATE = -5,    /* call order / capacity violation                */
  PYSDR_ERR_RCCL = -6      /* RCCL failure                                    */
} pysdr_status;

/* Index into the reference's MODES list (Tables.py:34). */
typedef enum {
  PYSDR_AM = 0, PYSDR_AM_SYNCH = 1, PYSDR_SSB = 2, PYSDR_USB = 3, PYSDR_LSB = 4,
  PYSDR_CW = 5, PYSDR_IQ = 6, PYSDR_WFM = 7, PYSDR_WFM2 = 8, PYSDR_NFM = 9,
  PYSDR_RTTY = 10
} pysdr_mode;

/* Rates and sizes: the fields of the reference's P object that Receiver reads
 * (params.py:405-406,440-444; receiver.py:818-820). */
typedef struct {
  double  srate;       /* P.SRATE, input complex sample rate (Hz)              */
  int32_t up, down;    /* P.UP, P.DOWN = up_dn(SRATE, FS_OUT)                  */
  int32_t in_chunk;    /* P.IN_CHUNK_SIZE, nominal samples per chunk           */
  int32_t max_chunks;  /* capacity: chunks per pysdr_process_batch call, <= 16384 */
  int32_t ntaps_dec;   /* P.FILT_LEN, prototype length at srate*up             */
  int32_t ntaps_af;    /* AF filter length at FS_OUT                           */
  int32_t device;      /* HIP device ordinal                                   */
  int32_t reserved;
} pysdr_cfg;

typedef struct pysdr_ctx pysdr_ctx;
typedef struct pysdr_spectrum pysdr_spectrum;

/* rx.agc fields read by watchdog.py:298-302 */
typedef struct { float agc, gain, maxbuf, ref, err; } pysdr_agc_state;

/* Per-RX result of one pysdr_process() call.  `am`/`iq` are caller buffers with
 * room for `cap` output samples (complex ones need 2*cap floats); either may be
 * NULL to skip the copy.  On return n_out = samples produced (1023/1024 at
 * 8 MS/s -> 48 kHz, SURVEY.md 5 "non-integer chunk ratio"). */
typedef struct {
  float*  am;            /* rx.am  : demodulated audio (receiver.py:235)       */
  float*  iq;            /* rx.iq  : baseband IQ at FS_OUT (receiver.py:265)   */
  int32_t cap;
  int32_t n_out;
  int32_t am_is_complex; /* 1 in IQ mode                                        */
  float   peak_in;       /* max |x|^2 of the raw chunk (rx.auto_mute input)    */
} pysdr_out;

/* ---- library ------------------------------------------------------------- */
const char* pysdr_strerror(int status);
const char* pysdr_last_error(void);          /* text of the last HIP/rocFFT error (thread local) */
int pysdr_device_count(int* n);
int pysdr_version(void);

/* ---- context = one wideband stream + its sub-receivers ------------------------
 * stands behind SDR_EXECUTIVE.create_Receivers (receiver.py:826-835). */
int  pysdr_create(const pysdr_cfg* cfg, pysdr_ctx** out);
void pysdr_destroy(pysdr_ctx* ctx);

/* dsp.Receiver(P, frq, irx, ...) (receiver.py:835).  lo_freq is the generator
 * frequency (= -frq); h = rx.dec.h prototype (ntaps_dec doubles);
 * af = complex AF taps (2*ntaps_af doubles, re/im interleaved). */
int pysdr_rx_add(pysdr_ctx* ctx, int mode, double lo_freq, const double* h,
                 const double* af, double bfo, int* irx);

/* rx.lo.change_freq(f) -> actual f (receiver.py:112,352; gui.py:1938,2011) */
int pysdr_set_lo(pysdr_ctx* ctx, int irx, double f_hz, double* f_actual);
/* rx.dec.h = rx.dec.filter_bank[idx] (receiver.py:127,371; gui.py:1713,1762) */
int pysdr_set_dec_taps(pysdr_ctx* ctx, int irx, const double* h, int n);
/* P.MODE / P.AF_FILTER_NUM / P.BFO as read by Receiver each chunk
 * (receiver.py:114-116,130-131; gui.py:1756-1757) */
int pysdr_set_mode(pysdr_ctx* ctx, int irx, int mode, const double* af, int n, double bfo);
/* Broadcast FM (modes WFM/WFM2; gui.py:1703-1704 `rx.demod.wfm_video.h = wfm_filter_bank[idx]`):
 * video = pre-detection low-pass at SRATE (ntaps_dec doubles); resamp = prototype of the
 * fs1 -> FS_OUT resampler (a multiple of up2 doubles).  pysdr_wfm_params gives the IF
 * decimation d1 (fs1 = SRATE/d1 ~ 250 kHz) and up2/down2 = up_dn(fs1, FS_OUT). */
int pysdr_wfm_params(double srate, double fs_out, int* d1, int* up2, int* down2);
int pysdr_set_wfm_taps(pysdr_ctx* ctx, int irx, const double* video, int nv, const double* resamp, int nr);
/* rx.agc.reset() / rx.demod.am_pll.reset() (receiver.py:648-649): what = 1 AGC, 2 PLL, 3 both */
int pysdr_reset(pysdr_ctx* ctx, int irx, unsigned what);
int pysdr_agc_get(pysdr_ctx* ctx, int irx, pysdr_agc_state* st);
/* The serial PLLs (rx.demod.am_pll, receiver.py:649; the WFM2 pilot PLL) run time-parallel on
 * long calls: `segments` the last call was cut into, `patched` = segments a serial patch-up pass
 * had to recompute because the loop had not forgotten its start state (0 for a locked loop). */
int pysdr_pll_stats(pysdr_ctx* ctx, int irx, int* segments, int* patched);
/* A/B knob for the above: at most `max_segments` per call (1 = the plain serial walk, 0 = default) */
int pysdr_set_pll_segments(pysdr_ctx* ctx, int max_segments);
/* WFM2 pilot loop of the last batch: the widest join between two segments of the time-parallel walk BEFORE any patching --
 * |phase difference| in words of 2^32 (accepted up to 512) and |integrator difference| in rad/sample (up to 1e-9).  A
 * diagnostic of this build (no reference call site): what a shorter / cheaper warm-up setting (PYSDR_WFM_PLL) is judged by. */
int pysdr_pll_join_margin(pysdr_ctx* ctx, int irx, int* max_words, float* max_dw);
/* AM-Synch carrier loop of the last batch: how many of its segments got their start state from ONE LINEAR SOLVE over the
 * warm-up window instead of walking it (the loop is linear in the phase domain while its detector does not wrap, which is
 * checked per window; a noisy or badly guessed window is walked as before).  A diagnostic of this build. */
int pysdr_pll_linear_starts(pysdr_ctx* ctx, int irx, int* n);
/* NFM noise squelch (north_star "AGC/squelch"; design notes sigs/squelch.m:92-145): per chunk the
 * mean |2nd difference| of the discriminator output is smoothed (one pole) and the chunk is
 * muted while it exceeds `thresh`; thresh <= 0 disables (default). */
int pysdr_set_squelch(pysdr_ctx* ctx, int irx, float thresh);
int pysdr_squelch_get(pysdr_ctx* ctx, int irx, float* level, int* open);
/* The RATIO squelch, as the only design the reference holds sketches it (sigs/squelch.m:92-145; no run-time call site): z1 =
 * low-pass < 3 kHz and z2 = high-pass > 4 kHz of the discriminator output (`lp`, `hp`: FIR taps, ntaps <= 64, for the
 * context's FS_OUT; there: elliptic IIRs of order 5, :103-105), sq1 / sq2 = one-pole envelopes of |z1| / |z2| per SAMPLE with
 * alpha = 0.001 (:131-134), and the gate open while sq1 / sq2 >= min_ratio (:145) -- independent of the signal's level.  One
 * decision per chunk (the AGC block), on the envelopes behind its last sample.  min_ratio > 0 arms it for that sub-receiver
 * (NFM) and takes precedence over pysdr_set_squelch; 0 disarms.  pysdr_squelch_ratio_get: the two envelopes and the gate
 * behind the last call. */
int pysdr_set_squelch_ratio(pysdr_ctx* ctx, int irx, float min_ratio, const float* lp, const float* hp, int ntaps);
int pysdr_squelch_ratio_get(pysdr_ctx* ctx, int irx, float* sq_lp, float* sq_hp, int* open);
/* AGC on/off and reference level */
int pysdr_set_agc(pysdr_ctx* ctx, int irx, int enable, float ref);

/* rx.demod_data(x) for every RX of the stream on ONE chunk of host samples
 * (receiver.py:231-235,724-725).  n = complex samples; outs[num_rx]. */
int pysdr_process(pysdr_ctx* ctx, const float* iq_interleaved, size_t n, pysdr_out* outs);

/* Same arithmetic on `nchunks` consecutive chunks of `chunk_len` samples in one
 * launch sequence; results are identical, bit for bit, to nchunks pysdr_process() calls -- except
 * in the two modes with a serial loop (AM-Synch carrier PLL, WFM2 pilot PLL): a long call runs the
 * loop in segments whose joins are accepted within 2e-6 rad / 512 words of 2^32 of phase, so there
 * the audio equals the chunk-by-chunk run within the 1e-5 parity bar, not bitwise
 * (pysdr_set_pll_segments(ctx, 1) forces the serial walk).
 * on_device != 0: iq is a device pointer (pysdr_dev_alloc) -- the replay path
 * (receiver.py:541-559) and the throughput benchmark. */
int pysdr_process_batch(pysdr_ctx* ctx, const void* iq, int nchunks, size_t chunk_len, int on_device);
/* Batch pipelining (a build feature, no reference call site; the analogue in the reference is MP_SCHEME 2/3 running the
 * demodulators beside the chunk acquisition, receiver.py:726-739).  A call is three groups of launches: F the front end
 * (mix + decimate, and the parallel kernel in front of a serial loop), P the segment walks of a serial loop (AM-Synch
 * carrier PLL, WFM2 pilot PLL: thousands of single-wave chains, latency bound, no LDS), T the tail (detector + AF
 * filter, AGC, output; for broadcast FM the audio resampler first).  Overlapped, P(k) is queued on a second HIP stream and
 * T(k) is DEFERRED: it is queued behind F(k+1) by the next pysdr_process_batch, or at once by whatever asks for the
 * call's results (pysdr_fetch, pysdr_sync, the state getters, a setter with pending work) -- so the walks of call k run
 * beside the tail of call k-1 and the front end of call k+1.  Results, state and every other entry point are unchanged, bit
 * for bit (tests/test_gpu_overlap.py).  enable = 0 off (default), 1 = the calls with a serial loop in them, 2 = every
 * call (tests: without a loop nothing runs on the second stream, the buffers and the deferral are exercised all the
 * same).  A context that feeds an ingest ring runs single-stream: pysdr_ingest_create* switches this off, and switching it
 * on then fails with PYSDR_ERR_STATE.  pysdr_last_call_overlapped: the form the last call really took. */
/* 0 for a library built from the sources as they are; otherwise a 31-bit hash of the extra compiler flags (A/B switches
 * from PYSDR_*_FLAGS under PYSDR_TUNING=1, or the diagnostic build's) it was built with -- echoed by bench.py. */
int pysdr_build_flags_hash(void);
int pysdr_set_overlap(pysdr_ctx* ctx, int enable);
int pysdr_get_overlap(pysdr_ctx* ctx);
int pysdr_last_call_overlapped(pysdr_ctx* ctx);
/* Copy results of the last process call to the host.  am/iq may be NULL.
 * chunk_nout[nchunks] (may be NULL) receives the per-chunk output counts,
 * peaks[nchunks] the raw-chunk max |x|^2. */
int pysdr_fetch(pysdr_ctx* ctx, int irx, float* am, float* iq, int cap, int* n_out,
                int* am_is_complex, int* chunk_nout, float* peaks);
int pysdr_sync(pysdr_ctx* ctx);

/* Per-call timing with HIP events recorded on the context's stream (a ring of the last
 * 64 calls; back = 0 is the most recent call): which = 0 mix+decimate kernel (front end),
 * 1 detector/AF/AGC kernels and the history roll behind them, 2 whole call, 3 from the start of the call before it to the start of
 * this one (the period of a step when calls follow each other on the stream).  Events are only
 * recorded while enabled. */
int pysdr_set_profile(pysdr_ctx* ctx, int enable);
int pysdr_get_elapsed_ms(pysdr_ctx* ctx, int which, int back, float* ms);
/* tuning knob: LDS bytes of input tile per workgroup in the mix+decimate kernel */
int pysdr_set_tile(pysdr_ctx* ctx, int tile_bytes, int threads);
/* What the context was really created with, so that a benchmark line can echo it (VERDICT r1
 * hygiene): out = {diagnostic build (-DPYSDR_DIAG) 0/1, PYSDR_DEBUG_FLAGS (always 0 outside a
 * diagnostic build), workgroups per CU, output-stage flush cap, tile bytes, threads, CUs,
 * long single-RX prototypes on the matrix cores 0/1}. */
int pysdr_get_tuning(pysdr_ctx* ctx, int32_t out[8]);

/* ---- signal_generator.quad_mixer (receiver.py:552-553,822) --------------------
 * y = x * exp(+j*phi_n), 32-bit phase accumulator, returns phase after n samples */
int pysdr_quad_mixer(int device, const float* x, float* y, size_t n, uint32_t phase0,
                     uint32_t fword, uint32_t* phase_out);
uint32_t pysdr_freq_word(double f_hz, double fs_hz, double* f_actual);

/* ---- convolver.convolve_fast (receiver.py:207,216,862): streaming real FIR.
 * xx = [ntaps-1 history samples | n new samples]; y[i] = sum_k h[k]*xx[i+ntaps-1-k] */
int pysdr_fir_real(int device, const float* xx, const float* h, int ntaps, float* y, size_t n);

/* ---- spectrum (Plotting.py:376,462; gui.py:611-631) --------------------------- */
int  pysdr_spectrum_create(int device, int chunk_size, int nfft, int max_frames,
                           const float* window, pysdr_spectrum** out);
void pysdr_spectrum_destroy(pysdr_spectrum* sp);
/* spectrum.periodogram: one frame of chunk_size (complex or real) host samples ->
 * psd_db[nfft] (complex, fftshifted) or psd_db[nfft/2] (real input) */
int pysdr_spectrum_frame(pysdr_spectrum* sp, const float* x, int is_complex, int db,
                         float* psd_out, int* n_out);
/* nframes frames taken every `hop` samples from a device-resident complex stream;
 * d_out = device [nframes][nfft] float (dB, fftshifted).  The fused 64k path deals its groups of frames over two HIP
 * streams by default (per-kernel durations of a profiler trace then overlap; PYSDR_TUNING=1 PYSDR_PSD_STREAMS=1 for
 * un-overlapped kernel times, INTEGRATION.md "tuning environment"); the call's own stream waits for the side stream. */
int pysdr_spectrum_batch(pysdr_spectrum* sp, const void* d_iq, int nframes, size_t hop,
                         void* d_out);
int pysdr_spectrum_sync(pysdr_spectrum* sp);
/* out = {frames per launch group of the fused 64k path, rocFFT forced 0/1, streams the groups are dealt over,
 * four-step intermediate stored as 24-bit block-scaled fixed point 0/1} */
int pysdr_spectrum_get_tuning(pysdr_spectrum* sp, int32_t out[4]);
int pysdr_spectrum_elapsed_ms(pysdr_spectrum* sp, float* ms);
/* Ordering between the spectrum's stream and a receiver context's stream (both read the same
 * device-resident chunk; Plotting.py:462 runs the PSD after the chunk's demod in one thread):
 * direction 0: spectrum work queued from now on starts after everything queued on ctx so far;
 * direction 1: ctx work queued from now on starts after everything queued on sp so far;
 * direction 2: like 0, but only behind the FRONT END (mix + decimate) of ctx's last process call --
 *   the PSD reads the same input and may run beside the audio-rate stages. */
int pysdr_spectrum_order(pysdr_spectrum* sp, pysdr_ctx* ctx, int direction);

/* ---- waterfall numeric back-end (three_box_plot.plot, Plotting.py:536-626; shift_waterfall
 * :689-695): device history ring [nfft][ncols]; push = shift-in of one PSD line (shorter lines
 * padded with -1e38); roll = np.roll(wf, -nbins, axis=0); image = max(wf - median(mean(wf[:,
 * -cnt:], 1)), nanmax(wf - bkgnd) - pan_dr), returned column-major: image_out[c][i], c = 0 the
 * oldest column (numpy: image_out.reshape(ncols, nfft).T).  mean_out[nfft] is the PSD2 vector the
 * reference feeds to find_peaks (Plotting.py:583,595). */
typedef struct pysdr_waterfall pysdr_waterfall;
int  pysdr_waterfall_create(int device, int nfft, int ncols, pysdr_waterfall** out);
void pysdr_waterfall_destroy(pysdr_waterfall* wf);
int  pysdr_waterfall_push(pysdr_waterfall* wf, const float* line_db, int n, int on_device);
int  pysdr_waterfall_roll(pysdr_waterfall* wf, int nbins);
int  pysdr_waterfall_image(pysdr_waterfall* wf, float pan_dr, float* image_out, float* mean_out,
                           float* bkgnd_out);
/* The same with the dynamic-range maximum taken over the rows [0, npsd) only, npsd = length of the
 * PSD line just pushed: `zz = self.wf[0:npsd,:] - med; zmax = np.nanmax(zz)` (Plotting.py:618-619;
 * a real-input line is half length, :536-540).  image_out stays [ncols][nfft]; the reference draws
 * its first npsd rows. */
int  pysdr_waterfall_image_rows(pysdr_waterfall* wf, float pan_dr, int npsd, float* image_out,
                                float* mean_out, float* bkgnd_out);

/* The peak pick of the display, `peaks, _ = signal.find_peaks(PSD2, distance=dist, height=bkgnd+10)` (Plotting.py:594-602), on
 * the device: local maxima with flat tops (midpoint of a plateau, none at the ends of the line), height >= `height` (double),
 * then SciPy's priority-by-height distance rule with distance = ceil(PEAK_DIST / df) >= 1 (a kept peak removes every peak
 * closer than that).  line == NULL: over the first n values of the averaged line the last pysdr_waterfall_image left on the
 * device (PSD2); otherwise over the n host values given.  idx_out[min(*n_out, cap)] ascending; *n_out = how many there are.
 * Equal heights closer than `distance`: SciPy's choice rests on an unstable argsort; here the higher index outranks. */
int  pysdr_waterfall_peaks(pysdr_waterfall* wf, const float* line, int n, double height, int distance, int* idx_out, int cap,
                           int* n_out);

/* ---- device memory for resident streams --------------------------------------- */
int pysdr_dev_alloc(int device, size_t bytes, void** out);
int pysdr_dev_free(int device, void* p);
int pysdr_dev_upload(int device, void* dst, const void* src_host, size_t bytes);
int pysdr_dev_download(int device, void* dst_host, const void* src, size_t bytes);
int pysdr_dev_copy(int device, void* dst, const void* src, size_t bytes);

/* ---- multi-GPU: RCCL broadcast of the wideband chunk when ONE stream's
 * sub-receivers are split across GPUs (the analogue of MP_SCHEME 3's
 * que[irx].put(x), receiver.py:728-739).  id = 128-byte ncclUniqueId. */
int pysdr_comm_unique_id(char id_out[128]);
int pysdr_comm_init(pysdr_ctx* ctx, const char id[128], int rank, int nranks);
int pysdr_comm_bcast(pysdr_ctx* ctx, void* d_buf, size_t bytes, int root);
int pysdr_comm_destroy(pysdr_ctx* ctx);

/* ---- ingest ring (receiver.py:579-631, soapy.py:33-48) ------------------
 * The step in front of pysdr_process for a live device: `nslots` pinned host chunk buffers
 * that sdr.readStream() fills directly, an asynchronous H2D copy on its own stream, the
 * chunk's kernels behind it and an asynchronous D2H of every sub-receiver's result into pinned
 * per-slot buffers.  submit() returns at once, so the host assembles chunk k+1 (short reads,
 * xold carry) while chunk k is copied and demodulated; collect() waits for one slot and hands
 * out pointers into its result buffers (valid until the slot is submitted again).  Destroy the
 * ring before the context it was created on. */
typedef struct pysdr_ingest pysdr_ingest;
int  pysdr_ingest_create(pysdr_ctx* ctx, int nslots, pysdr_ingest** out);
/* Slots of `chunks_per_slot` chunks (<= cfg.max_chunks): one DMA, one launch sequence and one set of
 * result copies per slot instead of per chunk -- the per-chunk fixed costs (about ten kernel
 * launches, 2 NUM_RX + 1 copies) are what bounds the one-chunk ring at 1.6 GS/s.  A slot submitted
 * with a whole number of IN_CHUNK_SIZE chunks is processed exactly as that many pysdr_process
 * calls; pysdr_ingest_chunks tells the per-chunk output counts and raw peaks (rx.auto_mute input)
 * so that the caller can cut the slot's audio back into chunks (receiver.py:238-252). */
int  pysdr_ingest_create_batched(pysdr_ctx* ctx, int nslots, int chunks_per_slot, pysdr_ingest** out);
int  pysdr_ingest_chunks(pysdr_ingest* ing, int slot, int cap, int* nchunks, int* chunk_nout, float* peaks);
void pysdr_ingest_destroy(pysdr_ingest* ing);
int  pysdr_ingest_buffer(pysdr_ingest* ing, int slot, float** iq, size_t* cap_samples);
int  pysdr_ingest_submit(pysdr_ingest* ing, int slot, size_t n);
int  pysdr_ingest_collect(pysdr_ingest* ing, int slot, pysdr_out* outs);

#ifdef __cplusplus
}
#endif
#endif /* PYSDR_HIP_H */
