// Driver of the sanitizer build of the HOST half of libpysdr_hip.so (tests/host_san): the C ABI of
// include/pysdr_hip.h exercised end to end over the fake HIP runtime and the checking launch layer
// (stub_kernels.cpp) -- every BASELINE rate, ragged call lengths, every setter between calls, the
// lazily allocated buffers (AM-Synch, WFM), the spectrum object on both its paths, the ingest ring's
// slot state machine with its misuse errors.  `san_main race` runs pysdr_process on one thread against
// the setters on another (the reference's RX thread vs Qt thread) for ThreadSanitizer.
//   build + run: tests/host_san/run.sh   (tests/test_host_sanitizers.py does that)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pysdr_hip.h"

#define OK(expr)                                                                                    \
  do {                                                                                              \
    const int rc_ = (expr);                                                                         \
    if (rc_ != 0) { std::fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #expr, rc_, pysdr_last_error()); std::exit(1); } \
  } while (0)
#define FAILS(expr)                                                                                 \
  do {                                                                                              \
    if ((expr) == 0) { std::fprintf(stderr, "%s:%d %s unexpectedly succeeded\n", __FILE__, __LINE__, #expr); std::exit(1); } \
  } while (0)

struct Rate { double fs; int up, down, in_chunk; };
static const Rate kRates[] = {{8e6, 3, 500, 170666}, {2.048e6, 3, 128, 43690}, {256e3, 3, 16, 5461}, {10e6, 3, 625, 213333},
                              {1.024e6, 3, 64, 21845}, {6e6, 1, 125, 128000}};

static std::vector<double> taps(int n, double scale = 1.0) {
  std::vector<double> h(n);
  for (int i = 0; i < n; ++i) h[i] = scale * std::sin(0.01 * (i + 1)) / n;
  return h;
}

static int g_overlap = 0;      // passes of main(): 0 single-stream, 2 every call in two overlapped halves, 1 the calls with a serial loop (pysdr_set_overlap)

static pysdr_ctx* make_ctx(const Rate& r, int max_chunks, int ntaps_dec, int ntaps_af) {
  pysdr_cfg cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.srate = r.fs; cfg.up = r.up; cfg.down = r.down; cfg.in_chunk = r.in_chunk;
  cfg.max_chunks = max_chunks; cfg.ntaps_dec = ntaps_dec; cfg.ntaps_af = ntaps_af;
  pysdr_ctx* c = nullptr;
  OK(pysdr_create(&cfg, &c));
  if (g_overlap) {
    FAILS(pysdr_set_overlap(c, 3));
    OK(pysdr_set_overlap(c, g_overlap));
    if (!pysdr_get_overlap(c)) { std::fprintf(stderr, "overlap did not switch on\n"); std::exit(1); }     // (PYSDR_OVERLAP may overrule WHICH form)
  }
  return c;
}

static void run_calls(pysdr_ctx* c, const Rate& r, int nrx, int max_chunks, const std::vector<size_t>& lens) {
  const size_t cap_in = (size_t)max_chunks * r.in_chunk;
  std::vector<float> x(2 * cap_in, 0.25f);
  const int cap_out = (int)(cap_in * r.up / r.down) + 8;
  std::vector<std::vector<float>> am(nrx, std::vector<float>(2 * cap_out)), iq(nrx, std::vector<float>(2 * cap_out));
  std::vector<pysdr_out> outs(nrx);
  for (size_t n : lens) {
    for (int i = 0; i < nrx; ++i) { outs[i].am = am[i].data(); outs[i].iq = iq[i].data(); outs[i].cap = cap_out; }
    if (n <= (size_t)r.in_chunk * max_chunks) OK(pysdr_process(c, x.data(), n, outs.data()));
  }
  // a batch of whole chunks + per-chunk counts
  OK(pysdr_process_batch(c, x.data(), max_chunks, r.in_chunk, 0));
  std::vector<int> cn(max_chunks);
  std::vector<float> pk(max_chunks);
  int n_out = 0, cx = 0;
  for (int i = 0; i < nrx; ++i) {
    OK(pysdr_fetch(c, i, am[i].data(), iq[i].data(), cap_out, &n_out, &cx, cn.data(), pk.data()));
    int s = 0;
    for (int k = 0; k < max_chunks; ++k) s += cn[k];
    if (s != n_out) { std::fprintf(stderr, "chunk counts %d != n_out %d\n", s, n_out); std::exit(1); }
    FAILS(pysdr_fetch(c, i, am[i].data(), nullptr, n_out - 1, nullptr, nullptr, nullptr, nullptr));   // cap too small
  }
  FAILS(pysdr_process_batch(c, x.data(), max_chunks + 1, r.in_chunk, 0));                             // over capacity
}

static void narrowband(const Rate& r, int ntaps_dec) {
  const int max_chunks = 3, ntaps_af = 255, nrx = 4;
  pysdr_ctx* c = make_ctx(r, max_chunks, ntaps_dec, ntaps_af);
  const auto h = taps(ntaps_dec), af = taps(2 * ntaps_af);
  const int modes[nrx] = {PYSDR_USB, PYSDR_CW, PYSDR_NFM, PYSDR_AM};
  for (int i = 0; i < nrx; ++i) {
    int irx = -1;
    OK(pysdr_rx_add(c, modes[i], -1000.0 * (i + 1), h.data(), af.data(), i == 1 ? 700.0 : 0.0, &irx));
    if (irx != i) std::exit(1);
  }
  OK(pysdr_set_profile(c, 1));
  const size_t L = (size_t)r.in_chunk;
  run_calls(c, r, nrx, max_chunks, {L, 1, 2, 3, 17, L - 7, L + 11, 3 * L, 333, 2 * L + 1, L});
  // controls between calls (receiver.py:112-131,648-649; gui.py:1713,1938)
  double fa = 0;
  OK(pysdr_set_lo(c, 2, -4567.0, &fa));
  const auto h2 = taps(ntaps_dec, 0.5);
  OK(pysdr_set_dec_taps(c, 0, h2.data(), ntaps_dec));
  FAILS(pysdr_set_dec_taps(c, 0, h2.data(), ntaps_dec - 1));
  OK(pysdr_set_mode(c, 3, PYSDR_AM_SYNCH, af.data(), ntaps_af, 0.0));          // lazily allocates the PLL buffer
  OK(pysdr_set_mode(c, 0, PYSDR_IQ, af.data(), ntaps_af, 0.0));                // complex audio
  FAILS(pysdr_set_mode(c, 0, PYSDR_IQ, af.data(), ntaps_af + 1, 0.0));
  OK(pysdr_reset(c, 1, 3));
  OK(pysdr_set_agc(c, 1, 0, 0.4f));
  OK(pysdr_set_squelch(c, 2, 0.05f));
  {
    // the ratio squelch: its two FIRs belong to the context, the block sums are allocated when it is first armed
    std::vector<float> lp(63, 1.0f / 63), hp(63, 0.f);
    hp[31] = 1.f;
    FAILS(pysdr_set_squelch_ratio(c, 2, 2.0f, lp.data(), hp.data(), 65));
    FAILS(pysdr_set_squelch_ratio(c, 2, 2.0f, nullptr, hp.data(), 63));
    OK(pysdr_set_squelch_ratio(c, 2, 2.0f, lp.data(), hp.data(), 63));
    OK(pysdr_set_squelch_ratio(c, 1, 0.0f, nullptr, nullptr, 0));                // disarming needs no taps
  }
  run_calls(c, r, nrx, max_chunks, {L, 5, L});
  {
    float sq1 = 0, sq2 = 0;
    int gate = 0;
    OK(pysdr_squelch_ratio_get(c, 2, &sq1, &sq2, &gate));
    FAILS(pysdr_squelch_ratio_get(c, 9, &sq1, &sq2, &gate));
  }
  pysdr_agc_state st;
  OK(pysdr_agc_get(c, 3, &st));
  int seg = 0, pat = 0, open = 0;
  float lvl = 0;
  OK(pysdr_pll_stats(c, 3, &seg, &pat));
  OK(pysdr_squelch_get(c, 2, &lvl, &open));
  OK(pysdr_set_pll_segments(c, 1));
  run_calls(c, r, nrx, max_chunks, {L});
  float ms = 0;
  for (int which = 0; which < 4; ++which) OK(pysdr_get_elapsed_ms(c, which, 0, &ms));
  FAILS(pysdr_get_elapsed_ms(c, 4, 0, &ms));
  int32_t tune[8];
  OK(pysdr_get_tuning(c, tune));
  FAILS(pysdr_set_lo(c, 9, 0.0, &fa));
  FAILS(pysdr_agc_get(c, -1, &st));
  // ingest ring: slot state machine
  pysdr_ingest* g = nullptr;
  OK(pysdr_ingest_create_batched(c, 3, 2, &g));
  // a ring runs its context single-stream: creating one switches the overlap off, and it stays off while the ring lives
  if (pysdr_get_overlap(c)) { std::fprintf(stderr, "ingest ring on an overlapped context\n"); std::exit(1); }
  FAILS(pysdr_set_overlap(c, 1));
  OK(pysdr_set_overlap(c, 0));
  { pysdr_ingest* bad = nullptr; FAILS(pysdr_ingest_create_batched(c, 3, max_chunks + 1, &bad)); }
  float* buf = nullptr;
  size_t cap = 0;
  std::vector<pysdr_out> outs(nrx);
  for (int round = 0; round < 4; ++round) {
    const int slot = round % 3;
    OK(pysdr_ingest_buffer(g, slot, &buf, &cap));
    const size_t n = (round == 2) ? L - 5 : 2 * L;                               // whole chunks, and a short single chunk
    for (size_t i = 0; i < 2 * n; ++i) buf[i] = 0.1f;
    OK(pysdr_ingest_submit(g, slot, n));
    FAILS(pysdr_ingest_submit(g, slot, n));                                      // already in flight
    int nch = 0, cn[4];
    float pk[4];
    OK(pysdr_ingest_chunks(g, slot, 4, &nch, cn, pk));
    OK(pysdr_ingest_collect(g, slot, outs.data()));
    FAILS(pysdr_ingest_collect(g, slot, outs.data()));                           // not submitted any more
    for (int i = 0; i < nrx; ++i) {                                              // the result buffers hold n_out samples
      volatile float s = 0;
      for (int k = 0; k < outs[i].n_out * (outs[i].am_is_complex ? 2 : 1); ++k) s = s + outs[i].am[k];
      for (int k = 0; k < 2 * outs[i].n_out; ++k) s = s + outs[i].iq[k];
    }
  }
  FAILS(pysdr_ingest_submit(g, 0, cap + 1));
  FAILS(pysdr_ingest_submit(g, 7, 1));
  pysdr_ingest_destroy(g);
  OK(pysdr_set_overlap(c, g_overlap));                                           // the ring is gone: allowed again
  run_calls(c, r, nrx, max_chunks, {L, 3});                                      // ... and the stream goes on from whichever buffer of the pair is current
  OK(pysdr_set_overlap(c, 0));
  run_calls(c, r, nrx, max_chunks, {L});
  pysdr_destroy(c);
}

// One sub-receiver with the reference's default 1001-tap prototype at the am.py rate: the matrix-core form of the
// mix + decimate kernel (mixdec_mfma.hip).  Ragged and odd call lengths flip the parity of its LDS image.
namespace pysdr { extern int g_mfma_launches; }
static void single_rx_long_prototype(const Rate& r) {
  const int max_chunks = 3, ntaps_dec = 1001, ntaps_af = 255;
  pysdr_ctx* c = make_ctx(r, max_chunks, ntaps_dec, ntaps_af);
  const auto h = taps(ntaps_dec), af = taps(2 * ntaps_af);
  int irx = -1;
  OK(pysdr_rx_add(c, PYSDR_AM, 100e3, h.data(), af.data(), 0.0, &irx));
  const size_t L = (size_t)r.in_chunk;
  const int before = pysdr::g_mfma_launches;
  run_calls(c, r, 1, max_chunks, {L, 1, 2, 3, 17, L - 7, L + 11, 3 * L, 333, 2 * L + 1, L, 8191, 8193, 1, 1, 255, 256, 257});
  const char* tun = getenv("PYSDR_TUNING");
  const char* off = getenv("PYSDR_MIXDEC_MFMA");
  const bool expect = !(tun && atoi(tun) > 0 && off && atoi(off) == 0);
  if ((pysdr::g_mfma_launches > before) != expect) { std::fprintf(stderr, "matrix-core path: launches %d, expected %d\n", pysdr::g_mfma_launches - before, (int)expect); std::exit(1); }
  // Mode changes across the forms of a call: AM-Synch batches (overlapped when the pass allows it: the pairs' current buffer
  // alternates and is left on the SECOND one by an odd number of calls), then modes whose buffers have not existed yet
  // (their second buffer must come into being as the current one), and back.
  std::vector<float> x(2 * (size_t)max_chunks * r.in_chunk, 0.2f);
  for (int round = 0; round < 2; ++round) {
    OK(pysdr_set_mode(c, 0, PYSDR_AM_SYNCH, af.data(), ntaps_af, 0.0));
    for (int k = 0; k < 3; ++k) OK(pysdr_process_batch(c, x.data(), max_chunks, r.in_chunk, 0));
    OK(pysdr_set_mode(c, 0, PYSDR_NFM, af.data(), ntaps_af, 0.0));
    OK(pysdr_process_batch(c, x.data(), 2, r.in_chunk, 0));
    if (r.fs == 2.048e6) {                                  // a rate pysdr_wfm_params accepts
      int d1 = 0, up2 = 0, down2 = 0;
      if (pysdr_wfm_params(r.fs, 48000.0, &d1, &up2, &down2) == 0) {
        const auto video = taps(ntaps_dec), res = taps(64 * up2);
        OK(pysdr_set_wfm_taps(c, 0, video.data(), ntaps_dec, res.data(), 64 * up2));
        OK(pysdr_set_mode(c, 0, PYSDR_WFM2, af.data(), ntaps_af, 0.0));
        for (int k = 0; k < 2 + round; ++k) OK(pysdr_process_batch(c, x.data(), max_chunks, r.in_chunk, 0));
        OK(pysdr_set_mode(c, 0, PYSDR_WFM, af.data(), ntaps_af, 0.0));
        OK(pysdr_process_batch(c, x.data(), 1, r.in_chunk, 0));
      }
    }
    run_calls(c, r, 1, max_chunks, {L, 5});
  }
  pysdr_destroy(c);
}

static void broadcast_fm() {
  const Rate r = kRates[3];
  const int max_chunks = 4, ntaps = 255;
  pysdr_ctx* c = make_ctx(r, max_chunks, ntaps, ntaps);
  const auto h = taps(ntaps), af = taps(2 * ntaps);
  int irx = -1;
  OK(pysdr_rx_add(c, PYSDR_WFM2, -300e3, h.data(), af.data(), 0.0, &irx));
  std::vector<float> x(2 * (size_t)max_chunks * r.in_chunk, 0.2f);
  FAILS(pysdr_process_batch(c, x.data(), 1, r.in_chunk, 0));                     // WFM taps never set
  int d1 = 0, up2 = 0, down2 = 0;
  OK(pysdr_wfm_params(r.fs, 48000.0, &d1, &up2, &down2));
  const auto video = taps(ntaps), res = taps(64 * up2);
  OK(pysdr_set_wfm_taps(c, 0, video.data(), ntaps, res.data(), 64 * up2));
  run_calls(c, r, 1, max_chunks, {(size_t)r.in_chunk, 7, (size_t)r.in_chunk - 3, 2 * (size_t)r.in_chunk + 1});
  OK(pysdr_set_mode(c, 0, PYSDR_WFM, af.data(), ntaps, 0.0));
  run_calls(c, r, 1, max_chunks, {(size_t)r.in_chunk});
  int seg = 0, pat = 0;
  OK(pysdr_pll_stats(c, 0, &seg, &pat));
  pysdr_destroy(c);
}

// A narrow-band and a broadcast-FM sub-receiver in one context: the call is refused (one context runs one pipeline,
// receiver.py:718-719) and the context is torn down with whatever the refused call left behind.
static void mixed_modes_are_refused() {
  const Rate r = kRates[1];
  const int ntaps = 255;
  pysdr_ctx* c = make_ctx(r, 1, ntaps, ntaps);
  const auto h = taps(ntaps), af = taps(2 * ntaps);
  int irx = -1;
  OK(pysdr_rx_add(c, PYSDR_AM, 100e3, h.data(), af.data(), 0.0, &irx));
  OK(pysdr_rx_add(c, PYSDR_AM, 200e3, h.data(), af.data(), 0.0, &irx));
  int d1 = 0, up2 = 0, down2 = 0;
  OK(pysdr_wfm_params(r.fs, 48000.0, &d1, &up2, &down2));
  const auto res = taps(64 * up2);
  OK(pysdr_set_wfm_taps(c, 1, h.data(), ntaps, res.data(), 64 * up2));
  OK(pysdr_set_mode(c, 1, PYSDR_WFM, af.data(), ntaps, 0.0));
  std::vector<float> x(2 * (size_t)r.in_chunk, 0.f);
  std::vector<float> am(4 * 2048), iq(4 * 2048);
  pysdr_out outs[2];
  for (int i = 0; i < 2; ++i) { outs[i].am = am.data() + 4096 * i; outs[i].iq = iq.data() + 4096 * i; outs[i].cap = 2048; }
  FAILS(pysdr_process(c, x.data(), r.in_chunk, outs));
  FAILS(pysdr_process(c, x.data(), r.in_chunk, outs));
  pysdr_destroy(c);
}

static void spectrum() {
  std::vector<float> win(32768, 1.0f);
  pysdr_spectrum* sp = nullptr;
  OK(pysdr_spectrum_create(0, 32768, 65536, 1000, win.data(), &sp));            // the fused 64k path
  void *d_x = nullptr, *d_o = nullptr;
  const int nframes = 1000;
  OK(pysdr_dev_alloc(0, (size_t)nframes * 32768 * 8, &d_x));
  OK(pysdr_dev_alloc(0, (size_t)nframes * 65536 * 4, &d_o));
  OK(pysdr_spectrum_batch(sp, d_x, nframes, 32768, d_o));                        // 448 + 448 + 104 frames
  OK(pysdr_spectrum_batch(sp, d_x, 37, 800000 < (size_t)nframes * 32768 / 37 ? 800000 : 32768, d_o));
  FAILS(pysdr_spectrum_batch(sp, d_x, nframes + 1, 32768, d_o));
  std::vector<float> one(2 * 32768, 0.1f), psd(65536);
  int n_out = 0;
  OK(pysdr_spectrum_frame(sp, one.data(), 1, 1, psd.data(), &n_out));
  OK(pysdr_spectrum_sync(sp));
  float ms = 0;
  OK(pysdr_spectrum_elapsed_ms(sp, &ms));
  int32_t t4[4];
  OK(pysdr_spectrum_get_tuning(sp, t4));
  pysdr_spectrum_destroy(sp);
  // the rocFFT path at the sizes the unchanged GUI passes (Plotting.py:370-376) and the AF PSD (real input)
  for (int chunk : {32818, 4096}) {
    std::vector<float> w(chunk, 1.0f), xin(2 * (size_t)chunk, 0.1f), out(2 * (size_t)chunk);
    OK(pysdr_spectrum_create(0, chunk, 2 * chunk, 4, w.data(), &sp));
    OK(pysdr_spectrum_frame(sp, xin.data(), 1, 1, out.data(), &n_out));
    if (n_out != 2 * chunk) std::exit(1);
    OK(pysdr_spectrum_frame(sp, xin.data(), 0, 1, out.data(), &n_out));          // real input: first NFFT/2 bins
    if (n_out != chunk) std::exit(1);
    pysdr_spectrum_destroy(sp);
  }
  OK(pysdr_dev_free(0, d_x));
  OK(pysdr_dev_free(0, d_o));
  // stand-alone helpers
  std::vector<float> a(2 * 1001, 0.5f), b(2 * 1001);
  uint32_t ph = 0;
  OK(pysdr_quad_mixer(0, a.data(), b.data(), 1001, 123u, 456789u, &ph));
  std::vector<float> xx(255 - 1 + 500, 0.1f), hh(255, 0.01f), yy(500);
  OK(pysdr_fir_real(0, xx.data(), hh.data(), 255, yy.data(), 500));
}

static void race() {
  // one thread processes chunks, another turns the knobs (receiver.py RX thread vs the Qt thread)
  const Rate r = kRates[2];
  const int ntaps = 255;
  pysdr_ctx* c = make_ctx(r, 2, ntaps, ntaps);
  const auto h = taps(ntaps), af = taps(2 * ntaps);
  int irx = 0;
  OK(pysdr_rx_add(c, PYSDR_AM, -1000.0, h.data(), af.data(), 0.0, &irx));
  OK(pysdr_rx_add(c, PYSDR_NFM, 2000.0, h.data(), af.data(), 0.0, &irx));
  std::atomic<bool> stop{false};
  std::thread knobs([&] {
    int k = 0;
    while (!stop.load()) {
      double fa;
      OK(pysdr_set_lo(c, k & 1, -1000.0 - k, &fa));
      OK(pysdr_set_dec_taps(c, k & 1, h.data(), ntaps));
      OK(pysdr_set_mode(c, 0, (k & 2) ? PYSDR_AM_SYNCH : PYSDR_USB, af.data(), ntaps, 0.0));
      OK(pysdr_reset(c, 1, 3));
      OK(pysdr_set_agc(c, 0, k & 1, 0.5f));
      OK(pysdr_set_squelch(c, 1, (k & 1) ? 0.1f : 0.0f));
      pysdr_agc_state st;
      OK(pysdr_agc_get(c, 0, &st));
      ++k;
    }
  });
  std::vector<float> x(2 * (size_t)r.in_chunk * 2, 0.3f);
  const int cap = 2 * 1024 + 16;
  std::vector<float> am0(2 * cap), iq0(2 * cap), am1(2 * cap), iq1(2 * cap);
  for (int it = 0; it < 300; ++it) {
    pysdr_out outs[2] = {{am0.data(), iq0.data(), cap, 0, 0, 0.f}, {am1.data(), iq1.data(), cap, 0, 0, 0.f}};
    OK(pysdr_process(c, x.data(), (size_t)r.in_chunk - (it % 5), outs));
  }
  stop.store(true);
  knobs.join();
  pysdr_destroy(c);
}

int main(int argc, char** argv) {
  if (argc > 1 && std::strcmp(argv[1], "race") == 0) {
    race();
    std::puts("HOST_SAN_RACE_OK");
    return 0;
  }
  int ndev = 0;
  OK(pysdr_device_count(&ndev));
  FAILS(pysdr_create(nullptr, nullptr));
  FAILS(pysdr_set_overlap(nullptr, 1));
  for (int pass = 0; pass < 3; ++pass) {
  g_overlap = pass == 0 ? 0 : (pass == 1 ? 2 : 1);
  for (const Rate& r : kRates) {
    if (r.fs == 10e6) continue;
    narrowband(r, 255);
  }
  narrowband(kRates[1], 1001);                       // the reference's default prototype at the am.py rate
  single_rx_long_prototype(kRates[1]);               // 3/128
  single_rx_long_prototype(kRates[4]);               // 3/64
  single_rx_long_prototype(Rate{2.56e6, 3, 160, 54613});
  single_rx_long_prototype(Rate{1.792e6, 3, 112, 38229});
  single_rx_long_prototype(Rate{1.536e6, 1, 32, 32768});   // 12-wave shapes: all 1001 taps in one branch
  single_rx_long_prototype(Rate{1.92e6, 1, 40, 40960});
  narrowband(kRates[0], 1001);
  {
    const int before = pysdr::g_mfma_launches;
    broadcast_fm();                                  // its 10 MS/s / 40 IF decimator is the second matrix-core shape
    const char* tun = getenv("PYSDR_TUNING");
    const char* off = getenv("PYSDR_MIXDEC_MFMA");
    const bool expect = !(tun && atoi(tun) > 0 && off && atoi(off) == 0);
    if ((pysdr::g_mfma_launches > before) != expect) { std::fprintf(stderr, "broadcast FM: matrix-core launches %d\n", pysdr::g_mfma_launches - before); return 1; }
  }
  mixed_modes_are_refused();
  spectrum();
  }
  std::puts("HOST_SAN_OK");
  return 0;
}
