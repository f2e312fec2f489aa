"""Host logic of the executive (receiver.py:538-631 chunk assembly with short reads and the
xold carry, :250-252 DC removal, :153-225 audio routing) and the ring buffers, on CPU.
The DSP behind it is the oracle (tests/oracle_dsp.py)."""
import numpy as np
import pytest

from oracle import sdr_oracle as so
from pysdr_amd import executive, rates, stream
from pysdr_amd.params import RunTimeParams
from pysdr_amd.sig_proc import ring_buffer2, ring_buffer3
from tests import oracle_dsp


def make_P(cfg, nchunks, **kw):
    r0 = cfg['rx'][0]
    P = RunTimeParams(fs=cfg['fs'], fsout=cfg['fs_out'], fc=[7.1e6] * len(cfg['rx']), mode=r0['mode'],
                      foffset=r0['frq'], nfilt=cfg['ntaps_dec'], vid_bw=r0.get('video_bw', 10e3),
                      af_bw=r0.get('af_bw', 0.0), **kw)
    P.DURATION = nchunks * P.IN_CHUNK_SIZE / P.SRATE - 1e-9      # receiver.py:689,764
    return P


def test_params_arithmetic_matches_reference_rules():
    P = RunTimeParams(fs=8e6, fc=[14.074e6, 14.08e6, 14.1e6], mode='USB', foffset=0.0)
    assert (P.UP, P.DOWN, P.FS_OUT, P.IN_CHUNK_SIZE) == (3, 500, 48000, 170666)   # params.py:405-444
    assert P.RB_SIZE == 32 * 1024 * 4                                              # NUM_RX > 2
    m = P.FOFFSET * P.RB_SIZE / P.SRATE
    assert abs(m - round(m)) < 1e-9                                                # utils.py:277-289
    assert P.MUTE_CHUNKS == int(.25 * 48000 / 1024)
    assert rates.up_dn(10e6, 192e3) == (12, 625)
    assert RunTimeParams(mode='CW').BFO == 700                                     # params.py:319-320
    assert RunTimeParams(mode='WFM').VIDEO_BW == 200e3 and RunTimeParams(mode='AM').VIDEO_BW == 10e3


def test_short_reads_lose_and_duplicate_nothing():
    cfg = so.CONFIGS['C1']
    nchunks = 5
    P = make_P(cfg, nchunks)
    L = P.IN_CHUNK_SIZE
    P.sdr = stream.SynthSDR(cfg, seed=21, nsamp=nchunks * L + 1234)
    seen = []
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    ex.Run(on_chunk=lambda e: seen.append(e.x.copy()))
    assert len(seen) == nchunks
    assert np.array_equal(np.concatenate(seen), P.sdr.samples[:nchunks * L])
    assert P.sdr.ncall > nchunks            # the reads really were short
    assert not P.sdr.active                 # quit_rx closed the stream (receiver.py:483-484)


def test_run_loop_equals_direct_oracle_with_gain_and_the_dc_removed_psd_tap():
    cfg = so.CONFIGS['C1']
    nchunks = 6
    P = make_P(cfg, nchunks)
    L = P.IN_CHUNK_SIZE
    P.sdr = stream.SynthSDR(cfg, seed=22, nsamp=(nchunks + 1) * L)
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    P.SHOW_AF_PSD, P.PLOT_RX, P.rb_af = True, 0, ring_buffer2('AF', 1 << 20)
    ex.Run()
    got = P.players[0].rb.pull(P.players[0].rb.nsamps)
    tap = P.rb_af.pull(P.rb_af.nsamps)
    ref = so.Receiver(P.SRATE, P.FS_OUT, P.rx_offset(0), mode='AM', ntaps_dec=P.FILT_LEN,
                      ntaps_af=P.AF_FILT_LEN, video_bw=P.VIDEO_BW, af_bw=P.AF_BW, dtype=np.float32)
    want, want_tap = [], []
    for k in range(nchunks):
        am = ref.demod_data(P.sdr.samples[k * L:(k + 1) * L])
        want.append(am * (pow(10., P.AF_GAIN) - 1))            # receiver.py:195-200,212: rx.am, slider gain
        want_tap.append(am - np.mean(am))                      # receiver.py:250-252,263: the AF PSD tap is DC-free
    assert np.allclose(got, np.concatenate(want), rtol=0, atol=1e-7)
    assert np.allclose(tap, np.concatenate(want_tap), rtol=0, atol=1e-7)


def test_replay_mode_and_stereo_audio_scheme():
    cfg = so.CONFIGS['C3']
    nchunks = 3
    P = make_P(cfg, nchunks, audio=2)
    L = P.IN_CHUNK_SIZE
    P.REPLAY_MODE = True
    P.sdr = stream.SynthSDR(cfg, seed=23, nsamp=(nchunks + 1) * L)
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    for i, r in enumerate(cfg['rx']):
        P.rx[i].mode, P.rx[i].af_bw, P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
    ex.Run()
    assert P.NUM_PLAYERS == 2
    st = P.players[0].rb.pull(P.players[0].rb.nsamps)           # rx0 + 1j*rx2 (receiver.py:185)
    assert np.iscomplexobj(st) and len(st) >= nchunks * 1023


def test_ring_buffer2_semantics():
    rb = ring_buffer2('t', 10000)
    assert rb.pull(10) == [] and not rb.ready(1)
    rb.push(np.arange(100, dtype=np.float32))
    rb.push(np.arange(100, 250, dtype=np.float32))
    assert rb.nsamps == 250 and rb.ready(250) and rb.buf.qsize() == 2
    assert np.array_equal(rb.pull(30), np.arange(30))
    assert np.array_equal(rb.pull(100), np.arange(30, 130))     # spans two pushed blocks
    assert np.array_equal(rb.pull(50, True), np.arange(200, 250))   # flush drops the backlog
    assert rb.nsamps == 0
    rb.push_zeros(16)
    assert rb.nsamps == 16 and np.all(rb.pull(16) == 0)
    small = ring_buffer2('s', 100)
    assert small.push(np.zeros(80)) and not small.push(np.zeros(80))   # overflow prevented
    small.clear()
    assert small.nsamps == 0 and small.tag == 's' and small.size == 100
    z = small.push(np.ones(4) + 1j)
    assert z and np.iscomplexobj(small.pull(4))


def test_ring_buffer3_queue_backed():
    rb = ring_buffer3('q', 1000)
    rb.push(np.arange(10.0))
    rb.push(np.arange(10.0, 30.0))
    import time
    time.sleep(0.05)
    assert rb.ready(30)
    assert np.array_equal(rb.pull(15), np.arange(15.0))
    assert np.array_equal(rb.pull(15), np.arange(15.0, 30.0))
    assert rb.pull(1) == []


class _ScriptedSDR(stream.SynthSDR):
    """SynthSDR with an explicit per-call sample count (the schedule of tests/golden/read_chunk_ref.npz)."""

    def __init__(self, samples, schedule):
        super().__init__(samples)
        self.schedule = [int(v) for v in schedule]

    def readStream(self, stream_, buffs, n, timeoutUs=100000):
        want = self.schedule[self.ncall % len(self.schedule)]
        self.ncall += 1
        k = min(want, n, len(self.samples) - self.pos)
        if k <= 0:
            return stream.StreamResult(0)
        buffs[0][:k] = self.samples[self.pos:self.pos + k]
        self.pos += k
        return stream.StreamResult(k)


def _chunk_P(L, sdr, replay=False):
    P = RunTimeParams(fs=64e3, fc=[7.1e6], mode='AM', nfilt=63)
    P.IN_CHUNK_SIZE = L
    P.sdr, P.REPLAY_MODE = sdr, replay
    return P


def test_read_chunk_equals_the_executed_reference_text():
    """tests/golden/read_chunk_ref.npz = SDR_EXECUTIVE.read_chunk of receiver.py:538-631 EXECUTED as it
    stands on a scripted SoapySDR-shaped device (short reads, zero-length reads, reads that overshoot
    the chunk; tests/golden/make_read_chunk_ref_golden.py, build container only): this build's
    executive assembles the same chunks and carries the same number of samples over (`xold`)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "read_chunk_ref.npz"))
    L = int(g["L"])
    P = _chunk_P(L, _ScriptedSDR(g["stream"], g["schedule"]))
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    ex.Startup()
    for k in range(len(g["live"])):
        ex.read_chunk()
        assert np.array_equal(ex.x, g["live"][k]), k
        assert len(ex.xold) == int(g["carry"][k]), k
    # replay leg: the strict `<` of receiver.py:543 drops the tail chunk and raises RX_DONE
    P2 = _chunk_P(L, stream.SynthSDR(g["stream"][:4 * L]), replay=True)
    ex2 = executive.SDR_EXECUTIVE(P2, dsp=oracle_dsp)
    ex2.Startup()
    for k in range(len(g["replay"])):
        ex2.read_chunk()
        assert bool(P2.RX_DONE) == bool(g["replay_done"][k]), k
        assert np.array_equal(np.asarray(ex2.x), g["replay"][k]), k


def test_demodulate_data_and_audio_out_equal_the_executed_reference_text():
    """tests/golden/host_loop_ref.npz = receiver.py:231-297 (demodulate_data) and :153-225 (audio_out)
    EXECUTED as they stand around a scripted `rx.demod_data` (tests/golden/make_host_loop_ref_golden.py,
    build container only).  Same scripts through this build's executive: the audio handed to every
    player (slider gain receiver.py:200, manual / automatic mute, scheme-2 stereo packing :185), the AF
    and baseband PSD taps, the saved files and the auto-mute state, sample for sample.  (This is what
    showed that the DC removal of :250-252 never reaches the audio: `am` is re-bound, `rx.am` is not.)"""
    import os
    import types
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "host_loop_ref.npz"))
    spec = importlib.util.spec_from_file_location("mk", os.path.join(here, "golden", "make_host_loop_ref_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)                      # the scripted rx / ring buffer / player doubles (no reference access)

    nch = g["A_am"].shape[1]
    rxs = [mk.FakeRx(list(g["A_am"][i]), list(g["A_iq"][i]), list(g["A_mutes"][i])) for i in range(2)]
    P = types.SimpleNamespace(rx=rxs, NUM_RX=2, MODE='AM', ENABLE_AUTO_MUTE=True, AUTO_MUTED=False, SHOW_AF_PSD=True, PLOT_RX=0,
                              PANADAPTOR=False, MP_SCHEME=1, rb_af=mk.Rec(), SHOW_BASEBAND_PSD=False, ENABLE_RTTY=False,
                              SAVE_BASEBAND=False, SAVE_DEMOD=True, demod_io=mk.Rec(), AUDIO_SCHEME=1,
                              players=[mk.Player(), mk.Player()], MUTED=[False, True, False, False, False, False],
                              AF_GAIN=float(g["A_af_gain"]), audio_playback=True, LOOPBACK=False, AUX_AUDIO=False, DELAY=1024)
    _, trace = mk.scenario(P, rxs, nch, executive.demodulate_data, executive.audio_out)
    assert np.array_equal(np.stack(P.players[0].rb.items), g["A_player0"])
    assert np.array_equal(np.stack(P.players[1].rb.items), g["A_player1"]) and not g["A_player1"].any()   # muted by hand
    assert np.array_equal(np.stack(P.rb_af.items), g["A_rb_af"]) and np.array_equal(np.stack(P.demod_io.items), g["A_saved"])
    assert list(trace) == list(g["A_auto_muted"])
    assert [len(p.started) for p in P.players] == list(g["A_started"])

    rxs = [mk.FakeRx(list(g["B_am"][i]), list(g["B_iq"][i]), [0] * nch) for i in range(3)]
    P = types.SimpleNamespace(rx=rxs, NUM_RX=3, MODE='CW', ENABLE_AUTO_MUTE=False, AUTO_MUTED=False, SHOW_AF_PSD=False, PLOT_RX=0,
                              PANADAPTOR=False, MP_SCHEME=1, SHOW_BASEBAND_PSD=True, rb_baseband=mk.Rec(), ENABLE_RTTY=False,
                              SAVE_BASEBAND=True, baseband_iq_io=mk.Rec(), SAVE_DEMOD=False, AUDIO_SCHEME=2,
                              players=[mk.Player(), mk.Player()], MUTED=[False, False, True, False, False, False],
                              AF_GAIN=float(g["B_af_gain"]), audio_playback=True, LOOPBACK=False, AUX_AUDIO=False, DELAY=2048)
    mk.scenario(P, rxs, nch, executive.demodulate_data, executive.audio_out)
    assert np.array_equal(np.stack(P.players[0].rb.items), g["B_player0"])
    assert np.array_equal(np.stack(P.players[1].rb.items), g["B_player1"])
    assert np.array_equal(np.stack(P.rb_baseband.items), g["B_rb_baseband"])
    assert np.array_equal(np.stack(P.baseband_iq_io.items), g["B_saved"])
