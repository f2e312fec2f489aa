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


def test_run_loop_equals_direct_oracle_with_dc_removal_and_gain():
    cfg = so.CONFIGS['C1']
    nchunks = 6
    P = make_P(cfg, nchunks)
    L = P.IN_CHUNK_SIZE
    P.sdr = stream.SynthSDR(cfg, seed=22, nsamp=(nchunks + 1) * L)
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    ex.Run()
    got = P.players[0].rb.pull(P.players[0].rb.nsamps)
    ref = so.Receiver(P.SRATE, P.FS_OUT, P.rx_offset(0), mode='AM', ntaps_dec=P.FILT_LEN,
                      ntaps_af=P.AF_FILT_LEN, video_bw=P.VIDEO_BW, af_bw=P.AF_BW, dtype=np.float32)
    want = []
    for k in range(nchunks):
        am = ref.demod_data(P.sdr.samples[k * L:(k + 1) * L])
        am = am - np.mean(am)                                  # receiver.py:250-252
        want.append(am * (pow(10., P.AF_GAIN) - 1))            # receiver.py:200,212
    assert np.allclose(got, np.concatenate(want), rtol=0, atol=1e-7)


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
