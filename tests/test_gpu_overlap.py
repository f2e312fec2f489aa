"""pysdr_set_overlap: the audio-rate half of a call on a second HIP stream beside the mix + decimate of the next call
(include/pysdr_hip.h; the reference's analogue is MP_SCHEME 2/3 running the demodulators beside the acquisition,
receiver.py:726-739).  The overlapped form changes WHEN kernels run, never what they compute: every test here holds a
context in that form against one that runs the same calls on a single stream, BIT FOR BIT -- audio, baseband IQ, raw
peaks, per-chunk counts, AGC / squelch / PLL state -- with calls queued back to back (nothing fetched in between, so
three calls are in flight on the two streams), with a fetch after every call, through mode changes that switch the
form between two calls, and through the ingest ring, which runs its context single-stream."""
import ctypes as C

import numpy as np
import pytest

from oracle import sdr_oracle as so
from oracle import wfm_oracle as wo
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams

pytestmark = pytest.mark.gpu


def _narrow(cfg, B, form):
    from tests.test_gpu_parity import make_gpu_receivers
    P, g = make_gpu_receivers(cfg, max_batch_chunks=B, overlap_calls=False)
    ctx = P._pysdr_stream
    _lib.check(_lib.lib().pysdr_set_overlap(ctx.h, form), "set_overlap")
    return P, g, ctx


def _state(g):
    return [(rx.agc.gain, rx.agc.maxbuf, rx.agc.agc, rx.agc.err) for rx in g]


def _fetch_all(ctx, nrx, B):
    out = []
    for i in range(nrx):
        am, iq, cn, pk = ctx.fetch(i, B)
        out.append((am.copy(), iq.copy(), cn.copy(), pk.copy()))
    return out


def _same(a, b):
    assert len(a) == len(b)
    for (am1, iq1, cn1, pk1), (am2, iq2, cn2, pk2) in zip(a, b):
        assert np.array_equal(cn1, cn2) and np.array_equal(pk1, pk2)
        assert np.array_equal(iq1, iq2)
        assert np.array_equal(am1, am2)


@pytest.mark.parametrize("fetch_each", [False, True])
def test_four_rx_overlapped_calls_equal_single_stream_bit_for_bit(fetch_each):
    """C3's four sub-receivers (USB / CW / NBFM / AM), six calls of five chunks.  form 2 overlaps EVERY call (the
    default form 1 would leave these modes single-stream: their AF FIR cannot co-reside with the front end)."""
    cfg = so.CONFIGS['C3']
    L, B, K = so.chunk_sizes(cfg['fs'], 48e3)[3], 5, 6
    x = so.synth_iq(cfg, K * B * L, 41)
    res = {}
    for form in (0, 2):
        P, g, ctx = _narrow(cfg, B, form)
        outs = []
        for k in range(K):
            ctx.process_batch(x[k * B * L:(k + 1) * B * L], B, L, on_device=False)
            assert _lib.lib().pysdr_last_call_overlapped(ctx.h) == (1 if form else 0)
            if fetch_each or k == K - 1:
                outs.append(_fetch_all(ctx, len(g), B))
        res[form] = (outs, _state(g))
        ctx.close()
    for a, b in zip(res[0][0], res[2][0]):
        _same(a, b)
    assert res[0][1] == res[2][1]


def test_am_synch_default_form_overlaps_and_equals_single_stream():
    """One RX in AM-Synch, the am.py rate with the reference's default 1001-tap prototype (matrix-core front end): the
    DEFAULT form (1) overlaps these calls -- the carrier loop's segment walks run beside the next mix + decimate.  Then
    the mode goes to AM (the form falls back to one stream between two calls, both streams drained, the stream's history
    carried over in whichever buffer of the pair was current) and back."""
    cfg = dict(so.CONFIGS['C1'])
    cfg['rx'] = [dict(frq=100e3 - 3.0, mode='AM-Synch', video_bw=10e3, af_bw=5e3)]
    L, B, K = so.chunk_sizes(cfg['fs'], 48e3)[3], 16, 7
    x = so.synth_iq(cfg, K * B * L, 42)
    modes = ['AM-Synch', 'AM-Synch', 'AM-Synch', 'AM', 'AM', 'AM-Synch', 'AM-Synch']
    res = {}
    for form in (0, 1):
        P, g, ctx = _narrow(cfg, B, form)
        outs, forms = [], []
        for k in range(K):
            g[0].mode = modes[k]
            ctx.process_batch(x[k * B * L:(k + 1) * B * L], B, L, on_device=False)
            forms.append(_lib.lib().pysdr_last_call_overlapped(ctx.h))
            if k in (2, 4, 6):
                outs.append(_fetch_all(ctx, 1, B))
        res[form] = (outs, _state(g), forms)
        ctx.close()
    assert res[0][2] == [0] * K
    assert res[1][2] == [1 if m == 'AM-Synch' else 0 for m in modes]
    for a, b in zip(res[0][0], res[1][0]):
        _same(a, b)
    assert res[0][1] == res[1][1]


def test_wfm2_overlapped_calls_equal_single_stream():
    """Broadcast FM stereo, ONE continuous stream over three calls of 24 chunks (63 pilot-loop segments each; the IF decimator of call k + 1 beside
    discriminator, pilot loop, audio resampler and AF stage of call k; the pilot loop's mean-increment guess and the
    1-sample IF history travel from call to call through the other buffer of each pair)."""
    L, B, K = 213333, 24, 3
    x = wo.synth_wfm(10e6, K * B * L, 7)
    res = {}
    for form in (0, 1):
        P = RunTimeParams(fs=10e6, fc=[98.1e6], mode='WFM2', nfilt=255, foffset=300e3, vid_bw=200e3,
                          max_batch_chunks=B, overlap_calls=False)
        g = sig_proc.Receiver(P, 300e3, 0, '1')
        ctx = P._pysdr_stream
        _lib.check(_lib.lib().pysdr_set_overlap(ctx.h, form), "set_overlap")
        outs = []
        for k in range(K):
            ctx.process_batch(x[k * B * L:(k + 1) * B * L], B, L, on_device=False)
            assert _lib.lib().pysdr_last_call_overlapped(ctx.h) == form
            if k >= 1:
                outs.append(_fetch_all(ctx, 1, B))
        seg, pat = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().pysdr_pll_stats(ctx.h, 0, C.byref(seg), C.byref(pat)), "pll_stats")
        res[form] = (outs, (seg.value, pat.value))
        ctx.close()
    for a, b in zip(res[0][0], res[1][0]):
        _same(a, b)
    assert res[0][1] == res[1][1] and res[0][1][0] > 16


def test_batch_contexts_of_the_facade_default_to_form_one_and_a_ring_switches_it_off():
    cfg = so.CONFIGS['C2']
    from tests.test_gpu_parity import make_gpu_receivers
    P, g = make_gpu_receivers(cfg, max_batch_chunks=4)
    ctx = P._pysdr_stream
    lib = _lib.lib()
    assert lib.pysdr_get_overlap(ctx.h) == 1
    P1, g1 = make_gpu_receivers(cfg)                       # a live, one-chunk context: nothing to overlap with
    assert lib.pysdr_get_overlap(P1._pysdr_stream.h) == 0
    from pysdr_amd.ingest import IngestRing
    ring = IngestRing(ctx, 3, 2)
    assert lib.pysdr_get_overlap(ctx.h) == 0
    assert lib.pysdr_set_overlap(ctx.h, 1) < 0 and b"ingest" in lib.pysdr_last_error()
    ring.close()
    assert lib.pysdr_set_overlap(ctx.h, 1) == 0
    assert lib.pysdr_set_overlap(ctx.h, 3) < 0
