"""Drop-in for the ``sig_proc`` module pySDR imports (``receiver.py:39,45``:
``from sig_proc import up_dn`` / ``import sig_proc as dsp``), backed by hand-written HIP
kernels on MI355X through the C ABI in ``include/pysdr_hip.h``.

Surface: ``up_dn``, ``Receiver``, ``signal_generator``, ``spectrum``,
``ring_buffer2``, ``ring_buffer3``, ``convolver``, ``bpf``.  Names, argument meaning and
the attributes the reference's callers touch are kept; there is no CPU arithmetic path --
every DSP call goes to the GPU and raises ``PysdrError`` if it cannot.

All sub-receivers built from the same ``P`` share ONE device context, so the wideband
chunk is uploaded and read once for every RX (the reference loops
``for irx: rx.demod_data(x)`` over the same ``x``, ``receiver.py:724-725``).
"""
from __future__ import annotations

import ctypes as C
import sys
import multiprocessing as mp
import queue
import threading

import numpy as np

from . import _lib, design
from ._lib import PysdrError, check
from .rates import up_dn  # noqa: F401  (re-exported: ``from sig_proc import up_dn``)
from .tables import AF_BWs as _AF_BWs
from .tables import MODE_INDEX, VIDEO_BWs as _VIDEO_BWs
from .tables import index_of_bw, label_hz

bpf = design.bpf

_WFM_MODES = ("WFM", "WFM2")


def _device_of(P):
    return int(getattr(P, 'GPU_DEVICE', 0) or 0)


def _chunk_key(x):
    """Identity of a chunk for the per-chunk cache of ``Receiver.demod_data``: how long it is and 32 of its samples.  NOT where
    it lives: a buffer reused in place for the next chunk keeps its address while its contents change, and a caller that hands
    every sub-receiver its own copy of the chunk (a slice of an ``np.load`` archive is a new array on every access) has one
    chunk at several addresses -- with the address in the key such a caller ran the stream once per sub-receiver, or not,
    as the allocator happened to reuse memory (found by tests/test_golden.py in round 6)."""
    x = np.asarray(x)
    n = x.shape[0] if x.ndim else 0
    if n == 0:
        return (0, b'')
    idx = (np.arange(32, dtype=np.int64) * (n - 1)) // 31
    return (n, x[idx].tobytes())


# ----------------------------------------------------------------------------------------
class _StreamContext:
    """One wideband stream on one GPU: owns the ``pysdr_ctx`` and the per-chunk cache
    that lets N ``Receiver.demod_data(x)`` calls share one launch sequence."""

    def __init__(self, P):
        L = _lib.lib()
        _lib.require_gpu()
        self.L = L
        self.ntaps_dec = int(getattr(P, 'FILT_LEN', 1001))
        self.ntaps_af = int(getattr(P, 'AF_FILT_LEN', 255))
        self.max_chunks = int(getattr(P, 'MAX_BATCH_CHUNKS', 1))
        self.cfg = _lib.Cfg(float(P.SRATE), int(P.UP), int(P.DOWN), int(P.IN_CHUNK_SIZE),
                            self.max_chunks, self.ntaps_dec, self.ntaps_af, _device_of(P), 0)
        self.fs_out = int(P.FS_OUT)
        h = C.c_void_p()
        check(L.pysdr_create(C.byref(self.cfg), C.byref(h)), "pysdr_create")
        self.h = h
        # A context sized for batches (replay, the benchmark) runs the audio-rate half of a call beside the mix +
        # decimate of the next one (pysdr_set_overlap: two HIP streams, results unchanged); a live one-chunk context
        # has nothing to overlap with -- every call is fetched before the next one exists
        if self.max_chunks > 1 and bool(getattr(P, 'OVERLAP_CALLS', True)):
            check(L.pysdr_set_overlap(self.h, 1), "pysdr_set_overlap")
        self.receivers = []
        self.seq = 0                 # chunks processed
        self.cache = {}              # irx -> (am, iq, peak)
        self.cache_key = None        # which chunk the cache holds (_chunk_key) ...
        self.served = set()          # ... and which sub-receivers have taken their share of it
        self.lock = threading.Lock()
        self.cap = int(self.max_chunks * int(P.IN_CHUNK_SIZE) * int(P.UP) // int(P.DOWN)) + 8

    def close(self):
        if self.h:
            # dependants hold raw pointers into the context: they go first
            for ring in list(getattr(self, '_rings', [])):
                ring.close()
            self.L.pysdr_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be torn down: leave the device
        # memory to process exit rather than call into a dead runtime
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def add(self, rx, mode, lo_freq, h, af, bfo):
        irx = C.c_int(-1)
        h = np.ascontiguousarray(h, np.float64)
        afi = np.ascontiguousarray(np.asarray(af, np.complex128)).view(np.float64)
        check(self.L.pysdr_rx_add(self.h, MODE_INDEX[mode], float(lo_freq), _lib.as_pd(h),
                                  _lib.as_pd(afi), float(bfo), C.byref(irx)), "pysdr_rx_add")
        self.receivers.append(rx)
        return irx.value

    def process_chunk(self, x):
        """rx.demod_data(x) for every RX of the stream (``receiver.py:231-235``)."""
        x = np.ascontiguousarray(x, np.complex64)
        n = len(x)
        nrx = len(self.receivers)
        outs = (_lib.Out * nrx)()
        cap = int(n * self.cfg.up // self.cfg.down) + 8
        bufs = []
        for r in range(nrx):
            am = np.empty(2 * cap, np.float32)
            iq = np.empty(2 * cap, np.float32)
            bufs.append((am, iq))
            outs[r].am = _lib.as_pf(am)
            outs[r].iq = _lib.as_pf(iq)
            outs[r].cap = cap
        check(self.L.pysdr_process(self.h, _lib.as_pf(x.view(np.float32)), n, outs),
              "pysdr_process")
        for r in range(nrx):
            k = outs[r].n_out
            am, iq = bufs[r]
            a = am[:2 * k].view(np.complex64) if outs[r].am_is_complex else am[:k]
            self.cache[r] = (a, iq[:2 * k].view(np.complex64), float(outs[r].peak_in))
        self.seq += 1

    # batch (device- or host-resident) path used by replay and the benchmark
    def process_batch(self, iq, nchunks, chunk_len, on_device=False):
        for rx in self.receivers:
            rx._sync_controls()
        if on_device:
            ptr = C.c_void_p(int(iq))
        else:
            iq = np.ascontiguousarray(iq, np.complex64)
            ptr = C.c_void_p(iq.ctypes.data)
        check(self.L.pysdr_process_batch(self.h, ptr, int(nchunks), int(chunk_len),
                                         1 if on_device else 0), "pysdr_process_batch")
        self.seq += nchunks

    def fetch(self, irx, nchunks, want_iq=True):
        cap = self.cap
        am = np.empty(2 * cap, np.float32)
        iq = np.empty(2 * cap, np.float32) if want_iq else None
        n, cx = C.c_int(0), C.c_int(0)
        cn = np.zeros(nchunks, np.int32)
        pk = np.zeros(nchunks, np.float32)
        check(self.L.pysdr_fetch(self.h, irx, _lib.as_pf(am), _lib.as_pf(iq) if want_iq else None,
                                 cap, C.byref(n), C.byref(cx),
                                 cn.ctypes.data_as(C.POINTER(C.c_int)), _lib.as_pf(pk)), "pysdr_fetch")
        k = n.value
        a = am[:2 * k].view(np.complex64) if cx.value else am[:k]
        q = iq[:2 * k].view(np.complex64) if want_iq else None
        return a, q, cn, pk


def _context_for(P):
    ctx = getattr(P, '_pysdr_stream', None)
    if ctx is None or ctx.h is None:
        ctx = _StreamContext(P)
        P._pysdr_stream = ctx
    return ctx


# ----------------------------------------------------------------------------------------
class signal_generator:
    """``dsp.signal_generator(f, N, fs, complex_flag)`` (``receiver.py:822``): complex NCO
    with a persistent 32-bit phase accumulator.  ``quad_mixer(x) = x*exp(+j*phi_n)``
    (``receiver.py:552-553``); ``change_freq(f)`` returns the frequency really generated
    (used as the new FOFFSET, ``gui.py:1928``)."""

    def __init__(self, f, N, fs, complex_flag=True, device=0):
        self.N = int(N)
        self.fs = float(fs)
        self.complex_flag = complex_flag
        self.device = device
        self.phase = 0
        self.fword = 0
        self.fo = 0.0
        self.change_freq(f)

    def change_freq(self, f):
        act = C.c_double(0.0)
        self.fword = int(_lib.lib().pysdr_freq_word(float(f), self.fs, C.byref(act)))
        self.fo = act.value
        return self.fo

    def quad_mixer(self, x):
        _lib.require_gpu()
        x = np.ascontiguousarray(x, np.complex64)
        y = np.empty_like(x)
        ph = C.c_uint32(0)
        check(_lib.lib().pysdr_quad_mixer(self.device, _lib.as_pf(x.view(np.float32)),
                                          _lib.as_pf(y.view(np.float32)), len(x),
                                          self.phase, self.fword, C.byref(ph)), "pysdr_quad_mixer")
        self.phase = ph.value
        return y


class _ReceiverLO:
    """``rx.lo``: ``change_freq(f)`` retunes the sub-receiver (``receiver.py:112,352``;
    ``gui.py:1938,2011``).  Takes effect at the next chunk."""

    def __init__(self, rx, f):
        self._rx = rx
        self.fo = 0.0
        self.fs = rx._ctx.cfg.srate

    def change_freq(self, f):
        act = C.c_double(0.0)
        check(self._rx._ctx.L.pysdr_set_lo(self._rx._ctx.h, self._rx.irx, float(f), C.byref(act)),
              "pysdr_set_lo")
        self.fo = act.value
        return self.fo


class _Decimator:
    """``rx.dec``: ``filter_bank[len(VIDEO_BWs)]`` of prototypes and the live-swappable
    ``h`` (``rx.dec.h = rx.dec.filter_bank[idx]``, ``receiver.py:127,371``; ``gui.py:1713``)."""

    def __init__(self, rx, bank, idx):
        self._rx = rx
        self.filter_bank = bank
        self._h = bank[idx]

    @property
    def h(self):
        return self._h

    @h.setter
    def h(self, taps):
        taps = np.ascontiguousarray(taps, np.float64)
        check(self._rx._ctx.L.pysdr_set_dec_taps(self._rx._ctx.h, self._rx.irx, _lib.as_pd(taps),
                                                 len(taps)), "pysdr_set_dec_taps")
        self._h = taps


class _PLLHandle:
    def __init__(self, rx):
        self._rx = rx

    def reset(self):
        """``rx.demod.am_pll.reset()`` (``receiver.py:649``)."""
        check(self._rx._ctx.L.pysdr_reset(self._rx._ctx.h, self._rx.irx, 2), "pysdr_reset")


class _WfmVideo:
    """``rx.demod.wfm_video``: the pre-detection filter of the broadcast-FM path;
    ``wfm_video.h = wfm_filter_bank[idx]`` (``gui.py:1704``) swaps it live."""

    def __init__(self, rx, h):
        self._rx = rx
        self._h = np.ascontiguousarray(h, np.float64)

    @property
    def h(self):
        return self._h

    @h.setter
    def h(self, taps):
        self._h = np.ascontiguousarray(taps, np.float64)
        self._rx._push_wfm_taps()


class _Demod:
    """``rx.demod``: AF filter banks (``receiver.py:873-874``), the AM-Synch PLL and the
    broadcast-FM video filter bank (``gui.py:1704``)."""

    def __init__(self, rx, fs_out, ntaps):
        self.filter_bank_real = design.af_bank_real(fs_out, ntaps)
        self.filter_bank_cmpx = design.af_bank_cmpx(fs_out, ntaps)
        self.am_pll = _PLLHandle(rx)
        ctx = rx._ctx
        d1, up2, down2 = C.c_int(0), C.c_int(0), C.c_int(0)
        check(ctx.L.pysdr_wfm_params(float(ctx.cfg.srate), float(fs_out), C.byref(d1), C.byref(up2),
                                     C.byref(down2)), "pysdr_wfm_params")
        self.wfm_d1, self.wfm_up2, self.wfm_down2 = d1.value, up2.value, down2.value
        self.wfm_fs1 = ctx.cfg.srate / d1.value
        vbw = float(getattr(rx.P, 'VIDEO_BW', 200e3) or 200e3)
        self.wfm_filter_bank = design.wfm_video_bank(ctx.cfg.srate, self.wfm_fs1, ctx.ntaps_dec, vbw,
                                                     rx._video_labels)
        vidx = index_of_bw(vbw, rx._video_labels, len(rx._video_labels) - 1)
        self.wfm_video = _WfmVideo(rx, self.wfm_filter_bank[vidx])
        self.wfm_resamp = design.wfm_resampler_taps(self.wfm_fs1, self.wfm_up2)


class _AGC:
    """``rx.agc``: ``reset()`` (``receiver.py:648``) and the fields the watchdog prints
    (``watchdog.py:298-302``)."""

    def __init__(self, rx):
        self._rx = rx

    def reset(self):
        check(self._rx._ctx.L.pysdr_reset(self._rx._ctx.h, self._rx.irx, 1), "pysdr_reset")

    def _get(self):
        st = _lib.AgcState()
        check(self._rx._ctx.L.pysdr_agc_get(self._rx._ctx.h, self._rx.irx, C.byref(st)),
              "pysdr_agc_get")
        return st

    agc = property(lambda s: s._get().agc)
    gain = property(lambda s: s._get().gain)
    maxbuf = property(lambda s: s._get().maxbuf)
    ref = property(lambda s: s._get().ref)
    err = property(lambda s: s._get().err)


class Receiver:
    """``dsp.Receiver(P, frq, irx, name, VIDEO_BWs, AF_BWs)`` (``receiver.py:65,835``).

    ``frq`` = offset (Hz) of the wanted signal from the SDR centre frequency
    (``foff + FC[irx] - FC[0]``, ``receiver.py:831-834``).  ``demod_data(x)`` runs
    LO mix -> rational resample -> detector -> AF filter -> AGC on the GPU and sets
    ``.am`` / ``.iq`` (``receiver.py:235,265``).  ``P.MODE``, ``P.AF_BW``,
    ``P.AF_FILTER_NUM`` and ``P.BFO`` are re-read every chunk, as the reference's worker
    loop does (``receiver.py:114-116,130-131``); ``rx.mode`` / ``rx.af_bw`` / ``rx.bfo``
    override them per sub-receiver (config "4 independent RX")."""

    def __init__(self, P, frq, irx, name, VIDEO_BWs=_VIDEO_BWs, AF_BWs=_AF_BWs):
        self.P = P
        self.name = name
        self.sub = 0
        self.mode = None
        self.af_bw = None
        self.bfo = None
        self._ctx = _context_for(P)
        ctx = self._ctx
        self._video_labels = list(VIDEO_BWs)
        self._af_labels = list(AF_BWs)
        video_bw = float(getattr(P, 'VIDEO_BW', 10e3) or 10e3)
        bank = design.decimator_bank(P.SRATE, int(P.UP), ctx.fs_out, ctx.ntaps_dec, video_bw,
                                     self._video_labels)
        vidx = getattr(P, 'VIDEO_FILTER_NUM', None)
        if vidx is None:
            vidx = index_of_bw(video_bw, self._video_labels, len(self._video_labels) - 1)
        self.demod = _Demod(self, ctx.fs_out, ctx.ntaps_af)
        self._applied = self._want()
        af = self._af_taps(*self._applied)
        self.irx = ctx.add(self, self._applied[0], -float(frq), bank[vidx], af, self._applied[3])
        self.lo = _ReceiverLO(self, -float(frq))
        self.lo.fo = -float(frq)
        self.dec = _Decimator(self, bank, vidx)
        self.agc = _AGC(self)
        self.am = np.zeros(0, np.float32)
        self.iq = np.zeros(0, np.complex64)
        self.peak_in = 0.0
        self._seen = ctx.seq
        self._mute_left = 0
        self._squelch = 0.0

    # -- what the controls currently ask for: (mode, af_idx, af_bw, bfo, lsb)
    def _want(self):
        P = self.P
        mode = self.mode if self.mode is not None else P.MODE
        if mode == 'FM':
            mode = 'NFM'
        af_bw = self.af_bw if self.af_bw is not None else float(getattr(P, 'AF_BW', 0) or 0)
        idx = getattr(P, 'AF_FILTER_NUM', None) if self.af_bw is None else None
        if idx is None:
            idx = index_of_bw(af_bw, self._af_labels, 0) if af_bw else 0
        bfo = self.bfo if self.bfo is not None else float(getattr(P, 'BFO', 0) or 0)
        if mode == 'CW' and bfo == 0:
            bfo = 700.0                           # params.py:319-320
        fc = getattr(P, 'FC', None)
        lsb = bool(mode == 'SSB' and fc is not None and len(fc) and float(fc[0]) < 10e6)
        return (mode, int(idx), af_bw, bfo, lsb)

    def _af_taps(self, mode, idx, af_bw, bfo, lsb):
        if mode in _WFM_MODES:
            return design.wfm_af_taps(self._ctx.fs_out, self._ctx.ntaps_af, af_bw).astype(np.complex128)
        if mode not in MODE_INDEX:
            raise ValueError(f"unknown mode {mode}")
        if mode == 'CW':
            bw = af_bw if af_bw else (label_hz(self._af_labels[idx]) or 0.0)
            return design.cw_taps(self._ctx.fs_out, self._ctx.ntaps_af, bw, bfo)
        if mode in ('AM', 'AM-Synch', 'NFM', 'IQ'):
            return self.demod.filter_bank_real[idx].astype(np.complex128)
        c = self.demod.filter_bank_cmpx[idx]
        return np.conj(c) if (mode == 'LSB' or lsb) else c

    def _push_wfm_taps(self):
        v = np.ascontiguousarray(self.demod.wfm_video.h, np.float64)
        rs = np.ascontiguousarray(self.demod.wfm_resamp, np.float64)
        check(self._ctx.L.pysdr_set_wfm_taps(self._ctx.h, self.irx, _lib.as_pd(v), len(v),
                                             _lib.as_pd(rs), len(rs)), "pysdr_set_wfm_taps")
        self._wfm_pushed = True

    def _sync_controls(self):
        want = self._want()
        if want[0] in _WFM_MODES and not getattr(self, '_wfm_pushed', False):
            self._push_wfm_taps()
        if want != self._applied:
            af = np.ascontiguousarray(self._af_taps(*want), np.complex128).view(np.float64)
            check(self._ctx.L.pysdr_set_mode(self._ctx.h, self.irx, MODE_INDEX[want[0]],
                                             _lib.as_pd(af), self._ctx.ntaps_af, float(want[3])),
                  "pysdr_set_mode")
            self._applied = want

    def demod_data(self, x):
        """``rx.demod_data(x)`` (``receiver.py:235``).  The sub-receivers of a stream share ONE launch sequence per chunk: the
        first of them to be handed a chunk runs it for all, the others take their share from the cache.  WHICH chunk the
        cache holds is decided by the chunk itself (its length and a fingerprint of 32 of its samples), not by
        counting calls: a caller that leaves a sub-receiver out for a chunk (the reference's MP_SCHEME 3 workers each call
        their own ``rx``, ``receiver.py:726-739``) then neither shifts that receiver onto the previous chunk's results nor
        the others onto a chunk that was never run.  The same receiver asking twice for the same samples runs them twice
        (a stream may repeat).  What this cannot tell apart: two different chunks with the same length and
        fingerprint (all-zero chunks, a stream that repeats chunk for chunk) when a receiver skipped the first of them -- it
        then gets the first one's share."""
        ctx = self._ctx
        key = _chunk_key(x)
        with ctx.lock:
            if key != ctx.cache_key or self.irx in ctx.served or self.irx not in ctx.cache:
                # first receiver to see this chunk: run the whole stream once
                for rx in ctx.receivers:
                    rx._sync_controls()
                ctx.process_chunk(x)
                ctx.cache_key = key
                ctx.served = set()
            am, iq, pk = ctx.cache[self.irx]
            ctx.served.add(self.irx)
            self._seen = ctx.seq
        self.am, self.iq, self.peak_in = am, iq, pk
        return am

    # -- NFM noise squelch (not a call site of the reference: north_star lists it; the idea
    # is sigs/squelch.m:92-145).  ``rx.squelch = thresh`` arms it, 0 disables.
    @property
    def squelch(self):
        return self._squelch

    @squelch.setter
    def squelch(self, thresh):
        self._squelch = float(thresh)
        check(self._ctx.L.pysdr_set_squelch(self._ctx.h, self.irx, self._squelch), "pysdr_set_squelch")

    # -- the ratio squelch as sigs/squelch.m:92-145 sketches it: envelopes of the < 3 kHz and > 4 kHz parts of the discriminator
    # output (one-pole per sample, alpha = 0.001), gate open while sq1 / sq2 >= ``rx.squelch_ratio`` -- independent of the signal's
    # level.  0 disables; armed, it takes precedence over ``rx.squelch``.
    @property
    def squelch_ratio(self):
        return getattr(self, '_squelch_ratio', 0.0)

    @squelch_ratio.setter
    def squelch_ratio(self, min_ratio):
        from . import design
        self._squelch_ratio = float(min_ratio)
        lp, hp = design.squelch_ratio_taps(float(self._ctx.fs_out))
        check(self._ctx.L.pysdr_set_squelch_ratio(self._ctx.h, self.irx, self._squelch_ratio, _lib.as_pf(lp), _lib.as_pf(hp), len(lp)),
              "pysdr_set_squelch_ratio")

    @property
    def squelch_ratio_state(self):
        """(sq1, sq2, gate open) behind the last chunk."""
        lo, hi, op = C.c_float(0), C.c_float(0), C.c_int(1)
        check(self._ctx.L.pysdr_squelch_ratio_get(self._ctx.h, self.irx, C.byref(lo), C.byref(hi), C.byref(op)),
              "pysdr_squelch_ratio_get")
        return lo.value, hi.value, bool(op.value)

    @property
    def squelch_state(self):
        lvl, op = C.c_float(0), C.c_int(1)
        check(self._ctx.L.pysdr_squelch_get(self._ctx.h, self.irx, C.byref(lvl), C.byref(op)),
              "pysdr_squelch_get")
        return lvl.value, bool(op.value)

    def auto_mute(self, x=None):
        """``rx.auto_mute(x)`` (``receiver.py:238-245``): big-signal detector on the raw
        chunk, held for ``P.MUTE_CHUNKS`` chunks (``params.py:446-450``).  The peak
        |x|^2 was reduced by the mix+decimate kernel while it read the chunk."""
        thr = float(getattr(self.P, 'AUTO_MUTE_THRESH', 0.7))
        if self.peak_in > thr * thr:
            self._mute_left = int(getattr(self.P, 'MUTE_CHUNKS', 1))
            return True
        if self._mute_left > 0:
            self._mute_left -= 1
            return True
        return False


# ----------------------------------------------------------------------------------------
class spectrum:
    """``dsp.spectrum(fs_kHz, chunk_size, NFFT, overlap, TAG=)`` (``Plotting.py:376``;
    ``gui.py:611-631`` pass fs in kHz).  ``periodogram(y, True)`` slides ``len(y)`` new
    samples into the ``chunk_size`` window, then window -> zero-pad -> FFT (rocFFT) ->
    ``10*log10(re^2+im^2)`` -> fftshift on the GPU (formula: ``rtty.py:839-841``)."""

    def __init__(self, fs, chunk_size, NFFT, overlap, TAG='', device=0):
        _lib.require_gpu()
        self.fs = float(fs)
        self.chunk_size = int(chunk_size)
        self.NFFT = int(NFFT)
        self.overlap = float(overlap)
        self.new_samps = int(round(self.chunk_size * (1.0 - self.overlap)))
        self.TAG = TAG
        self.df = self.fs / self.NFFT
        self.frq2 = (np.arange(self.NFFT) - self.NFFT // 2) * self.df
        self.frq = self.frq2.copy()
        self._buf = np.zeros(self.chunk_size, np.complex64)
        win = np.ascontiguousarray(design.psd_window(self.chunk_size), np.float32)
        h = C.c_void_p()
        check(_lib.lib().pysdr_spectrum_create(device, self.chunk_size, self.NFFT, 1,
                                               _lib.as_pf(win), C.byref(h)), "pysdr_spectrum_create")
        self._h = h

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be torn down: leave the device
        # memory to process exit rather than call into a dead runtime
        if sys.is_finalizing():
            return
        try:
            if self._h:
                _lib.lib().pysdr_spectrum_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def periodogram(self, y, db=True):
        y = np.asarray(y)
        n = len(y)
        if n == 0:
            return np.zeros(0, np.float32)
        is_real = not np.iscomplexobj(y)
        if n >= self.chunk_size:
            self._buf = np.ascontiguousarray(y[n - self.chunk_size:], np.complex64)
        else:
            self._buf = np.concatenate((self._buf[n:], y.astype(np.complex64)))
        if is_real:
            frame = np.ascontiguousarray(self._buf.real, np.float32)
            out = np.empty(self.NFFT // 2, np.float32)
            self.frq = np.arange(self.NFFT // 2) * self.df
        else:
            frame = self._buf.view(np.float32)
            out = np.empty(self.NFFT, np.float32)
            self.frq = self.frq2
        nout = C.c_int(0)
        check(_lib.lib().pysdr_spectrum_frame(self._h, _lib.as_pf(frame), 0 if is_real else 1,
                                              1 if db else 0, _lib.as_pf(out), C.byref(nout)),
              "pysdr_spectrum_frame")
        return out[:nout.value]

    def psd_est(self, x, db=True):
        """Welch average over a long record (``sigs/iq.py:76``): mean of the linear
        periodograms of successive ``new_samps`` hops, then dB."""
        x = np.asarray(x)
        acc, cnt = None, 0
        self._buf[:] = 0
        for i in range(0, len(x) - self.new_samps + 1, self.new_samps):
            p = self.periodogram(x[i:i + self.new_samps], False)
            if i + self.new_samps >= self.chunk_size:
                acc = p.astype(np.float64) if acc is None else acc + p
                cnt += 1
        if acc is None:
            return np.zeros(0, np.float32)
        acc /= cnt
        return (10 * np.log10(acc + 1e-30)).astype(np.float32) if db else acc.astype(np.float32)


# ----------------------------------------------------------------------------------------
class convolver:
    """``dsp.convolver(h, dtype)`` with ``convolve_fast(x)`` (``receiver.py:207,216,862``):
    streaming FIR with carried state (the aux-audio band-pass).  Complex input is filtered
    as two real streams."""

    def __init__(self, h, dtype=np.float32, device=0):
        self.h = np.ascontiguousarray(h, np.float32)
        self.dtype = dtype
        self.device = device
        self._hist = np.zeros(len(self.h) - 1, np.complex64)

    def _run(self, xx):
        n = len(xx) - (len(self.h) - 1)
        y = np.empty(n, np.float32)
        xx = np.ascontiguousarray(xx, np.float32)
        check(_lib.lib().pysdr_fir_real(self.device, _lib.as_pf(xx), _lib.as_pf(self.h), len(self.h),
                                        _lib.as_pf(y), n), "pysdr_fir_real")
        return y

    def convolve_fast(self, x):
        _lib.require_gpu()
        x = np.asarray(x)
        if len(x) == 0:
            return x.astype(np.float32)
        buf = np.concatenate((self._hist, x.astype(np.complex64)))
        self._hist = buf[len(buf) - (len(self.h) - 1):]
        if np.iscomplexobj(x):
            return (self._run(buf.real) + 1j * self._run(buf.imag)).astype(np.complex64)
        return self._run(buf.real)


# ----------------------------------------------------------------------------------------
class ring_buffer2:
    """Thread-safe sample FIFO between the RX thread and audio/PSD consumers
    (``pySDR.py:103-109``; ``receiver.py:72,848``).  Pure plumbing, no arithmetic.
    ``buf`` is a queue of the pushed blocks: callers poll ``rb.buf.qsize()``
    (``receiver.py:565``)."""

    def __init__(self, tag, size, PREVENT_OVERFLOW=True):
        self.tag = tag
        self.size = int(size)
        self.prevent_overflow = PREVENT_OVERFLOW
        self.buf = queue.Queue()
        self._lock = threading.Lock()
        self._head = None          # partially consumed block
        self.nsamps = 0

    def push(self, x):
        x = np.asarray(x)
        if len(x) == 0:
            return True
        with self._lock:
            if self.prevent_overflow and self.nsamps + len(x) > self.size:
                return False           # drop the block rather than grow without bound
            self.buf.put(x.copy())
            self.nsamps += len(x)
        return True

    def push_zeros(self, n):
        return self.push(np.zeros(int(n), np.float32))

    def ready(self, n):
        return self.nsamps >= n

    def clear(self):
        with self._lock:
            while not self.buf.empty():
                self.buf.get_nowait()
            self._head = None
            self.nsamps = 0

    def _take(self, n):
        parts, got = [], 0
        while got < n:
            if self._head is None:
                self._head = self.buf.get_nowait()
            blk = self._head
            k = min(n - got, len(blk))
            parts.append(blk[:k])
            got += k
            self._head = blk[k:] if k < len(blk) else None
        self.nsamps -= n
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def pull(self, n, flush=False):
        """Oldest ``n`` samples; ``flush=True`` first drops any backlog beyond the most
        recent ``n`` (``gui.py:1266`` "backlog flushed").  ``[]`` when not enough."""
        n = int(n)
        with self._lock:
            if self.nsamps < n:
                return []
            if flush and self.nsamps > n:
                self._take(self.nsamps - n)
            return self._take(n)


class ring_buffer3(ring_buffer2):
    """Same interface over a ``multiprocessing.Queue`` (MP_SCHEME 2, ``utils.py:84``)."""

    def __init__(self, tag, size):
        super().__init__(tag, size, PREVENT_OVERFLOW=False)
        self.buf = mp.Queue()
        self._count = mp.Value('l', 0)

    def push(self, x):
        x = np.asarray(x)
        if len(x):
            self.buf.put(x.copy())
            with self._count.get_lock():
                self._count.value += len(x)
            self.nsamps = self._count.value
        return True

    def ready(self, n):
        self.nsamps = self._count.value
        return self.nsamps >= n

    def clear(self):
        try:
            while True:
                self.buf.get_nowait()
        except queue.Empty:
            pass
        self._head = None
        with self._count.get_lock():
            self._count.value = 0
        self.nsamps = 0

    def _take(self, n):
        parts, got = [], 0
        while got < n:
            if self._head is None:
                self._head = self.buf.get(timeout=1.0)
            blk = self._head
            k = min(n - got, len(blk))
            parts.append(blk[:k])
            got += k
            self._head = blk[k:] if k < len(blk) else None
        with self._count.get_lock():
            self._count.value -= n
        self.nsamps = self._count.value
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def pull(self, n, flush=False):
        n = int(n)
        if self._count.value < n:
            return []
        if flush and self._count.value > n:
            self._take(self._count.value - n)
        return self._take(n)
