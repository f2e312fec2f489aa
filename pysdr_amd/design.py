"""Host-side filter design for the receiver (SciPy): the filter banks the reference
keeps on ``rx.dec.filter_bank`` / ``rx.demod.filter_bank_real`` /
``rx.demod.filter_bank_cmpx`` (``receiver.py:866-874``; ``gui.py:1704,1713``).  The
reference's own design code is in the absent ``sig_proc``; the choices below are this
build's spec."""
from __future__ import annotations

import numpy as np
from scipy.signal import firwin

from .tables import AF_BWs, VIDEO_BWs, label_hz

NYQ_FRACTION = 0.45      # widest pass band as a fraction of the lower sample rate


def decimator_bank(srate, up, fs_out, ntaps, video_bw_other=10e3, labels=VIDEO_BWs):
    """One low-pass prototype per VIDEO_BWs label, designed at srate*up, DC gain UP."""
    fs_up = float(srate) * up
    widest = NYQ_FRACTION * min(float(srate), float(fs_out))
    rows = []
    for lab in labels:
        hz = label_hz(lab)
        if lab == 'Max':
            cut = widest
        elif lab == 'Other':
            cut = 0.5 * video_bw_other
        else:
            cut = 0.5 * hz
        rows.append(up * firwin(ntaps, min(cut, widest), window='hamming', fs=fs_up))
    return np.asarray(rows, np.float64)


def af_bank_real(fs_out, ntaps, labels=AF_BWs):
    """Real low-pass per AF_BWs label; 'Max' is a unit impulse."""
    widest = NYQ_FRACTION * fs_out
    bank = np.zeros((len(labels), ntaps), np.float64)
    for i, lab in enumerate(labels):
        hz = label_hz(lab)
        if hz is None:
            bank[i, 0] = 1.0
        else:
            bank[i] = firwin(ntaps, min(hz, widest), window='hamming', fs=fs_out)
    return bank


def _one_sided(fs_out, ntaps, bw, centre):
    k = np.arange(ntaps) - 0.5 * (ntaps - 1)
    lp = firwin(ntaps, 0.5 * bw, window='hamming', fs=fs_out)
    return 2.0 * lp * np.exp(2j * np.pi * centre * k / fs_out)


def af_bank_cmpx(fs_out, ntaps, labels=AF_BWs):
    """Analytic band-pass [0, bw] per label (upper side band), gain 2."""
    widest = NYQ_FRACTION * fs_out
    bank = np.zeros((len(labels), ntaps), np.complex128)
    for i, lab in enumerate(labels):
        hz = label_hz(lab)
        bw = widest if hz is None else min(hz, widest)
        bank[i] = _one_sided(fs_out, ntaps, bw, 0.5 * bw)
    return bank


def cw_taps(fs_out, ntaps, bw, bfo):
    widest = NYQ_FRACTION * fs_out
    bw = widest if not bw else min(bw, widest)
    return _one_sided(fs_out, ntaps, bw, bfo)


def bpf(f1, f2, fs, ntaps):
    """``dsp.bpf(800.,1300.,P.FS_OUT,1001)`` (``receiver.py:861``)."""
    return firwin(ntaps, [f1, f2], pass_zero=False, window='hamming', fs=fs)


def psd_window(n, beta=8.6):
    """Kaiser(8.6) as in ``rtty.py`` (pin P5), scaled to unit coherent gain."""
    w = np.kaiser(n, beta)
    return w / np.sum(w)


# ---- broadcast FM (modes WFM / WFM2), DESIGN.md 3.10 ------------------------------------
WFM_DEEMPH_TAU = 75e-6
WFM_DEEMPH_TAPS = 96
WFM_AUDIO_CUT = 15e3
WFM_RESAMP_CUT = 19.5e3
WFM_RESAMP_TAPS_PER_PHASE = 64


def wfm_video_bank(srate, fs1, ntaps, video_bw_other=200e3, labels=VIDEO_BWs):
    """``rx.demod.wfm_filter_bank`` (``gui.py:1704``): pre-detection low-pass at SRATE, one
    per VIDEO_BWs label, clamped to 0.45*fs1 (the IF rate after the integer decimation)."""
    widest = NYQ_FRACTION * fs1
    rows = []
    for lab in labels:
        hz = label_hz(lab)
        cut = widest if lab == 'Max' else (0.5 * video_bw_other if lab == 'Other' else 0.5 * hz)
        rows.append(firwin(ntaps, min(cut, widest), window='hamming', fs=float(srate)))
    return np.asarray(rows, np.float64)


def wfm_resampler_taps(fs1, up2):
    """Prototype of the fs1 -> FS_OUT rational resampler (designed at fs1*up2, gain up2)."""
    return up2 * firwin(up2 * WFM_RESAMP_TAPS_PER_PHASE, WFM_RESAMP_CUT, window='hamming',
                        fs=float(fs1) * up2)


def wfm_af_taps(fs_out, ntaps, af_bw=0.0):
    """15 kHz audio low-pass convolved with the 75 us de-emphasis: the one-pole IIR
    y = (1-b)*x + b*y[-1], b = exp(-1/(fs*tau)), as its impulse response truncated after 96
    taps (b^96 = 2.5e-12, below float32 resolution)."""
    cut = af_bw if 0 < af_bw <= WFM_AUDIO_CUT else WFM_AUDIO_CUT
    b = np.exp(-1.0 / (fs_out * WFM_DEEMPH_TAU))
    de = (1.0 - b) * b ** np.arange(WFM_DEEMPH_TAPS)
    lp = firwin(ntaps - WFM_DEEMPH_TAPS + 1, cut, window='hamming', fs=float(fs_out))
    return np.convolve(lp, de)


def squelch_ratio_taps(fs_out, ntaps=63):
    """The two FIRs of the ratio squelch (``sigs/squelch.m:103-105``: low-pass 3 kHz, high-pass 4 kHz; there elliptic IIRs of
    order 5, here their linear-phase FIR equivalents) -> (lp, hp) float32."""
    lp = firwin(ntaps, 3000.0, window='hamming', fs=fs_out)
    hp = firwin(ntaps, 4000.0, window='hamming', pass_zero=False, fs=fs_out)
    return np.ascontiguousarray(lp, np.float32), np.ascontiguousarray(hp, np.float32)
