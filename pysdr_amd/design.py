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
