"""Deterministic synthetic wideband IQ (SURVEY.md 8(d) configurations C1-C3): the signal
source of the synthetic SoapySDR-shaped device (``pysdr_amd/stream.py``), of the tests and
of ``bench.py``.  NumPy only; nothing here is on the measured path."""
from __future__ import annotations

import math

import numpy as np


def synth_iq(cfg, nsamp, seed, n_start=0):
    """Deterministic synthetic wideband IQ: a sum of
    modulated carriers + AWGN.  ``cfg`` = dict(fs=, carriers=[dict(f=, kind=, ...)],
    noise=)."""
    fs = cfg['fs']
    rng = np.random.default_rng(seed)
    n = np.arange(n_start, n_start + nsamp, dtype=np.float64)
    t = n / fs
    x = np.zeros(nsamp, np.complex128)
    for c in cfg['carriers']:
        a, f, kind = c.get('amp', 0.2), c['f'], c['kind']
        ft = c.get('tone', 1000.0)
        if kind == 'am':
            env = 1.0 + c.get('depth', 0.5) * np.sin(2 * np.pi * ft * t)
            x += a * env * np.exp(2j * np.pi * f * t)
        elif kind == 'fm':
            dev = c.get('dev', 3000.0)
            ph = 2 * np.pi * f * t - (dev / ft) * np.cos(2 * np.pi * ft * t)
            x += a * np.exp(1j * ph)
        elif kind == 'usb':
            x += a * np.exp(2j * np.pi * (f + ft) * t)
            x += 0.5 * a * np.exp(2j * np.pi * (f + 1.7 * ft) * t)
        elif kind == 'cw':
            key = (np.floor(t * c.get('wpm_hz', 12.0)) % 2 == 0).astype(np.float64)
            x += a * key * np.exp(2j * np.pi * f * t)
        else:
            raise ValueError(kind)
    sig = cfg.get('noise', 1e-2)
    x += sig * (rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) / math.sqrt(2)
    return x.astype(np.complex64)


CONFIGS = {
    # am.py path
    'C1': dict(fs=2.048e6, fs_out=48e3, ntaps_dec=1001, noise=2e-3,
               carriers=[dict(f=100e3, kind='am', amp=0.3, tone=1000.0, depth=0.5)],
               rx=[dict(frq=100e3, mode='AM', video_bw=10e3, af_bw=5e3)]),
    # C2: 8 MS/s, 1 RX NBFM, 255-tap prototype
    'C2': dict(fs=8e6, fs_out=48e3, ntaps_dec=255, noise=2e-3,
               carriers=[dict(f=455e3, kind='fm', amp=0.3, tone=1000.0, dev=3000.0)],
               rx=[dict(frq=455e3, mode='NFM', video_bw=20e3, af_bw=4e3)]),
    # C3: 8 MS/s, 4 RX USB/CW/NBFM/AM (+ RF PSD in the harness)
    'C3': dict(fs=8e6, fs_out=48e3, ntaps_dec=255, noise=2e-3,
               carriers=[dict(f=200e3, kind='usb', amp=0.15, tone=900.0),
                         dict(f=-310e3, kind='cw', amp=0.15),
                         dict(f=455e3, kind='fm', amp=0.2, tone=1000.0, dev=3000.0),
                         dict(f=-1.2e6, kind='am', amp=0.2, tone=700.0, depth=0.6)],
               rx=[dict(frq=200e3, mode='USB', video_bw=10e3, af_bw=3e3),
                   dict(frq=-310e3, mode='CW', video_bw=10e3, af_bw=500.0, bfo=700.0),
                   dict(frq=455e3, mode='NFM', video_bw=20e3, af_bw=4e3),
                   dict(frq=-1.2e6, mode='AM', video_bw=10e3, af_bw=5e3)]),
    # The reference's largest evidenced operating point (FT8tri:26,47,56,74): `-fc 18100 21074 24915 -fs 8 -IF 0
    # -foffset 0 -mode USB -fsout 48 -af_bw 5 -vid_bw 45`, filter length left at its default 1001 (params.py:134).
    # Sub-receiver frq = FC[irx] - FC[0] (receiver.py:834): 0, +2974 kHz, +6815 kHz -- the last one beyond fs/2, which the
    # 32-bit LO word wraps to -1185 kHz; the synthetic USB signals sit where those LOs listen.
    'FT8TRI': dict(fs=8e6, fs_out=48e3, ntaps_dec=1001, noise=2e-3,
                   carriers=[dict(f=0.0, kind='usb', amp=0.15, tone=1200.0),
                             dict(f=2974e3, kind='usb', amp=0.2, tone=900.0),
                             dict(f=-1185e3, kind='usb', amp=0.1, tone=1500.0)],
                   rx=[dict(frq=0.0, mode='USB', video_bw=45e3, af_bw=5e3),
                       dict(frq=2974e3, mode='USB', video_bw=45e3, af_bw=5e3),
                       dict(frq=6815e3, mode='USB', video_bw=45e3, af_bw=5e3)]),
    # TEST:13,30,32 `-fc 145300 144700 -fs 4 -IF 0 -af_bw 5 -mode NFM -fsout 48`: a repeater's output and input, two NFM
    # sub-receivers 600 kHz apart at 4 MS/s (3/250), default 1001-tap prototype, default video bandwidth (10 kHz, params.py:329)
    'TEST2RX': dict(fs=4e6, fs_out=48e3, ntaps_dec=1001, noise=2e-3,
                    carriers=[dict(f=0.0, kind='fm', amp=0.25, tone=1000.0, dev=3000.0),
                              dict(f=-600e3, kind='fm', amp=0.2, tone=700.0, dev=2500.0)],
                    rx=[dict(frq=0.0, mode='NFM', video_bw=10e3, af_bw=5e3),
                        dict(frq=-600e3, mode='NFM', video_bw=10e3, af_bw=5e3)]),
}


def synth_wfm(fs, nsamp, seed, f_carrier=300e3, tone_l=1000.0, tone_r=2500.0, amp=0.3, noise=2e-3):
    """Synthetic stereo FM broadcast: composite = 0.9*((L+R)/2 +
    (L-R)/2*sin(2 wp t)) + 0.1*sin(wp t), +-75 kHz deviation."""
    rng = np.random.default_rng(seed)
    t = np.arange(nsamp, dtype=np.float64) / fs
    left = 0.8 * np.sin(2 * np.pi * tone_l * t)
    right = 0.6 * np.sin(2 * np.pi * tone_r * t)
    wp = 2 * np.pi * 19000.0
    mpx = 0.9 * (0.5 * (left + right) + 0.5 * (left - right) * np.sin(2 * wp * t)) + 0.1 * np.sin(wp * t)
    phase = 2 * np.pi * f_carrier * t + 2 * np.pi * 75e3 * np.cumsum(mpx) / fs
    x = amp * np.exp(1j * phase)
    x += noise * (rng.standard_normal(nsamp) + 1j * rng.standard_normal(nsamp)) / math.sqrt(2)
    return x.astype(np.complex64)
