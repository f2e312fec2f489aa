"""CPU oracle of the broadcast-FM path (modes WFM = mono, WFM2 = stereo; BASELINE config
"WBFM stereo path: 10 MS/s IQ, 1 RX, pilot-PLL stereo demod + 75 us de-emphasis").
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity unpinned: the reference only
tells the ORDER (``gui.py:1703-1704,1759-1762``; ``receiver.py:718-719``: for BCB FM the
signal is demodulated before the sample-rate reduction, ``rx.demod.wfm_video`` is the
pre-detection filter and the resampler's ``dec.h`` does the audio filtering) and the defaults
(``params.py:325-327`` VIDEO_BW 200 kHz).  Spec = DESIGN.md 3.10:

  x --LO mix--> video FIR (FILT_LEN taps at SRATE) + integer decimation D1 --> y1 @ fs1~250 kHz
    --polar discriminator arg(y[n]*conj(y[n-1])), 75 kHz = 1.0 (the central-difference form of
      sigs/nfm.m is only linear for phase steps << 1 rad; here the step reaches 1.9 rad)--> mpx
    --[WFM2: 19 kHz pilot PLL, w = mpx*(1 + 2j*sin(2*theta))]--> w
    --rational resample fs1 -> FS_OUT (UP2/DOWN2, 19.5 kHz prototype)--> z = S + jD
    --AF FIR = 15 kHz low-pass (*) 75 us de-emphasis (truncated one-pole, exact to 1e-11)-->
    WFM: am = S ;  WFM2: am = (S+D) + j(S-D) = L + jR
"""
from __future__ import annotations

import math

import numpy as np
from scipy import signal

from . import sdr_oracle as so

WFM_IF_TARGET = 250e3
WFM_FULL_SCALE_DEV = 75e3
WFM_PILOT_HZ = 19000.0
WFM_PILOT_LEVEL = 0.1
WFM_PLL_BW_HZ = 30.0
WFM_DEEMPH_TAU = 75e-6
WFM_DEEMPH_TAPS = 96
WFM_AUDIO_CUT = 15e3
WFM_RESAMP_CUT = 19.5e3
WFM_RESAMP_TAPS_PER_PHASE = 64


def wfm_if_decim(srate):
    """Integer decimation D1 to the FM IF rate: the divisor of SRATE whose quotient is
    closest to 250 kHz."""
    srate = int(round(srate))
    best, bd = 1, None
    for d in range(1, max(2, srate // 100000) + 1):
        if srate % d:
            continue
        err = abs(srate / d - WFM_IF_TARGET)
        if bd is None or err < bd:
            best, bd = d, err
    return best


def wfm_video_bank(srate, fs1, ntaps, video_bw_other=200e3, labels=so.VIDEO_BWs):
    """``rx.demod.wfm_filter_bank`` (``gui.py:1704``): pre-detection low-pass at SRATE."""
    fmax = 0.45 * fs1
    bank = np.empty((len(labels), ntaps), np.float64)
    for i, lab in enumerate(labels):
        bw = so.parse_bw(lab)
        fc = fmax if lab == 'Max' else (0.5 * video_bw_other if lab == 'Other' else 0.5 * bw)
        bank[i] = signal.firwin(ntaps, min(fc, fmax), window='hamming', fs=float(srate))
    return bank


def wfm_resampler_taps(fs1, up2):
    n = up2 * WFM_RESAMP_TAPS_PER_PHASE
    return up2 * signal.firwin(n, WFM_RESAMP_CUT, window='hamming', fs=float(fs1) * up2)


def wfm_af_taps(fs_out, ntaps, af_bw=0.0):
    """15 kHz audio low-pass convolved with the 75 us de-emphasis (one-pole IIR
    y = a*x + b*y[-1], b = exp(-1/(fs*tau)), truncated after 96 taps: b^96 = 2.5e-12)."""
    cut = af_bw if 0 < af_bw <= WFM_AUDIO_CUT else WFM_AUDIO_CUT
    b = math.exp(-1.0 / (fs_out * WFM_DEEMPH_TAU))
    de = (1.0 - b) * b ** np.arange(WFM_DEEMPH_TAPS)
    lp = signal.firwin(ntaps - WFM_DEEMPH_TAPS + 1, cut, window='hamming', fs=float(fs_out))
    return np.convolve(lp, de)


class PilotPLL:
    """19 kHz pilot PLL at fs1 on a 32-bit phase accumulator (exact wrap; float32 rounding
    only touches the small loop quantities):
        c = cos(theta), e = mpx*c*2/level, w += ki*e,
        phase += fword0 + rint((w + kp*e) * 2^32/(2*pi)),   theta = 2*pi*phase/2^32
    emits w[n] = mpx*(1 + 2j*sin(2*theta))."""

    def __init__(self, fs1, dtype=np.float32):
        self.rd = dtype
        wn = 2 * math.pi * WFM_PLL_BW_HZ / fs1
        self.kp = dtype(2 * 0.7071 * wn)
        self.ki = dtype(wn * wn)
        self.fword0 = so.freq_word(WFM_PILOT_HZ, fs1)[0]
        self.norm = dtype(2.0 / WFM_PILOT_LEVEL)
        self.rad2word = dtype(so.TWO32 / (2 * math.pi))
        self.reset()

    def reset(self):
        self.phase = 0
        self.w = self.rd(0)

    def process(self, mpx):
        rd = self.rd
        cd = np.complex64 if rd == np.float32 else np.complex128
        out = np.empty(len(mpx), cd)
        ph, w = int(self.phase), self.w
        two, inv = rd(2), rd(1.0 / so.TWO32)
        twopi = 2 * math.pi
        for i in range(len(mpx)):
            m = rd(mpx[i])
            sp = ph - so.TWO32 if ph >= (so.TWO32 >> 1) else ph          # signed 32-bit phase
            rev = rd(rd(sp) * inv)                                       # revolutions in [-0.5, 0.5)
            c = rd(np.cos(twopi * float(rev)))
            s2 = rd(np.sin(2 * twopi * float(rev)))
            e = rd(rd(m * c) * self.norm)
            out[i] = complex(m, rd(m * rd(two * s2)))
            w = rd(w + rd(self.ki * e))
            corr = int(np.rint(rd(rd(w + rd(self.kp * e)) * self.rad2word)))
            ph = (ph + self.fword0 + corr) % so.TWO32
        self.phase, self.w = ph, w
        return out


class WfmReceiver:
    """``dsp.Receiver`` in mode WFM / WFM2 (``Tables.py:34``)."""

    def __init__(self, srate, fs_out_req, frq, stereo=True, ntaps_dec=255, ntaps_af=255,
                 video_bw=200e3, af_bw=0.0, dtype=np.float32):
        self.rd = dtype
        self.cd = np.complex64 if dtype == np.float32 else np.complex128
        self.mode = 'WFM2' if stereo else 'WFM'
        self.srate = float(srate)
        self.fs_out = so.chunk_sizes(srate, fs_out_req)[2]
        self.in_chunk = so.chunk_sizes(srate, fs_out_req)[3]
        self.d1 = wfm_if_decim(srate)
        self.fs1 = self.srate / self.d1
        self.up2, self.down2 = so.up_dn(self.fs1, self.fs_out)
        self.lo = so.NCO(-frq, srate, dtype)
        bank = wfm_video_bank(srate, self.fs1, ntaps_dec, video_bw)
        vidx = so.Receiver._video_index(video_bw)
        self.front = so.RationalDecimator(bank[vidx], 1, self.d1, dtype)
        self.front.filter_bank = bank
        self.xhist = np.zeros(self.front.kmax - 1, self.cd)
        self.y1hist = np.zeros(1, self.cd)
        self.pll = PilotPLL(self.fs1, dtype)
        self.audio = so.RationalDecimator(wfm_resampler_taps(self.fs1, self.up2), self.up2, self.down2, dtype)
        self.demod = so.Demodulator(self.fs_out, ntaps_af, dtype)
        self.demod.set_taps(wfm_af_taps(self.fs_out, ntaps_af, af_bw).astype(np.complex128))
        self.agc = so.AGC(dtype)
        self.am = np.zeros(0, dtype)
        self.iq = np.zeros(0, self.cd)

    def demod_data(self, x):
        x = np.asarray(x, self.cd)
        hl = len(self.xhist)
        raw = np.concatenate((self.xhist, x))
        idx = np.arange(-hl, len(x), dtype=np.int64)
        ph = (self.lo.phase + self.lo.fword * idx) % so.TWO32
        v = (raw * so.phase_to_cplx(ph.astype(np.uint32), self.cd)).astype(self.cd)
        self.lo.phase = (self.lo.phase + self.lo.fword * len(x)) % so.TWO32
        self.front.hist = v[:hl]
        y1 = self.front.process(v[hl:])
        self.xhist = raw[len(raw) - hl:]
        y2 = np.concatenate((self.y1hist, y1))
        scale = self.rd(self.fs1 / (2 * math.pi * WFM_FULL_SCALE_DEV))
        ya, yb = y2[:-1], y2[1:]
        re = (yb.real * ya.real + yb.imag * ya.imag).astype(self.rd)
        im = (yb.imag * ya.real - yb.real * ya.imag).astype(self.rd)
        mpx = (np.arctan2(im, re).astype(self.rd) * scale).astype(self.rd)
        self.y1hist = y2[len(y2) - 1:]
        if self.mode == 'WFM2':
            w = self.pll.process(mpx)
        else:
            w = mpx.astype(self.cd)
        z = self.audio.process(w)
        a = self.demod.process(z, 'IQ', 0.0)
        peak = np.max(np.abs(a)) if len(a) else 0.0
        self.agc.update(peak, False)
        if self.mode == 'WFM2':
            am = ((a.real + a.imag) + 1j * (a.real - a.imag)).astype(self.cd)
        else:
            am = a.real.astype(self.rd)
        self.am, self.iq = am, z
        return am


from pysdr_amd.synth import synth_wfm  # noqa: E402,F401  (synthetic input generator, shared with the product's probes)
