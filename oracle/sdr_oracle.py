"""NumPy/SciPy restatement of the pySDR receiver hot path (TEST INFRASTRUCTURE).

See ``oracle/__init__.py`` for the parity status ("partially pinned") and the
list of in-tree pins.  Every class below states which reference call site it
stands behind.  ``dtype=np.float32`` is the float32 mirror the GPU is compared
with (tolerance 1e-5 of the output peak); ``dtype=np.float64`` is the master
used to check that the float32 mirror itself is sane.

The normative DSP definitions live in DESIGN.md section 3; this file and the HIP
kernels are two independent implementations of that text.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np
from scipy import signal

# Tables.py:34-42 (data tables of the reference; labels are the filter-bank keys)
MODES = ["AM", "AM-Synch", "SSB", "USB", "LSB", "CW", "IQ", "WFM", "WFM2", "NFM", "RTTY"]
AF_BWs = ['Max', '50 Hz', '100 Hz', '500 Hz', '1 KHz', '2 KHz', '3 KHz',
          '4 KHz', '5 KHz', '8 KHz', '10 KHz', '15 KHz', '20 KHz', '45 KHz',
          '50 KHz', '100 KHz', '150 KHz', '200 KHz']
VIDEO_BWs = ['Max', '5 KHz', '10 KHz', '20 KHz', '25 KHz', '45 KHz', '50 KHz',
             '100 KHz', '150 KHz', '200 KHz', '300 KHz', '400 KHz', '500 KHz',
             '750 KHz', '1 MHz', 'Other']

OUT_CHUNK_SIZE = 1024          # params.py:440
TWO32 = 1 << 32

# ---- spec constants (all "unpinned" unless noted) ------------
NFM_FULL_SCALE_DEV = 5000.0    # Hz of deviation that maps to audio amplitude 1.0
AGC_BETA = 0.1                 # sigs/agc.m:6 (pin P3)
AGC_REF = 0.5                  # target block peak
AGC_GAIN_MAX = 1.0e4
AGC_MODES = ("AM", "AM-Synch", "SSB", "USB", "LSB", "CW", "RTTY")
PSD_KAISER_BETA = 8.6          # rtty.py window (pin P5)
PSD_FLOOR = 1.0e-30
AUTO_MUTE_THRESH = 0.7         # |x| peak of the raw chunk that trips auto-mute
PLL_ZETA = 0.7071
PLL_BW_HZ = 50.0
SQUELCH_ALPHA = 0.64           # per-block smoothing = 1-(1-0.001)^1024 (sigs/squelch.m alpha=0.001/sample)
# The RATIO squelch, as sigs/squelch.m:92-145 sketches it: z1 = low-pass < 3 kHz and z2 = high-pass > 4 kHz of the
# discriminator output (there elliptic IIRs of order 5, :103-105; here their FIR equivalents -- the path has no IIR machinery and a
# linear-phase pair keeps the two envelopes aligned), sq1 / sq2 = one-pole envelopes of |z1| / |z2| PER SAMPLE with
# alpha = 0.001 (:131-134), the decision on sq1 / sq2 (:145) -- amplitude independent.  "parity unpinned": the reference has
# the sketch, no run-time call.
SQ_RATIO_ALPHA = 0.001
SQ_RATIO_TAPS = 63
SQ_RATIO_LP_HZ = 3000.0
SQ_RATIO_HP_HZ = 4000.0


def squelch_ratio_taps(fs_out, dtype=np.float32):
    """(low-pass < 3 kHz, high-pass > 4 kHz), 63 taps each, Hamming (``sigs/squelch.m:103-105``: the bands)."""
    lp = signal.firwin(SQ_RATIO_TAPS, SQ_RATIO_LP_HZ, window='hamming', fs=fs_out)
    hp = signal.firwin(SQ_RATIO_TAPS, SQ_RATIO_HP_HZ, window='hamming', pass_zero=False, fs=fs_out)
    return lp.astype(dtype), hp.astype(dtype)


# ------------------------------------------------------------------ rates / sizes
def up_dn(fs_in, fs_out):
    """(UP, DOWN) = reduced fraction fs_out/fs_in.  Call sites ``params.py:405``,
    ``receiver.py:818``; known answers ``srates.py:35-74`` (pin P1)."""
    fr = Fraction(int(round(fs_out)), int(round(fs_in)))
    return fr.numerator, fr.denominator


def chunk_sizes(srate, fs_out_req):
    """``params.py:405-406,440-444``: returns (UP, DOWN, FS_OUT, IN_CHUNK_SIZE)."""
    up, down = up_dn(srate, fs_out_req)
    fs_out = int(srate * up / down)
    in_chunk = int(OUT_CHUNK_SIZE * down / float(up) + 0 * 0.5)
    return up, down, fs_out, in_chunk


def rb_size(num_rx, sdr_type, fs_out):
    """``params.py:456-468`` ring-buffer sizing rules."""
    n = 32 * OUT_CHUNK_SIZE
    if num_rx > 2:
        n *= 4
    if sdr_type == 'rtlsdr':
        n *= 2
    if fs_out > 100e3:
        n *= 4
    elif fs_out > 50e3:
        n *= 2
    return n


def adjust_foffset(foffset, srate, rbsize):
    """``utils.py:277-289``: snap the tuning offset to M*SRATE/RB_SIZE."""
    m = round(rbsize * foffset / srate)
    return m * srate / rbsize


def af_gain(slider):
    """``receiver.py:200``: the AF slider is a dB-like scale."""
    return pow(10., slider) - 1


def parse_bw(label):
    """'5 KHz' -> 5000.0 ; 'Max'/'Other' -> None  (``Tables.py:48-62`` parsing)."""
    if label in ('Max', 'Other'):
        return None
    a = label.split(' ')
    b = float(int(a[0]))
    if a[1] == 'KHz':
        b *= 1e3
    elif a[1] == 'MHz':
        b *= 1e6
    return b


def out_index_range(s0, s1, up, down):
    """Outputs m of the rational resampler whose newest input sample
    n_m = floor(m*DOWN/UP) lies in [s0, s1)."""
    return (s0 * up + down - 1) // down, (s1 * up + down - 1) // down


# ------------------------------------------------------------------ filter design
def dec_filter_bank(srate, up, fs_out, ntaps, video_bw_other=10e3, labels=VIDEO_BWs):
    """Prototype low-pass bank of the rational resampler (``rx.dec.filter_bank``,
    ``receiver.py:127,866``; ``gui.py:1713``).  Designed at the up-sampled rate
    ``srate*up``, DC gain UP, Hamming window, length ``ntaps`` (P.FILT_LEN,
    ``params.py:134,345``).  Cut-off = label/2, clamped to 0.45*min(srate, fs_out)."""
    fs_up = float(srate) * up
    fmax = 0.45 * min(float(srate), float(fs_out))
    bank = np.empty((len(labels), ntaps), np.float64)
    for i, lab in enumerate(labels):
        bw = parse_bw(lab)
        if lab == 'Max':
            fc = fmax
        elif lab == 'Other':
            fc = 0.5 * video_bw_other
        else:
            fc = 0.5 * bw
        fc = min(fc, fmax)
        bank[i] = up * signal.firwin(ntaps, fc, window='hamming', fs=fs_up)
    return bank


def af_filter_bank_real(fs_out, ntaps, labels=AF_BWs):
    """``rx.demod.filter_bank_real`` (``receiver.py:873``): real low-pass, cut-off =
    label, 'Max' = unit impulse (no filtering, no delay)."""
    bank = np.zeros((len(labels), ntaps), np.float64)
    fmax = 0.45 * fs_out
    for i, lab in enumerate(labels):
        bw = parse_bw(lab)
        if bw is None:
            bank[i, 0] = 1.0
        else:
            bank[i] = signal.firwin(ntaps, min(bw, fmax), window='hamming', fs=fs_out)
    return bank


def af_filter_bank_cmpx(fs_out, ntaps, labels=AF_BWs):
    """``rx.demod.filter_bank_cmpx`` (``receiver.py:874``): one-sided (analytic)
    band-pass covering [0, bw] Hz, gain 2 so that Re(.) keeps the amplitude."""
    bank = np.zeros((len(labels), ntaps), np.complex128)
    fmax = 0.45 * fs_out
    k = np.arange(ntaps) - 0.5 * (ntaps - 1)
    for i, lab in enumerate(labels):
        bw = parse_bw(lab)
        bw = fmax if bw is None else min(bw, fmax)
        lp = signal.firwin(ntaps, 0.5 * bw, window='hamming', fs=fs_out)
        bank[i] = 2.0 * lp * np.exp(2j * np.pi * (0.5 * bw) * k / fs_out)
    return bank


def cw_filter(fs_out, ntaps, bw, bfo):
    """CW: band-pass of width ``bw`` centred on the BFO pitch (``params.py:318-320``)."""
    fmax = 0.45 * fs_out
    bw = fmax if not bw else min(bw, fmax)
    k = np.arange(ntaps) - 0.5 * (ntaps - 1)
    lp = signal.firwin(ntaps, 0.5 * bw, window='hamming', fs=fs_out)
    return 2.0 * lp * np.exp(2j * np.pi * bfo * k / fs_out)


def af_taps_for_mode(mode, af_idx, af_bw, bfo, fs_out, ntaps, lsb=False):
    """The complex AF taps applied after the per-mode detector."""
    if mode in ("AM", "AM-Synch", "NFM", "IQ", "WFM", "WFM2"):
        return af_filter_bank_real(fs_out, ntaps)[af_idx].astype(np.complex128)
    if mode == "CW":
        return cw_filter(fs_out, ntaps, af_bw, bfo)
    c = af_filter_bank_cmpx(fs_out, ntaps)[af_idx]
    if mode == "LSB" or (mode == "SSB" and lsb):
        c = np.conj(c)
    return c


def bpf(f1, f2, fs, ntaps):
    """``dsp.bpf(800.,1300.,P.FS_OUT,1001)`` (``receiver.py:861``): real band-pass."""
    return signal.firwin(ntaps, [f1, f2], pass_zero=False, window='hamming', fs=fs)


# ------------------------------------------------------------------ NCO / mixer
def freq_word(f, fs):
    """32-bit phase increment and the frequency it really produces."""
    w = int(round(float(f) / float(fs) * TWO32))
    ws = ((w + (TWO32 >> 1)) % TWO32) - (TWO32 >> 1)       # signed wrap
    return ws % TWO32, ws * float(fs) / TWO32


def phase_to_cplx(phase_u32, cdtype):
    """exp(j*2*pi*phase/2^32), phase taken as signed 32-bit."""
    ph = np.asarray(phase_u32, np.uint32).astype(np.int32).astype(np.float64)
    return np.exp(2j * np.pi * ph / TWO32).astype(cdtype)


class NCO:
    """``dsp.signal_generator(f, N, fs, True)`` (``receiver.py:822``): complex NCO with
    a persistent 32-bit phase accumulator.  ``quad_mixer(x) = x*exp(+j*phi_n)``
    (``receiver.py:552-553``); ``change_freq`` returns the quantised frequency
    (its return value becomes FOFFSET, ``gui.py:1928``)."""

    def __init__(self, f, fs, dtype=np.float32):
        self.fs = float(fs)
        self.cdtype = np.complex64 if dtype == np.float32 else np.complex128
        self.phase = 0
        self.fword, self.fo = freq_word(f, fs)

    def change_freq(self, f):
        self.fword, self.fo = freq_word(f, self.fs)
        return self.fo

    def phases(self, n):
        ph = (self.phase + self.fword * np.arange(n, dtype=np.uint64)) % TWO32
        return ph.astype(np.uint32)

    def quad_mixer(self, x):
        x = np.asarray(x, self.cdtype)
        lo = phase_to_cplx(self.phases(len(x)), self.cdtype)
        self.phase = (self.phase + self.fword * len(x)) % TWO32
        return x * lo


# ------------------------------------------------------------------ resampler
class RationalDecimator:
    """``rx.dec``: polyphase rational resampler UP/DOWN with a swappable prototype
    ``h`` (``receiver.py:127``; ``gui.py:1713``).  Definition = zero-stuff by UP,
    FIR ``h``, keep every DOWN-th sample (identical to ``scipy.signal.upfirdn``
    run over the whole stream):

        y[m] = sum_k h[p_m + UP*k] * v[n_m - k],  n_m = floor(m*DOWN/UP),
                                                  p_m = (m*DOWN) mod UP

    State: absolute sample counter + the last Kmax-1 input samples."""

    def __init__(self, h, up, down, dtype=np.float32):
        self.up, self.down = int(up), int(down)
        self.rdtype = dtype
        self.cdtype = np.complex64 if dtype == np.float32 else np.complex128
        self.n_abs = 0
        self.set_taps(h)
        self.hist = np.zeros(self.kmax - 1, self.cdtype)

    def set_taps(self, h):
        h = np.asarray(h, np.float64)
        self.h = h
        self.kmax = -(-len(h) // self.up)
        self.hp = []
        for p in range(self.up):
            hp = np.zeros(self.kmax, np.float64)
            sub = h[p::self.up]
            hp[:len(sub)] = sub
            self.hp.append(hp.astype(self.cdtype))

    def process(self, v):
        v = np.asarray(v, self.cdtype)
        up, down = self.up, self.down
        s0, s1 = self.n_abs, self.n_abs + len(v)
        m0, m1 = out_index_range(s0, s1, up, down)
        hl = len(self.hist)
        buf = np.concatenate((self.hist, v))
        m = np.arange(m0, m1, dtype=np.int64)
        t = m * down
        nm, pm = t // up, t % up
        out = np.empty(len(m), self.cdtype)
        kk = np.arange(self.kmax, dtype=np.int64)
        for p in range(up):
            sel = np.nonzero(pm == p)[0]
            if len(sel):
                idx = (nm[sel] - (s0 - hl))[:, None] - kk[None, :]
                out[sel] = buf[idx] @ self.hp[p]
        self.hist = buf[len(buf) - hl:] if hl else buf[:0]
        self.n_abs = s1
        return out


# ------------------------------------------------------------------ detectors
def nfm_discriminator(y3, rdtype):
    """Pin P2 (``sigs/nfm.m:124-127``): fm = Re(y1)*Im(d) - Im(y1)*Re(d), d = y[n+1]-y[n-1],
    here normalised by 2*|y1|^2 so the result approximates the phase step (rad/sample).
    ``y3`` holds y[m-2..] i.e. out[i] uses y3[i], y3[i+1], y3[i+2]."""
    ya, y1, yb = y3[:-2], y3[1:-1], y3[2:]
    d = yb - ya
    fm = y1.real * d.imag - y1.imag * d.real
    den = 2 * (y1.real * y1.real + y1.imag * y1.imag) + rdtype(1e-20)
    return (fm / den).astype(rdtype)


class CarrierPLL:
    """``rx.demod.am_pll`` (``receiver.py:649``): second-order PLL for AM-Synch.
    Per sample: v = y*exp(-j*theta); e = atan2(Im v, Re v); w += ki*e;
    theta += w + kp*e (wrapped to [-pi,pi)); output Re(v)."""

    def __init__(self, fs, dtype=np.float32):
        wn = 2 * math.pi * PLL_BW_HZ / fs
        self.rd = dtype
        self.kp = dtype(2 * PLL_ZETA * wn)
        self.ki = dtype(wn * wn)
        self.reset()

    def reset(self):
        self.theta = self.rd(0)
        self.w = self.rd(0)

    def process(self, y):
        rd = self.rd
        out = np.empty(len(y), rd)
        th, w, kp, ki = self.theta, self.w, self.kp, self.ki
        pi, twopi = rd(math.pi), rd(2 * math.pi)
        yr, yi = y.real.astype(rd), y.imag.astype(rd)
        for i in range(len(y)):
            c, s = rd(np.cos(th)), rd(np.sin(th))
            vr = yr[i] * c + yi[i] * s
            vi = yi[i] * c - yr[i] * s
            # a sample without amplitude steers nothing: the loop coasts on its integrator.  (Written out because
            # atan2(+0, -0) = pi: for theta in the third quadrant y = +0 + 0j gives v = -0 + 0j, and the "atan2(0, 0) = 0"
            # of the spec would depend on the signs of zeros.)
            e = rd(0) if (vr == 0 and vi == 0) else rd(np.arctan2(vi, vr))
            w = rd(w + ki * e)
            th = rd(th + rd(w + kp * e))
            if th >= pi:
                th = rd(th - twopi)
            elif th < -pi:
                th = rd(th + twopi)
            out[i] = vr
        self.theta, self.w = th, w
        return out


class AGC:
    """``rx.agc`` (``receiver.py:648``; fields read by ``watchdog.py:298-302``).
    Block AGC: one update per demodulated chunk.  Decay path is the pinned loop
    filter y = beta*x + (1-beta)*y (``sigs/agc.m:6-12``), attack is immediate."""

    def __init__(self, dtype=np.float32):
        self.rd = dtype
        self.ref = dtype(AGC_REF)
        self.beta = dtype(AGC_BETA)
        self.reset()

    def reset(self):
        rd = self.rd
        self.agc = rd(0)        # loop-filter output (smoothed peak)
        self.gain = rd(1)
        self.maxbuf = rd(0)
        self.err = rd(0)

    def update(self, peak, active):
        rd = self.rd
        peak = rd(peak)
        self.maxbuf = peak
        if peak > self.agc:
            self.agc = peak
        else:
            self.agc = rd(self.agc + rd(self.beta * rd(peak - self.agc)))
        if active:
            self.gain = rd(min(rd(self.ref / max(self.agc, rd(1e-12))), rd(AGC_GAIN_MAX)))
        else:
            self.gain = rd(1)
        self.err = rd(self.ref - rd(self.gain * peak))
        return self.gain


class Demodulator:
    """``rx.demod``: per-mode detector + AF filter at FS_OUT (``receiver.py:649,873-874``;
    ``gui.py:1704``).  Stateful across chunks: detector history, AF FIR history,
    BFO phase (absolute output index), PLL."""

    def __init__(self, fs_out, ntaps_af, dtype=np.float32):
        self.fs_out = float(fs_out)
        self.ntaps = int(ntaps_af)
        self.rd = dtype
        self.cd = np.complex64 if dtype == np.float32 else np.complex128
        self.filter_bank_real = af_filter_bank_real(fs_out, ntaps_af)
        self.filter_bank_cmpx = af_filter_bank_cmpx(fs_out, ntaps_af)
        self.am_pll = CarrierPLL(fs_out, dtype)
        self.yhist = np.zeros(self.ntaps + 1, self.cd)     # y[m-(ntaps+1)] .. y[m-1]
        self.vhist = np.zeros(self.ntaps - 1, self.cd)     # PLL outputs (AM-Synch only)
        self.m_abs = 0
        self.taps = None

    def set_taps(self, c):
        self.taps = np.asarray(c, np.complex128).astype(self.cd)

    def process(self, y, mode, bfo):
        """detector + AF FIR for the new samples ``y``.  The detector is re-applied to
        the kept y history, so a mode change takes effect on the whole AF-filter window
        (DESIGN.md 3.5); only the AM-Synch PLL output has a history of its own."""
        y = np.asarray(y, self.cd)
        n, hl = len(y), self.ntaps - 1
        ybuf = np.concatenate((self.yhist, y))            # ybuf[j] <-> output m_abs-(hl+2)+j
        if mode == "AM-Synch":
            v = self.am_pll.process(y).astype(self.cd)
            d = np.concatenate((self.vhist, v))
            self.vhist = d[len(d) - hl:]
        elif mode == "AM":
            d = np.abs(ybuf[2:]).astype(self.rd).astype(self.cd)
        elif mode == "NFM":
            scale = self.rd(self.fs_out / (2 * math.pi * NFM_FULL_SCALE_DEV))
            d = (nfm_discriminator(ybuf, self.rd) * scale).astype(self.cd)
        elif mode == "CW":
            fw, _ = freq_word(bfo, self.fs_out)
            idx = self.m_abs - hl + np.arange(hl + n, dtype=np.int64)
            ph = ((idx % TWO32) * fw) % TWO32
            d = (ybuf[2:] * phase_to_cplx(ph.astype(np.uint32), self.cd)).astype(self.cd)
        else:                       # SSB/USB/LSB/IQ/RTTY: the AF filter does the work
            d = ybuf[2:]
        a = np.convolve(d, self.taps, mode='valid') if n else d[:0]
        # out-of-band noise of the detector output (2nd difference = crude high-pass), for
        # the NFM noise squelch (sigs/squelch.m:92-145: HP envelope, one-pole smoothing)
        dr = d.real.astype(self.rd)
        self.last_hp = np.abs(dr[hl:] - self.rd(2) * dr[hl - 1:-1] + dr[hl - 2:-2]).astype(self.rd) if n else dr[:0]
        # ... and for the ratio squelch the detector output itself, with the history its two FIRs reach back over
        self.last_det = dr[hl - (SQ_RATIO_TAPS - 1):] if n else dr[:0]
        self.yhist = ybuf[len(ybuf) - (hl + 2):]
        self.m_abs += n
        return a.astype(self.cd)


class Receiver:
    """``dsp.Receiver(P, frq, irx, name, VIDEO_BWs, AF_BWs)`` (``receiver.py:65,835``).
    ``demod_data(x)`` = LO mix -> rational resample -> detector -> AF filter -> AGC;
    sets ``.am`` and ``.iq`` (``receiver.py:235,265``).

    ``frq`` is the offset of the wanted signal from the SDR centre; the LO is a
    ``signal_generator(-frq)`` so that ``rx.lo.change_freq(-FOFFSET)``
    (``gui.py:1907,1938``) and the ctor's ``+FOFFSET`` (``receiver.py:829-835``)
    agree."""

    def __init__(self, srate, fs_out_req, frq, mode="AM", ntaps_dec=1001, ntaps_af=255,
                 video_bw=10e3, af_bw=0.0, bfo=0.0, lsb=False, dtype=np.float32):
        self.rd = dtype
        self.cd = np.complex64 if dtype == np.float32 else np.complex128
        self.srate = float(srate)
        self.up, self.down, self.fs_out, self.in_chunk = chunk_sizes(srate, fs_out_req)
        self.mode, self.bfo, self.lsb = mode, float(bfo), lsb
        self.lo = NCO(-frq, srate, dtype)
        bank = dec_filter_bank(srate, self.up, self.fs_out, ntaps_dec, video_bw)
        self.video_idx = self._video_index(video_bw)
        self.dec = RationalDecimator(bank[self.video_idx], self.up, self.down, dtype)
        self.dec.filter_bank = bank
        self.demod = Demodulator(self.fs_out, ntaps_af, dtype)
        self.agc = AGC(dtype)
        self.af_bw = float(af_bw)
        self.af_idx = self._af_index(af_bw)
        self._retap()
        self.am = np.zeros(0, dtype)
        self.iq = np.zeros(0, self.cd)
        self.peak_in = dtype(0)
        self.mute_count = 0
        self.squelch = dtype(0)        # NFM noise-squelch threshold, 0 = off
        self.sq_level = dtype(0)
        self.sq_open = True
        self.squelch_ratio = dtype(0)  # the ratio squelch: open while sq1 / sq2 >= this; 0 = off (takes precedence when armed)
        self.sq_lp = dtype(0)          # sq1: envelope of the < 3 kHz part of the discriminator output
        self.sq_hp = dtype(0)          # sq2: envelope of the > 4 kHz part
        self._sq_taps = squelch_ratio_taps(self.fs_out, dtype)
        self.xhist = np.zeros(0, self.cd)

    @staticmethod
    def _video_index(video_bw):
        for i, lab in enumerate(VIDEO_BWs):
            if parse_bw(lab) == video_bw:
                return i
        return len(VIDEO_BWs) - 1

    @staticmethod
    def _af_index(af_bw):
        if not af_bw:
            return 0
        for i, lab in enumerate(AF_BWs):
            if parse_bw(lab) == af_bw:
                return i
        return 0

    def _retap(self):
        self.demod.set_taps(af_taps_for_mode(self.mode, self.af_idx, self.af_bw, self.bfo,
                                             self.fs_out, self.demod.ntaps, self.lsb))

    def set_mode(self, mode, af_bw=None, bfo=None):
        self.mode = mode
        if af_bw is not None:
            self.af_bw = float(af_bw)
            self.af_idx = self._af_index(af_bw)
        if bfo is not None:
            self.bfo = float(bfo)
        self._retap()

    def demod_data(self, x):
        x = np.asarray(x, self.cd)
        # The FIR history is kept as RAW samples and re-mixed with the current LO
        # (phase-continuous at the first new sample), so a retune applies the new
        # frequency to the whole FIR window of every later output.
        hl = self.dec.kmax - 1
        if len(self.xhist) != hl:
            self.xhist = np.zeros(hl, self.cd)
        raw = np.concatenate((self.xhist, x))
        idx = np.arange(-hl, len(x), dtype=np.int64)
        ph = (self.lo.phase + self.lo.fword * idx) % TWO32
        v = (raw * phase_to_cplx(ph.astype(np.uint32), self.cd)).astype(self.cd)
        self.lo.phase = (self.lo.phase + self.lo.fword * len(x)) % TWO32
        self.dec.hist = v[:hl]
        y = self.dec.process(v[hl:])
        self.xhist = raw[len(raw) - hl:] if hl else raw[:0]
        a = self.demod.process(y, self.mode, self.bfo)
        if self.mode == "IQ":
            peak = np.max(np.abs(a)) if len(a) else 0.0
        else:
            a = a.real.astype(self.rd)
            peak = np.max(np.abs(a)) if len(a) else 0.0
        g = self.agc.update(peak, self.mode in AGC_MODES)
        if self.mode == 'NFM' and self.squelch_ratio > 0 and len(a):
            # sigs/squelch.m:127-145: z1, z2 -> |.| -> filter(alpha, [1 alpha-1], .) per sample -> ratio; the gate follows
            # the ratio behind the chunk's last sample (a chunk is the AGC block: one gain per chunk)
            al = self.rd(SQ_RATIO_ALPHA)
            b, aa = np.array([al], self.rd), np.array([1, al - 1], self.rd)
            det = self.demod.last_det
            for name, taps in (("sq_lp", self._sq_taps[0]), ("sq_hp", self._sq_taps[1])):
                z = np.abs(np.convolve(det, taps, mode='valid').astype(self.rd))
                env, _ = signal.lfilter(b, aa, z, zi=np.array([(self.rd(1) - al) * getattr(self, name)], self.rd))
                setattr(self, name, self.rd(env[-1]))
            self.sq_open = bool(self.sq_lp >= self.rd(self.squelch_ratio) * self.sq_hp)
            if not self.sq_open:
                g = self.rd(0)
        elif self.mode == 'NFM' and self.squelch > 0 and len(a):
            noise = self.rd(np.sum(self.demod.last_hp.astype(np.float64)) / len(a))
            self.sq_level = self.rd(self.sq_level + self.rd(SQUELCH_ALPHA) * self.rd(noise - self.sq_level))
            self.sq_open = bool(self.sq_level <= self.rd(self.squelch))
            if not self.sq_open:
                g = self.rd(0)
        am = (a * g).astype(a.dtype)
        self.iq = y
        self.am = am
        return am

    def auto_mute(self, x, mute_chunks=1):
        """``rx.auto_mute(x)`` (``receiver.py:238-245``; ``params.py:446-450``): big-signal
        detector on the raw chunk, held for MUTE_CHUNKS chunks."""
        x = np.asarray(x, self.cd)
        p2 = np.max(x.real * x.real + x.imag * x.imag) if len(x) else 0.0
        self.peak_in = self.rd(p2)
        if p2 > AUTO_MUTE_THRESH * AUTO_MUTE_THRESH:
            self.mute_count = int(mute_chunks)
            return True
        if self.mute_count > 0:
            self.mute_count -= 1
            return True
        return False


# ------------------------------------------------------------------ PSD / waterfall
def psd_window(n):
    w = np.kaiser(n, PSD_KAISER_BETA)
    return w / np.sum(w)


class Spectrum:
    """``dsp.spectrum(fs_kHz, chunk_size, NFFT, overlap, TAG=)`` (``Plotting.py:376``;
    sizes ``gui.py:611-631``).  ``periodogram(y, True)`` slides ``len(y)`` new samples
    into a ``chunk_size`` buffer, windows (Kaiser 8.6, unit coherent gain), zero-pads
    to NFFT and returns 10*log10(re^2+im^2) (pin P5, ``rtty.py:839-841``): complex
    input -> NFFT bins fftshifted, real input -> NFFT/2 bins [0, fs/2)."""

    def __init__(self, fs, chunk_size, nfft, overlap, dtype=np.float32, TAG=''):
        self.fs = float(fs)
        self.chunk_size = int(chunk_size)
        self.NFFT = int(nfft)
        self.overlap = float(overlap)
        self.new_samps = int(round(self.chunk_size * (1.0 - self.overlap)))
        self.rd = dtype
        self.cd = np.complex64 if dtype == np.float32 else np.complex128
        self.win = psd_window(self.chunk_size).astype(dtype)
        self.buf = np.zeros(self.chunk_size, self.cd)
        self.df = self.fs / self.NFFT
        self.frq2 = (np.arange(self.NFFT) - self.NFFT // 2) * self.df
        self.frq = self.frq2.copy()
        self.TAG = TAG

    def periodogram(self, y, db=True):
        y = np.asarray(y)
        is_real = not np.iscomplexobj(y)
        n = len(y)
        if n == 0 or n > self.chunk_size:
            if n == 0:
                return np.zeros(0, self.rd)
            y = y[-self.chunk_size:]
            n = self.chunk_size
        self.buf = np.concatenate((self.buf[n:], y.astype(self.cd)))
        # a real input stream (AF PSD) is transformed as a real signal
        src = self.buf.real.astype(self.cd) if is_real else self.buf
        xw = (src * self.win).astype(self.cd)
        X = np.fft.fft(xw, self.NFFT).astype(self.cd)
        p = (X.real * X.real + X.imag * X.imag).astype(self.rd)
        if db:
            p = (10 * np.log10(p + self.rd(PSD_FLOOR))).astype(self.rd)
        if is_real:
            self.frq = np.arange(self.NFFT // 2) * self.df
            return p[:self.NFFT // 2]
        self.frq = self.frq2
        return np.fft.fftshift(p)


def waterfall_push(wf, line_db):
    """``Plotting.py:536-547``: shift the history left and append the new line
    (shorter lines are padded with -1e38, ``Plotting.py:385,540-541``)."""
    nfft = wf.shape[0]
    col = np.full((nfft, 1), -1e38, wf.dtype)
    col[:len(line_db), 0] = line_db
    return np.concatenate((wf[:, 1:], col), axis=1)


def waterfall_roll(wf, wf_fc, frq, df):
    """``Plotting.py:689-695`` retune roll."""
    nbins = int(float(frq - wf_fc) / df + 0.5)
    if nbins != 0:
        return np.roll(wf, -nbins, axis=0), frq
    return wf, wf_fc


def waterfall_image(wf, wf_cnt, pan_dr, npsd=None):
    """``Plotting.py:583-587,618-626``: background = median over bins of the mean over
    the valid history; image = max(wf[0:npsd] - bkgnd, zmax - PAN_DR), npsd = length of the
    line pushed last (:536, :618: the image and its maximum cover those rows only)."""
    psd2 = np.mean(wf[:, -wf_cnt:], 1)
    bkgnd = np.median(psd2)
    zz = wf[0:(wf.shape[0] if npsd is None else npsd), :] - bkgnd
    zmax = np.nanmax(zz)
    return np.maximum(zz, zmax - pan_dr), bkgnd, psd2


def find_peaks_db(psd2, bkgnd, peak_dist_bins):
    """``Plotting.py:594-602``."""
    peaks, _ = signal.find_peaks(psd2, distance=peak_dist_bins, height=bkgnd + 10)
    return peaks


def find_peaks_greedy(x, height, distance):
    """``scipy.signal.find_peaks(x, height=height, distance=distance)`` (what ``Plotting.py:596`` calls) written out as its
    sequential walk: local maxima with flat tops (the midpoint of a plateau, ``(left + right) // 2``; a flat stretch that
    touches either end of the line is no peak), ``x[peak] >= height``, then greedily by height -- every kept peak removes all
    peaks closer than ``ceil(distance)``.  The ONE choice SciPy leaves to an unstable ``np.argsort`` is made explicit: of
    equal heights the higher index ranks first (what a stable sort gives).  Pinned to SciPy's own output on the tie-free
    lines of ``tests/golden/peaks_ref.npz`` (``tests/test_oracle_pins.py``); the checker of the device kernel where ties
    closer than ``distance`` make SciPy's answer depend on NumPy's sort."""
    x = np.asarray(x)
    n = len(x)
    pk, i = [], 1
    while i < n - 1:
        if x[i - 1] < x[i]:
            j = i + 1
            while j < n - 1 and x[j] == x[i]:
                j += 1
            if x[j] < x[i]:
                pk.append((i + j - 1) // 2)
                i = j
        i += 1
    pk = np.array([p for p in pk if float(x[p]) >= height], np.int64)
    d = math.ceil(distance)
    keep = np.ones(len(pk), bool)
    for j in sorted(range(len(pk)), key=lambda q: (x[pk[q]], pk[q]), reverse=True):
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and pk[j] - pk[k] < d:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < len(pk) and pk[k] - pk[j] < d:
            keep[k] = False
            k += 1
    return pk[keep]


# ------------------------------------------------------------------ test signals
# The synthetic wideband IQ source and the benchmark configurations are not part of
# the algorithm under test; they live with the synthetic SDR device.
from pysdr_amd.synth import CONFIGS, synth_iq  # noqa: E402,F401


def make_receivers(cfg, dtype=np.float32, ntaps_af=255):
    return [Receiver(cfg['fs'], cfg['fs_out'], r['frq'], mode=r['mode'],
                     ntaps_dec=cfg['ntaps_dec'], ntaps_af=ntaps_af,
                     video_bw=r.get('video_bw', 10e3), af_bw=r.get('af_bw', 0.0),
                     bfo=r.get('bfo', 0.0), dtype=dtype) for r in cfg['rx']]
