"""CPU oracle of the wideband RTTY filterbank.  TEST INFRASTRUCTURE ONLY --
see oracle/__init__.py.  Unlike the receiver chain this consumer is fully specified in-tree, so
the restatement follows the reference line by line:

  rtty.py:376-404  RTTY_Params: T = 22 ms, N = round(T*FS_OUT) samples per symbol,
                   NFFT = 2^nextpow2(N), NSTART[i] = int(N/4*i + 0.5), NBINS = round(170/(FS_OUT/NFFT))
  rtty.py:822-846  RTTY_Executive.run: window = np.kaiser(N, 8.6); pull N samples at a time, the
                   first pull only primes `prev`; x = [prev, iq]; for i in 0..3:
                   X = fftshift(fft(x[NSTART[i]:NSTART[i]+N] * window, NFFT));
                   line = flipud(10*log10(re^2 + im^2))
  rtty.py:485-492  RTTY_Decoder.decode: mark = line[mark_bin], space = line[mark_bin + NBINS],
                   signal = mark - space
The symbol decoder behind it (matched filter over 32 Baudot symbols, timing, SNR gate) is host
logic of the reference's GUI process and stays out of scope."""
from __future__ import annotations

import math

import numpy as np


class RttyParams:
    def __init__(self, fs_out):
        self.T = 22e-3
        self.FSK_SHIFT = 170
        self.SAMPS_PER_BIT = 4
        self.M = int(4 * (1 + 5 + 1.5))
        self.N = int(round(self.T * fs_out))
        self.NFFT = 1 << int(math.ceil(math.log2(self.N)))
        nstep = self.N / 4.0
        self.NSTART = [int(nstep * i + 0.5) for i in range(4)]
        self.NBINS = int(round(self.FSK_SHIFT / (fs_out / float(self.NFFT))))
        self.frq = np.fft.fftshift(np.fft.fftfreq(self.NFFT, d=1000.0 / fs_out))


class RttyFilterbank:
    """push(iq) -> the `line`s (one row per quarter symbol) the reference hands its decoders."""

    def __init__(self, fs_out, dtype=np.float64):
        self.p = RttyParams(fs_out)
        self.window = np.kaiser(self.p.N, 8.6)
        self.dtype = dtype
        self.fifo = np.zeros(0, np.complex128)
        self.prev = None

    def push(self, iq):
        p = self.p
        self.fifo = np.concatenate((self.fifo, np.asarray(iq, np.complex128)))
        lines = []
        while len(self.fifo) >= p.N:              # rb.ready(NFFT) gates the pull; see note below
            cur, self.fifo = self.fifo[:p.N], self.fifo[p.N:]
            if self.prev is None:
                self.prev = cur
                continue
            x = np.concatenate((self.prev, cur))
            for i in range(4):
                xx = x[p.NSTART[i]:p.NSTART[i] + p.N]
                X = np.fft.fftshift(np.fft.fft(xx * self.window, p.NFFT))
                XX = 10 * np.log10(np.square(X.real) + np.square(X.imag))
                lines.append(np.flipud(XX))
            self.prev = cur
        if not lines:
            return np.zeros((0, p.NFFT), self.dtype)
        return np.asarray(lines, self.dtype)

    # The reference only pulls when the ring holds NFFT (> N) samples (rtty.py:816); that delays
    # WHEN a symbol is processed, not what is computed from it, so the oracle pulls as soon as N
    # samples are there and flush() is not needed for parity of the lines.


def mark_space(lines, mark_bin, nbins):
    """rtty.py:485-492 for a block of lines."""
    lines = np.asarray(lines)
    return lines[:, mark_bin], lines[:, mark_bin + nbins]


def synth_rtty(fs_out, nsym, mark_hz, shift_hz=170.0, seed=0, noise=1e-3):
    """45.45-baud FSK test signal at baseband: random bits, one per 22 ms."""
    rng = np.random.default_rng(seed)
    n = int(round(22e-3 * fs_out)) * nsym
    bits = rng.integers(0, 2, nsym)
    per = int(round(22e-3 * fs_out))
    f = np.repeat(np.where(bits > 0, mark_hz, mark_hz - shift_hz), per)[:n]
    ph = 2 * np.pi * np.cumsum(f) / fs_out
    x = 0.5 * np.exp(1j * ph) + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64), bits
