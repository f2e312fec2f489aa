"""Wideband RTTY filterbank on the baseband-IQ tap: the FFT-heavy front of
the reference's RTTY process (``rtty.py:780-868``), fed with ``rx.iq`` like the reference feeds
its queue (``receiver.py:286-290``).  For every 22 ms symbol it produces four ``line``s, the
sliding Kaiser(8.6)-windowed, zero-padded FFT in dB, fftshifted and flipped
(``rtty.py:836-846``), computed in batches on the GPU through the spectrum entry points of the
C ABI (window / zero-pad / rocFFT / dB / fftshift kernels); ``mark_space`` picks the two bins a
decoder compares (``rtty.py:485-492``).  The Baudot symbol decoder itself is host logic of the
reference's GUI process and is not rebuilt here."""
from __future__ import annotations

import ctypes as C
import math
import sys

import numpy as np

from . import _lib
from ._lib import check


class RTTY_Params:
    """``rtty.py:376-404``."""

    def __init__(self, FS_OUT, mark_bins=(559,)):
        self.T = 22e-3
        self.FSK_SHIFT = 170
        self.SAMPS_PER_BIT = 4
        STOP_BITS = 1.5
        self.M = int(4 * (1 + 5 + STOP_BITS))
        self.N = int(round(self.T * FS_OUT))
        self.NFFT = 1 << int(math.ceil(math.log2(self.N)))
        NSTEP = self.N / 4.
        self.NSTART = [int(NSTEP * i + 0.5) for i in range(4)]
        bin_size = FS_OUT / float(self.NFFT)
        self.NBINS = int(round(self.FSK_SHIFT / bin_size))
        self.frq = np.fft.fftshift(np.fft.fftfreq(self.NFFT, d=1000. / FS_OUT)) + 0
        self.mark_bins = np.array(mark_bins)


class RTTY_Filterbank:
    def __init__(self, FS_OUT, max_symbols=512, device=0, mark_bins=(559,)):
        _lib.require_gpu()
        self.RTTY = RTTY_Params(FS_OUT, mark_bins)
        p = self.RTTY
        self.device = device
        self.max_symbols = int(max_symbols)
        self.window = np.kaiser(p.N, 8.6)
        self._L = _lib.lib()
        h = C.c_void_p()
        win = np.ascontiguousarray(self.window, np.float32)
        # the four quarter-symbol lines of a symbol start N/4 apart: when that is a whole number of
        # samples (48 kHz: N = 1056) all 4k lines of k symbols are ONE batch with hop N/4
        self._uniform = all(int(p.NSTART[i]) * 4 == i * p.N for i in range(4))
        check(self._L.pysdr_spectrum_create(device, p.N, p.NFFT, 4 * self.max_symbols, _lib.as_pf(win),
                                            C.byref(h)), "pysdr_spectrum_create")
        self._h = h
        self._d_in = C.c_void_p()
        self._d_out = C.c_void_p()
        check(self._L.pysdr_dev_alloc(device, (self.max_symbols + 1) * p.N * 8, C.byref(self._d_in)), "alloc")
        check(self._L.pysdr_dev_alloc(device, 4 * self.max_symbols * p.NFFT * 4, C.byref(self._d_out)), "alloc")
        self._fifo = np.zeros(0, np.complex64)
        self._prev = None

    def close(self):
        if self._h:
            self._L.pysdr_spectrum_destroy(self._h)
            self._L.pysdr_dev_free(self.device, self._d_in)
            self._L.pysdr_dev_free(self.device, self._d_out)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def push(self, iq):
        """Append baseband IQ; returns the lines [4*k, NFFT] of the k symbols completed by it
        (the very first symbol only primes the overlap, ``rtty.py:826-829``)."""
        p = self.RTTY
        self._fifo = np.concatenate((self._fifo, np.ascontiguousarray(iq, np.complex64)))
        out = []
        while True:
            k = len(self._fifo) // p.N
            if self._prev is None:
                if k == 0:
                    break
                self._prev, self._fifo = self._fifo[:p.N].copy(), self._fifo[p.N:]
                continue
            if k == 0:
                break
            k = min(k, self.max_symbols)
            cur, self._fifo = self._fifo[:k * p.N], self._fifo[k * p.N:]
            x = np.ascontiguousarray(np.concatenate((self._prev, cur)))
            check(self._L.pysdr_dev_upload(self.device, self._d_in, C.c_void_p(x.ctypes.data), x.nbytes), "upload")
            lines = np.empty((4 * k, p.NFFT), np.float32)
            if self._uniform:
                # one launch sequence, one download: frame 4 s + i starts at s N + i N/4
                check(self._L.pysdr_spectrum_batch(self._h, self._d_in, 4 * k, p.N // 4, self._d_out), "spectrum_batch")
                check(self._L.pysdr_spectrum_sync(self._h), "spectrum_sync")
                check(self._L.pysdr_dev_download(self.device, C.c_void_p(lines.ctypes.data), self._d_out, lines.nbytes),
                      "download")
                out.append(lines[:, ::-1].copy())                # np.flipud of the shifted spectrum
                self._prev = cur[-p.N:].copy()
                continue
            tmp = np.empty((k, p.NFFT), np.float32)
            for i in range(4):
                # frames of quarter i: start NSTART[i] + s*N, s = 0..k-1
                src = C.c_void_p(self._d_in.value + p.NSTART[i] * 8)
                check(self._L.pysdr_spectrum_batch(self._h, src, k, p.N, self._d_out), "spectrum_batch")
                check(self._L.pysdr_spectrum_sync(self._h), "spectrum_sync")
                check(self._L.pysdr_dev_download(self.device, C.c_void_p(tmp.ctypes.data), self._d_out, tmp.nbytes),
                      "download")
                lines[i::4] = tmp[:, ::-1]                       # np.flipud of the shifted spectrum
            out.append(lines)
            self._prev = cur[-p.N:].copy()
        if not out:
            return np.zeros((0, p.NFFT), np.float32)
        return np.concatenate(out)

    def mark_space(self, lines, mark_bin=None):
        """``rtty.py:485-492``: (mark, space) per line; ``signal = mark - space``."""
        mb = int(self.RTTY.mark_bins[0] if mark_bin is None else mark_bin)
        return lines[:, mb], lines[:, mb + self.RTTY.NBINS]
