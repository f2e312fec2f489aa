"""Numeric part of the reference's waterfall display (``three_box_plot.plot``,
``Plotting.py:536-626``; ``shift_waterfall`` :689-695) with the history kept on the GPU
(SURVEY.md 8(f) N1).  The Qt drawing stays with the caller; this returns what it blits."""
from __future__ import annotations

import ctypes as C
import sys

import math

import numpy as np

from . import _lib
from ._lib import check


class Waterfall:
    def __init__(self, nfft, ncols=100, device=0):
        _lib.require_gpu()
        self.nfft, self.ncols, self.device = int(nfft), int(ncols), device
        self.wf_cnt = 0                  # Plotting.py:386
        self.wf_fc = 0                   # Plotting.py:387
        h = C.c_void_p()
        check(_lib.lib().pysdr_waterfall_create(device, self.nfft, self.ncols, C.byref(h)),
              "pysdr_waterfall_create")
        self._h = h

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be torn down: leave the device
        # memory to process exit rather than call into a dead runtime
        if sys.is_finalizing():
            return
        try:
            if self._h:
                _lib.lib().pysdr_waterfall_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def push(self, PSD, flip=False):
        """``wf = concatenate((wf[:,1:], line), axis=1)`` (``Plotting.py:536-547``); ``flip`` is
        the ``RIG_IF<0`` ``np.flipud`` of :538-539."""
        line = np.ascontiguousarray(PSD[::-1] if flip else PSD, np.float32)
        check(_lib.lib().pysdr_waterfall_push(self._h, C.c_void_p(line.ctypes.data), len(line), 0),
              "pysdr_waterfall_push")
        self.npsd = len(line)            # Plotting.py:536
        if self.wf_cnt < self.ncols:
            self.wf_cnt += 1

    def shift_waterfall(self, frq, df):
        """``Plotting.py:689-695``: roll the history when the centre frequency moved."""
        nbins = int(float(frq - self.wf_fc) / df + 0.5)
        if nbins != 0:
            check(_lib.lib().pysdr_waterfall_roll(self._h, nbins), "pysdr_waterfall_roll")
            self.wf_fc = frq
        return nbins

    def image(self, pan_dr):
        """-> (image[npsd, ncols], bkgnd, PSD2): ``Plotting.py:583-587,618-626``.  As in the reference
        the image and its dynamic-range maximum cover the rows of the line pushed last
        (``zz = self.wf[0:npsd,:] - med``, :618): a half-length (real-input) line gives half an image."""
        img = np.empty((self.ncols, self.nfft), np.float32)
        mean = np.empty(self.nfft, np.float32)
        bk = C.c_float(0)
        npsd = int(getattr(self, 'npsd', self.nfft))
        check(_lib.lib().pysdr_waterfall_image_rows(self._h, float(pan_dr), npsd, _lib.as_pf(img), _lib.as_pf(mean),
                                                    C.byref(bk)), "pysdr_waterfall_image_rows")
        return img.T[:npsd], bk.value, mean

    def peaks(self, bkgnd, peak_dist, df, psd2=None):
        """``Plotting.py:594-602``: ``peaks, _ = signal.find_peaks(PSD2, distance=PEAK_DIST/df, height=bkgnd+10)`` on the
        device (``pysdr_waterfall_peaks``: flat tops, height, SciPy's priority-by-height distance rule) -> ascending bin
        indices.  Over the averaged line the last ``image()`` left on the device, or over ``psd2`` if one is given."""
        n = int(getattr(self, 'npsd', self.nfft)) if psd2 is None else len(psd2)
        dist = max(1, int(math.ceil(float(peak_dist) / float(df))))
        line = None
        if psd2 is not None:
            line = np.ascontiguousarray(psd2, np.float32)
            if n > self.nfft:
                raise ValueError(f"peaks: a line of {n} bins on a waterfall of {self.nfft}")
        else:
            n = self.nfft                      # the reference picks over all of PSD2 (Plotting.py:583,595)
        cap = self.nfft // 2 + 1
        idx = np.empty(cap, np.int32)
        cnt = C.c_int32(0)
        check(_lib.lib().pysdr_waterfall_peaks(self._h, None if line is None else _lib.as_pf(line), n, float(bkgnd) + 10.0, dist,
                                               idx.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(cnt)),
              "pysdr_waterfall_peaks")
        return idx[:cnt.value].astype(np.int64)
