"""A ``dsp`` module stand-in backed by the CPU oracle, so the executive's HOST logic
(chunk assembly, DC removal, audio routing) can be tested without a GPU.  Test
infrastructure only."""
import numpy as np

from oracle import sdr_oracle as so
from pysdr_amd.sig_proc import ring_buffer2  # noqa: F401  (plain Python FIFO, no GPU)


class Receiver:
    def __init__(self, P, frq, irx, name, VIDEO_BWs=None, AF_BWs=None, dtype=np.float32):
        self.P = P
        self.mode = None
        self.af_bw = None
        self.bfo = None
        self._rx = so.Receiver(P.SRATE, P.FS_OUT, frq, mode=P.MODE, ntaps_dec=P.FILT_LEN,
                               ntaps_af=P.AF_FILT_LEN, video_bw=P.VIDEO_BW, af_bw=P.AF_BW,
                               bfo=P.BFO, dtype=dtype)
        self.agc = self._rx.agc
        self.demod = self._rx.demod
        self.lo = self._rx.lo
        self.am = np.zeros(0, dtype)
        self.iq = np.zeros(0, np.complex64)

    def demod_data(self, x):
        mode = self.mode if self.mode is not None else self.P.MODE
        if mode != self._rx.mode or (self.af_bw is not None and self.af_bw != self._rx.af_bw):
            self._rx.set_mode(mode, af_bw=self.af_bw, bfo=self.bfo)
        self.am = self._rx.demod_data(x)
        self.iq = self._rx.iq
        return self.am

    def auto_mute(self, x):
        return self._rx.auto_mute(x, getattr(self.P, 'MUTE_CHUNKS', 1))

    @property
    def peak_in(self):
        return float(self._rx.peak_in)
