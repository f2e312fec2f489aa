"""Record / replay files: the ``fileio.sdr_fileio`` object (``import
fileio``, ``receiver.py:41``; ``import fileio as io``, ``pySDR.py:73``) of the
reference's ``pySDR.py:118-123`` (writers ``raw_iq`` / ``baseband_iq`` / ``demod``),
``receiver.py:293-297,759-761`` (``save_data`` taps), ``receiver.py:808-822`` (replay:
``.srate``, ``.fc``, ``read_data()``) and ``gui.py:1185-1219`` (``close()`` on toggle).

The reference's own module lives in the absent ``aa2il/libs``; what the tree pins is the
calling convention above, the file names (``demod_20190321_225218.dat``,
``baseband_iq_20190413_221346.dat``: ``<name>_<YYYYmmdd>_<HHMMSS>.dat``, ``sigs/nfm.m:41-44``)
and what its Octave reader returns: ``[y, hdr, str] = read_sdr_data(fname)`` with
``hdr(1) = fs``, ``hdr(4) = nchan`` and a text tag (``sigs/nfm.m:50-55``,
``sigs/sdr2wav.m:37-43``).  The byte layout is this build's (parity unpinned):

    bytes 0..7    magic  b"PYSDRIQ1"
    float64 hdr[8] (little endian): [srate, fc, foffset, nchan, tag_bytes, 0, 0, 0]
    tag_bytes of ASCII tag, zero-padded to a multiple of 8
    float32 samples, channel-interleaved (nchan = 2: re, im, re, im, ...)

so ``hdr[0] = fs`` and ``hdr[3] = nchan`` as the Octave scripts expect.  Beyond the
reference's whole-file ``read_data()``, ``read_chunk(n)`` / ``chunks(n)`` stream the file
through a memory map, which is what lets recorded IQ drive the batched GPU path at its own
rate instead of the Python loop's."""
from __future__ import annotations

import os
import time

import numpy as np

MAGIC = b"PYSDRIQ1"
NHDR = 8


class sdr_fileio:
    def __init__(self, fname, rw, P=None, nchan=2, tag='', out_dir=None):
        if rw not in ('r', 'w'):
            raise ValueError("sdr_fileio: rw must be 'r' or 'w'")
        self.rw = rw
        self.P = P
        self.nchan = int(nchan)
        self.tag = str(tag)
        self.fp = None
        self.fname = None
        self.nwritten = 0
        self._pos = 0
        if rw == 'w':
            # opened lazily by the first save_data(): the GUI creates the writers up front and
            # only some of them are ever switched on (gui.py:1185-1219)
            self.base = str(fname)
            self.out_dir = out_dir
            self.srate = None
            self.fc = None
            return
        self.fname = str(fname)
        with open(self.fname, 'rb') as f:
            if f.read(len(MAGIC)) != MAGIC:
                raise ValueError("%s: not a pySDR record file" % self.fname)
            hdr = np.frombuffer(f.read(8 * NHDR), '<f8')
            if len(hdr) != NHDR:
                raise ValueError("%s: truncated header" % self.fname)
            ntag = int(hdr[4])
            pad = (-ntag) % 8
            raw = f.read(ntag + pad)
            if len(raw) != ntag + pad:
                raise ValueError("%s: truncated tag" % self.fname)
            self.tag = raw[:ntag].decode('ascii', 'replace')
            self._data_off = f.tell()
        self.hdr = hdr.copy()
        self.srate = float(hdr[0])
        self.fc = float(hdr[1])
        self.foffset = float(hdr[2])
        self.nchan = int(hdr[3])
        if self.nchan not in (1, 2):
            raise ValueError("%s: nchan = %d" % (self.fname, self.nchan))
        nfloat = (os.path.getsize(self.fname) - self._data_off) // 4
        nfloat -= nfloat % self.nchan
        self._map = np.memmap(self.fname, '<f4', 'r', self._data_off, (nfloat,)) if nfloat else np.zeros(0, '<f4')
        self.nsamples = nfloat // self.nchan

    # ---- writing -----------------------------------------------------------------------
    def _rate_for(self):
        P = self.P
        # raw IQ is at SRATE, everything behind the decimator at FS_OUT (receiver.py:293-297)
        if self.base.startswith('raw'):
            return float(getattr(P, 'SRATE', 0.0) or 0.0)
        return float(getattr(P, 'FS_OUT', 0.0) or 0.0)

    def _open_w(self):
        stamp = time.strftime('%Y%m%d_%H%M%S', time.gmtime())
        name = '%s_%s.dat' % (self.base, stamp)
        d = self.out_dir if self.out_dir is not None else getattr(self.P, 'SAVE_DIR', None)
        self.fname = os.path.join(d, name) if d else name
        P = self.P
        self.srate = self._rate_for()
        fc = getattr(P, 'FC', [0.0]) if P is not None else [0.0]
        self.fc = float(fc[0] if np.ndim(fc) else fc)
        tag = self.tag.encode('ascii', 'replace')
        hdr = np.zeros(NHDR, '<f8')
        hdr[0], hdr[1], hdr[2] = self.srate, self.fc, float(getattr(P, 'FOFFSET', 0.0) or 0.0)
        hdr[3], hdr[4] = self.nchan, len(tag)
        self.fp = open(self.fname, 'wb')
        self.fp.write(MAGIC)
        self.fp.write(hdr.tobytes())
        self.fp.write(tag + b'\0' * ((-len(tag)) % 8))

    def save_data(self, x, VERBOSITY=0):
        if self.rw != 'w':
            raise IOError("sdr_fileio: file was opened for reading")
        if self.fp is None:
            self._open_w()
        x = np.asarray(x)
        if self.nchan == 2:
            buf = np.ascontiguousarray(x, np.complex64).view(np.float32)
        else:
            buf = np.ascontiguousarray(x.real if np.iscomplexobj(x) else x, np.float32)
        self.fp.write(buf.astype('<f4', copy=False).tobytes())
        self.nwritten += len(x)
        if VERBOSITY > 0:
            print('sdr_fileio: wrote', len(x), 'samples to', self.fname)

    def close(self):
        if self.fp is not None:
            self.fp.close()
            self.fp = None

    # ---- reading -----------------------------------------------------------------------
    def _view(self, start, stop):
        seg = self._map[start * self.nchan:stop * self.nchan]
        if self.nchan == 2:
            return np.array(seg, np.float32).view(np.complex64)
        return np.array(seg, np.float32)

    def read_data(self):
        """The whole recording (``receiver.py:526``)."""
        if self.rw != 'r':
            raise IOError("sdr_fileio: file was opened for writing")
        return self._view(0, self.nsamples)

    def read_chunk(self, n):
        """The next n samples, or None when fewer than n are left (the reference drops the
        tail of a replay the same way, ``receiver.py:543-557``)."""
        if self._pos + n > self.nsamples:
            return None
        out = self._view(self._pos, self._pos + n)
        self._pos += n
        return out

    def chunks(self, n):
        while True:
            x = self.read_chunk(n)
            if x is None:
                return
            yield x

    def rewind(self):
        self._pos = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


SDR_FILEIO = sdr_fileio          # the name sigs/iq.py:11,59 imports


def open_writers(P, out_dir=None):
    """``pySDR.py:118-123``: the three writers hang off P."""
    P.raw_iq_io = sdr_fileio('raw_iq', 'w', P, 2, 'RAW_IQ', out_dir)
    P.baseband_iq_io = sdr_fileio('baseband_iq', 'w', P, 2, 'BASEBAND_IQ', out_dir)
    P.demod_io = sdr_fileio('demod', 'w', P, 2 if P.MODE == 'IQ' else 1, P.MODE, out_dir)
    return P


def open_replay(P, dsp=None):
    """``receiver.py:808-822``: a recording replaces the radio.  Sets the rates the way the
    reference does (a ``baseband_iq`` recording is already at FS_OUT) and the tuning-offset
    generator ``P.lo`` that ``read_chunk`` applies (``receiver.py:552-553``)."""
    if dsp is None:
        from . import sig_proc as dsp
    from .rates import up_dn
    P.sdr = sdr_fileio(P.REPLAY, 'r', P)
    P.REPLAY_MODE = True
    P.SRATE = P.sdr.srate
    P.REPLAY_FC = P.sdr.fc
    P.FC[0] = P.sdr.fc
    # the reference tests `P.REPLAY.find('baseband_iq')`, which is true for every name that
    # does NOT start with it; the intent (a baseband recording is not decimated again) is
    # what is implemented here
    if 'baseband_iq' in os.path.basename(str(P.REPLAY)):
        P.FS_OUT = P.SRATE
    P.UP, P.DOWN = up_dn(P.SRATE, P.FS_OUT)
    P.FS_OUT = int(P.SRATE * P.UP / P.DOWN)
    P.IN_CHUNK_SIZE = int(P.OUT_CHUNK_SIZE * P.DOWN / float(P.UP))
    P.lo = dsp.signal_generator(0 * P.BFO, P.IN_CHUNK_SIZE, P.SRATE, True)
    return P
