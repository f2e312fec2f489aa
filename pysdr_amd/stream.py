"""SoapySDR-shaped synthetic / replay device: the stream-callback surface the receiver
executive drives (reference: the method set of ``utils.py:122-273`` and the canonical
loop of ``soapy.py:33-48`` -- ``setupStream`` / ``activateStream`` / ``readStream`` returning
an object whose ``.ret`` is the sample count or a negative error / ``deactivateStream`` /
``closeStream``).  ``readStream`` deliberately returns SHORT reads so the executive's
``xold`` carry logic (``receiver.py:586-622``) is exercised.  NumPy only."""
from __future__ import annotations

import numpy as np

from .synth import synth_iq

SOAPY_SDR_RX = 1
SOAPY_SDR_CF32 = 'CF32'
SOAPY_SDR_TIMEOUT = -1


class StreamResult:
    def __init__(self, ret, flags=0, timeNs=0):
        self.ret = ret
        self.flags = flags
        self.timeNs = timeNs

    def __repr__(self):
        return f"ret={self.ret}, flags={self.flags}, timeNs={self.timeNs}"


class Range(list):
    """What ``getGainRange`` / ``getFrequencyRange`` hand back: SoapySDR's range object as the reference reads it
    (``r.minimum(), r.maximum(), r.step()``, ``receiver.py:328-330``) that is also the plain ``[min, max, step]`` list its own
    RTL stand-in returns (``utils.py:175-178``)."""

    def __init__(self, lo, hi, step=0.0):
        super().__init__([lo, hi, step])

    def minimum(self):
        return self[0]

    def maximum(self):
        return self[1]

    def step(self):
        return self[2]


class SynthSDR:
    """``data`` is either a synth config dict (``pysdr_amd.synth.CONFIGS[...]``) or a
    complex64 array to replay.  ``read_pattern`` = fractions of the requested count handed
    out per ``readStream`` call (0 -> a timeout with no samples)."""

    def __init__(self, data, seed=1, nsamp=None, read_pattern=(1.0, 0.37, 0.0, 0.81, 0.5)):
        if isinstance(data, dict):
            self.cfg = data
            self.fs = float(data['fs'])
            self.samples = synth_iq(data, int(nsamp), seed)
        else:
            self.cfg = None
            self.samples = np.ascontiguousarray(data, np.complex64)
            self.fs = 0.0
        self.pos = 0
        self.pattern = tuple(read_pattern)
        self.ncall = 0
        self.active = False
        self.freq = {}
        self.gain = {}
        self.settings = {}
        self.key = 'synth'

    # -- configuration surface (utils.py:122-216)
    def setSampleRate(self, rx, ch, fs):
        self.fs = float(fs)

    def getSampleRate(self, rx, ch):
        return self.fs

    def listSampleRates(self, rx, ch):
        return [self.fs]

    def setFrequency(self, rx, ch, tag, f=None):
        if f is None:
            tag, f = 'RF', tag
        self.freq[tag] = float(f)

    def getFrequency(self, rx, ch, tag='RF'):
        return self.freq.get(tag, 0.0)

    def getNumChannels(self, rx):
        return 1

    def listGains(self, rx, ch):
        return []

    def setGain(self, rx, ch, stage, gain=None):
        self.gain[stage] = gain

    def getGain(self, rx, ch, stage=None):
        return self.gain.get(stage, 0)

    def getGainRange(self, rx, ch, stage=None):
        return Range(0.0, 49.6, 1.0)             # the span the reference's RTL stand-in clamps to (utils.py:204-205)

    def hasGainMode(self, rx, ch):
        return True

    def setGainMode(self, rx, ch, flag):
        self.gain['auto'] = flag

    def getGainMode(self, rx, ch):
        return bool(self.gain.get('auto', False))

    def getFrequencyRange(self, rx, ch, tag='RF'):
        return [Range(0.0, 6e9)]

    def listAntennas(self, rx, ch):
        return ['RX']

    def setAntenna(self, rx, ch, ant):
        self.settings['antenna'] = ant

    def getAntenna(self, rx, ch):
        return self.settings.get('antenna', 0)

    def listBandwidths(self, rx, ch):
        return []

    def setBandwidth(self, rx, ch, bw):
        self.settings['bandwidth'] = float(bw)

    def getBandwidth(self, rx, ch):
        return self.settings.get('bandwidth', self.fs)

    def writeSetting(self, key, val):
        self.settings[key] = val

    def readSetting(self, key):
        return self.settings.get(key)

    def getSettingInfo(self):
        return []

    def getDriverKey(self):
        return self.key

    def getHardwareKey(self):
        return 'synthetic'

    def getHardwareInfo(self):
        return []

    # -- stream surface (utils.py:228-260, soapy.py:33-48)
    def setupStream(self, rx, fmt, channels=None):
        self.fmt = fmt
        return 0

    def activateStream(self, stream):
        self.active = True

    def deactivateStream(self, stream):
        self.active = False

    def closeStream(self, stream):
        self.active = False

    def exhausted(self):
        return self.pos >= len(self.samples)

    def readStream(self, stream, buffs, n, timeoutUs=100000):
        if not self.active:
            return StreamResult(SOAPY_SDR_TIMEOUT)
        frac = self.pattern[self.ncall % len(self.pattern)]
        self.ncall += 1
        k = int(min(n, len(self.samples) - self.pos) * frac)
        if k <= 0:
            return StreamResult(0)
        buffs[0][:k] = self.samples[self.pos:self.pos + k]
        self.pos += k
        return StreamResult(k)

    def readStreamRTL(self, stream, n):
        """The reference's own RTL stand-in hands out an ARRAY instead of filling a buffer (``utils.py:241-251``,
        ``receiver.py:598-600`` under ``P.USE_FAKE_RTL``): whatever is ready, possibly nothing."""
        buf = np.empty(int(n), np.complex64)
        sr = self.readStream(stream, [buf], int(n))
        return buf[:max(sr.ret, 0)]

    # -- replay surface (fileio.sdr_fileio.read_data, receiver.py:526)
    def read_data(self):
        return self.samples
