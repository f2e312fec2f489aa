"""The fields of the reference's shared run-time parameter object ``P`` that the receiver
hot path reads (``params.py:38-486``), built from keyword arguments instead of argparse.
Only the arithmetic is restated (``params.py:245-281,318-329,399-472``); rig/GUI/CLI
fields are out of scope.  Any object with these attributes works with
``pysdr_amd.sig_proc`` -- the reference's own ``RUN_TIME_PARAMS`` instance included."""
from __future__ import annotations

import numpy as np

from . import rates


class RunTimeParams:
    def __init__(self, fs=1e6, fsout=48e3, fc=(0.0,), mode='AM', foffset=100e3, nfilt=1001,
                 vid_bw=0.0, af_bw=0.0, bfo=0.0, duration=1e38, sdr_type='sdrplay',
                 auto_mute=False, src=None, audio=1, af_filt_len=255, device=0,
                 max_batch_chunks=1, overlap_calls=True):
        self.SRATE = float(fs)
        self.SDR_TYPE = sdr_type
        fc = np.atleast_1d(np.asarray(fc, np.float64))
        self.MAX_RX = rates.MAX_RX
        if len(fc) > rates.MAX_RX:                      # params.py:267-274
            fc = fc[:rates.MAX_RX]
        self.NUM_RX = len(fc)
        self.FC = fc
        self.MODE = mode
        self.NEW_MODE = mode
        self.MODE_CHANGE = False
        self.FREQ_CHANGE = False
        self.FOFFSET = float(foffset)
        s = np.atleast_1d(np.asarray(src if src is not None else [], np.int64))
        self.SOURCE = np.concatenate((s, -np.ones(max(0, self.NUM_RX - len(s)), np.int64)))
        self.AUDIO_SCHEME = audio
        self.NUM_PLAYERS = self.NUM_RX if audio == 1 else int((self.NUM_RX + 1) / 2)
        self.rx = self.NUM_RX * [None]
        if self.FOFFSET == 0:                           # params.py:311-316
            fo = 0.5 * (max(fc) + min(fc))
            self.FOFFSET = fo - max(fc)
        self.BFO = float(bfo)
        if self.MODE == 'CW' and self.BFO == 0:         # params.py:318-320
            self.BFO = 700
        self.DURATION = duration
        self.VIDEO_BW = float(vid_bw)
        if self.VIDEO_BW == 0:                          # params.py:324-329
            self.VIDEO_BW = 200e3 if self.MODE == 'WFM' else 10e3
        self.FILT_LEN = int(nfilt)
        self.AF_BW = float(af_bw)
        self.AF_FILTER_NUM = None
        self.AF_GAIN = 0.5
        self.MUTED = rates.MAX_RX * [False]
        d = rates.derive(self.SRATE, fsout)             # params.py:405-406,440-444
        self.UP, self.DOWN = d['UP'], d['DOWN']
        self.FS_OUT = d['FS_OUT']
        self.OUT_CHUNK_SIZE = rates.OUT_CHUNK_SIZE
        self.IN_CHUNK_SIZE = d['IN_CHUNK_SIZE']
        self.ENABLE_AUTO_MUTE = auto_mute               # params.py:446-450
        self.MUTE_TIME = .25
        self.MUTE_CHUNKS = int(self.MUTE_TIME * self.FS_OUT / self.OUT_CHUNK_SIZE)
        self.AUTO_MUTED = False
        self.RB_SIZE = rates.ring_buffer_size(self.NUM_RX, sdr_type, self.FS_OUT)
        self.DELAY = self.OUT_CHUNK_SIZE
        self.FOFFSET = rates.adjust_foffset(self.FOFFSET, self.SRATE, self.RB_SIZE)  # :470-472
        # flags the executive reads
        self.MP_SCHEME = 1
        self.REPLAY_MODE = False
        self.USE_FAKE_RTL = False
        self.SHOW_RF_PSD = False
        self.SHOW_AF_PSD = False
        self.SHOW_BASEBAND_PSD = False
        self.PANADAPTOR = False
        self.PLOT_RX = 0
        self.SAVE_IQ = self.SAVE_BASEBAND = self.SAVE_DEMOD = False
        self.ENABLE_RTTY = False
        self.LOOPBACK = False
        self.AUX_AUDIO = False
        self.audio_playback = True
        self.RX_DONE = False
        self.Stopper = None
        self.evt = None
        self.gui = None
        self.sdr = None
        self.rxStream = None
        self.players = []
        self.nchunks = 0
        # build-specific knobs (not in the reference)
        self.AF_FILT_LEN = int(af_filt_len)
        self.GPU_DEVICE = int(device)
        self.MAX_BATCH_CHUNKS = int(max_batch_chunks)
        self.OVERLAP_CALLS = bool(overlap_calls)         # batch contexts: two overlapped halves per call (pysdr_set_overlap)

    def rx_offset(self, irx):
        """LO offset of sub-receiver ``irx`` (``receiver.py:829-834``)."""
        if self.SOURCE[irx] >= 0:
            return self.FC[irx] - self.FC[self.SOURCE[irx]]
        return self.FOFFSET + self.FC[irx] - self.FC[0]
