"""Record / replay files (pysdr_amd/fileio.py): header fields the reference's
Octave scripts read (sigs/nfm.m:50-55), the writer taps of receiver.py:293-297,759-761 and the
replay slicing of receiver.py:541-557, on CPU with the oracle as the DSP."""
import os

import numpy as np
import pytest

from oracle import sdr_oracle as so
from pysdr_amd import executive, fileio as file_io, stream
from tests import oracle_dsp
from tests.test_executive_host import make_P


def test_round_trip_header_and_data(tmp_path):
    P = make_P(so.CONFIGS['C1'], 1)
    x = so.synth_iq(so.CONFIGS['C1'], 5000, 3)
    w = file_io.sdr_fileio('raw_iq', 'w', P, 2, 'RAW_IQ', out_dir=str(tmp_path))
    assert w.fp is None                                     # opened by the first save_data
    w.save_data(x[:3000])
    w.save_data(x[3000:])
    w.close()
    name = os.path.basename(w.fname)
    assert name.startswith('raw_iq_') and name.endswith('.dat') and len(name) == len('raw_iq_20190321_225218.dat')
    r = file_io.sdr_fileio(w.fname, 'r', P)
    assert r.hdr[0] == P.SRATE and r.hdr[3] == 2            # hdr(1) = fs, hdr(4) = nchan
    assert (r.srate, r.fc, r.nchan, r.tag, r.nsamples) == (P.SRATE, P.FC[0], 2, 'RAW_IQ', 5000)
    assert np.array_equal(r.read_data(), x)
    got = list(r.chunks(2048))
    assert [len(c) for c in got] == [2048, 2048] and np.array_equal(np.concatenate(got), x[:4096])
    assert r.read_chunk(2048) is None                       # the tail is dropped, as in replay


def test_real_channel_file_and_rates(tmp_path):
    P = make_P(so.CONFIGS['C1'], 1)
    a = np.linspace(-1, 1, 1000).astype(np.float32)
    w = file_io.sdr_fileio('demod', 'w', P, 1, P.MODE, out_dir=str(tmp_path))
    w.save_data(a + 0j)                                     # complex in, real part kept
    w.close()
    r = file_io.sdr_fileio(w.fname, 'r')
    assert r.nchan == 1 and r.srate == P.FS_OUT and r.tag == 'AM'
    assert np.array_equal(r.read_data(), a)
    with pytest.raises(ValueError):
        bad = tmp_path / 'x.dat'
        bad.write_bytes(b'not a record')
        file_io.sdr_fileio(str(bad), 'r')


def test_record_then_replay_reproduces_the_live_run(tmp_path):
    """Run live with SAVE_IQ / SAVE_BASEBAND / SAVE_DEMOD, replay the raw file: same audio."""
    cfg = so.CONFIGS['C1']
    nchunks = 4
    P = make_P(cfg, nchunks)
    L = P.IN_CHUNK_SIZE
    P.sdr = stream.SynthSDR(cfg, seed=31, nsamp=(nchunks + 1) * L)
    file_io.open_writers(P, out_dir=str(tmp_path))
    P.SAVE_IQ = P.SAVE_BASEBAND = P.SAVE_DEMOD = True
    live = []
    ex = executive.SDR_EXECUTIVE(P, dsp=oracle_dsp)
    ex.Run(on_chunk=lambda e: live.append(np.array(P.rx[0].am)))
    for io in (P.raw_iq_io, P.baseband_iq_io, P.demod_io):
        io.close()
    assert P.raw_iq_io.nwritten == nchunks * L
    demod = file_io.sdr_fileio(P.demod_io.fname, 'r').read_data()
    # the saved demod is the DC-free copy (receiver.py:250-252,296-297; P.MODE = 'AM'), the audio is rx.am
    assert np.array_equal(demod, np.concatenate([a - np.mean(a) for a in live]).astype(np.float32))
    bb = file_io.sdr_fileio(P.baseband_iq_io.fname, 'r')
    assert bb.srate == P.FS_OUT and bb.nsamples == len(demod)

    # replay: one chunk fewer than recorded is READ (strict '<' of receiver.py:543) -- and, as in the reference's
    # Run loop (receiver.py:715-740), the pass in which read_chunk finds the recording exhausted still
    # demodulates the stale buffer once more before the loop ends
    P2 = make_P(cfg, nchunks)
    P2.REPLAY = P.raw_iq_io.fname
    file_io.open_replay(P2, dsp=type('D', (), {'signal_generator': lambda *a: type('L', (), {'fo': 0.0})()}))
    assert (P2.SRATE, P2.UP, P2.DOWN, P2.IN_CHUNK_SIZE) == (P.SRATE, P.UP, P.DOWN, L)
    rep = []
    ex2 = executive.SDR_EXECUTIVE(P2, dsp=oracle_dsp)
    ex2.Run(on_chunk=lambda e: rep.append(np.array(P2.rx[0].am)))
    assert len(rep) == nchunks
    assert np.array_equal(np.concatenate(rep[:nchunks - 1]), np.concatenate(live[:nchunks - 1]))
    assert P2.RX_DONE and not np.array_equal(rep[-1], live[nchunks - 1])      # chunk 2 again, not chunk 3
