"""The reference's main loop (SDR_EXECUTIVE.Run, receiver.py:684-773; am.py:54-75) driven
end to end on the GPU through the sig_proc facade, against the same loop on the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import sdr_oracle as so
from pysdr_amd import executive, stream
from tests import oracle_dsp
from tests.test_executive_host import make_P

pytestmark = pytest.mark.gpu


def run_exec(cfg, nchunks, dsp, seed, modes=True, **kw):
    P = make_P(cfg, nchunks, **kw)
    L = P.IN_CHUNK_SIZE
    P.sdr = stream.SynthSDR(cfg, seed=seed, nsamp=(nchunks + 1) * L)
    ex = executive.SDR_EXECUTIVE(P, dsp=dsp)
    if modes:
        for i, r in enumerate(cfg['rx']):
            P.rx[i].mode, P.rx[i].af_bw, P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
    P.trace = []          # per chunk: max |x|, max |rx.iq|, max |rx.am| of RX 0 (printed when an assert fails)
    ex.Run(on_chunk=lambda e: P.trace.append((float(np.abs(e.x).max()), float(np.abs(P.rx[0].iq).max()),
                                              float(np.abs(P.rx[0].am).max()), bool(P.MUTED[0]), bool(P.AUTO_MUTED))))
    return P, [pl.rb.pull(pl.rb.nsamps) for pl in P.players]


def test_am_py_path_c1():
    """config #1: 2.048 MS/s replayed IQ, 1 RX, AM to 48 kHz (am.bat: am.py -fake -fc 15e3)"""
    cfg = so.CONFIGS['C1']
    Pg, ag = run_exec(cfg, 8, None, 31)
    Po, ao = run_exec(cfg, 8, oracle_dsp, 31)
    assert len(ag[0]) == len(ao[0]) and len(ag[0]) in range(8 * 1023, 8 * 1024 + 1)
    assert np.max(np.abs(ag[0] - ao[0])) <= 1e-5 * np.max(np.abs(ao[0])), (Pg.trace, Po.trace)


def test_four_rx_loop_with_short_reads_and_stereo_routing():
    cfg = so.CONFIGS['C3']
    Pg, ag = run_exec(cfg, 4, None, 32, audio=2)
    Po, ao = run_exec(cfg, 4, oracle_dsp, 32, audio=2)
    for a, b in zip(ag, ao):
        assert a.shape == b.shape and np.iscomplexobj(a)
        assert np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(b))          # every sample, start-up included


def test_recorded_iq_replayed_in_batches_matches_the_chunked_oracle(tmp_path):
    """N2: a raw_iq recording with a tuning offset (P.lo, receiver.py:552-553) replayed through
    the batched device path (7 chunks in batches of 3) = the reference's chunk-by-chunk replay
    on the oracle, sub-receiver by sub-receiver, including the per-chunk DC removal."""
    from pysdr_amd import fileio as file_io
    cfg = so.CONFIGS['C3']
    nchunks = 7
    P0 = make_P(cfg, nchunks)
    L = P0.IN_CHUNK_SIZE
    x = so.synth_iq(cfg, (nchunks + 1) * L, 41)
    w = file_io.sdr_fileio('raw_iq', 'w', P0, 2, 'RAW_IQ', out_dir=str(tmp_path))
    w.save_data(x)
    w.close()
    foff = 12500.0

    def setup(dsp_mod):
        P = make_P(cfg, nchunks)
        P.REPLAY = w.fname
        file_io.open_replay(P) if dsp_mod is None else file_io.open_replay(P, dsp=dsp_mod)
        P.DURATION = nchunks * L / P.SRATE - 1e-9
        return P

    # GPU, batched
    Pg = setup(None)
    Pg.lo.change_freq(foff)
    got = [[] for _ in cfg['rx']]
    counts = []

    def on_batch(first, ams, iqs, cn):
        counts.append(list(cn))
        for i, a in enumerate(ams):
            got[i].append(np.array(a))

    # per-RX modes must be in place before the first batch: build the receivers, then replay
    orig = executive.SDR_EXECUTIVE.create_Receivers

    def create_with_modes(self):
        orig(self)
        for i, r in enumerate(cfg['rx']):
            self.P.rx[i].mode, self.P.rx[i].af_bw, self.P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)

    executive.SDR_EXECUTIVE.create_Receivers = create_with_modes
    try:
        n = executive.replay_batched(Pg, batch_chunks=3, on_batch=on_batch)
    finally:
        executive.SDR_EXECUTIVE.create_Receivers = orig
    assert n == nchunks and [len(c) for c in counts] == [3, 3, 1]

    # the reference's own replay loop (chunk by chunk) on the oracle, same 32-bit NCO for P.lo
    class _OracleGen:
        @staticmethod
        def signal_generator(f, n, fs, flag):
            return so.NCO(f, fs, np.float32)

    Po = setup(_OracleGen)
    Po.lo.change_freq(foff)
    want = [[] for _ in cfg['rx']]
    ex = executive.SDR_EXECUTIVE(Po, dsp=oracle_dsp)
    for i, r in enumerate(cfg['rx']):
        Po.rx[i].mode, Po.rx[i].af_bw, Po.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
    ex.Run(on_chunk=lambda e: [want[i].append(np.array(Po.rx[i].am)) for i in range(len(cfg['rx']))])
    assert len(want[0]) == nchunks
    rxs = cfg['rx']
    for i in range(len(rxs)):
        a, b = np.concatenate(got[i]), np.concatenate(want[i])
        assert a.shape == b.shape
        assert np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(b)), i


def test_ingest_ring_pipelined_run_equals_the_synchronous_run():
    """N4: the same loop with pinned ring slots, asynchronous H2D / kernels / D2H and the
    post-processing one chunk late = the synchronous loop, bit for bit (short reads, xold carry,
    4 RX, stereo routing, auto-mute input peak)."""
    cfg = so.CONFIGS['C3']
    nchunks = 6

    def run(pipelined, dsp=None, batch=1):
        P = make_P(cfg, nchunks, audio=2, max_batch_chunks=max(batch, 1))
        L = P.IN_CHUNK_SIZE
        P.sdr = stream.SynthSDR(cfg, seed=51, nsamp=(nchunks + 1) * L)
        P.ENABLE_AUTO_MUTE = True
        ex = executive.SDR_EXECUTIVE(P, dsp=dsp)
        for i, r in enumerate(cfg['rx']):
            P.rx[i].mode, P.rx[i].af_bw, P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
        seen, peaks = [], []

        def on_chunk(e):
            seen.append(e.x.copy())
            peaks.append(P.rx[0].peak_in)

        if pipelined:
            ex.Run_pipelined(on_chunk=on_chunk, batch_chunks=batch)
        else:
            ex.Run(on_chunk=on_chunk)
        return P, [pl.rb.pull(pl.rb.nsamps) for pl in P.players], seen, peaks

    Pa, aa, sa, pa = run(False)
    Pb, ab, sb, pb = run(True)
    assert len(sa) == len(sb) == nchunks
    assert all(np.array_equal(u, v) for u, v in zip(sa, sb))
    assert pa == pb and pa[0] > 0
    for u, v in zip(aa, ab):
        assert u.shape == v.shape and np.array_equal(u, v)
    assert Pb.sdr.ncall > nchunks
    # slots of 4 chunks (6 chunks = one full slot + a partial one): one DMA + one launch sequence per
    # slot, the audio cut back into chunks on the host -- still the synchronous run, bit for bit
    Pc, ac, sc, pc = run(True, batch=4)
    assert len(sc) == nchunks and all(np.array_equal(u, v) for u, v in zip(sa, sc)) and pa == pc
    for u, v in zip(aa, ac):
        assert u.shape == v.shape and np.array_equal(u, v)
    # ... and the pipelined run against the ORACLE's loop directly (not only against the other GPU
    # path): same chunks, same routing, same auto-mute input
    Po, ao, so_, po = run(False, dsp=oracle_dsp)
    assert all(np.array_equal(u, v) for u, v in zip(so_, sb))
    assert np.allclose(po, pb, rtol=1e-6, atol=0)
    for u, v in zip(ao, ab):
        assert u.shape == v.shape
        assert np.max(np.abs(u - v)) <= 1e-5 * np.max(np.abs(u))


def test_replay_run_and_pipelined_run_are_the_same_including_the_stale_last_pass(tmp_path):
    """REPLAY_MODE: when the recording runs out the reference's loop body runs once more on the chunk it already had
    (receiver.py:543-557,715-740).  ``Run`` reproduces that; ``Run_pipelined`` must too (it used to stop one
    chunk early) -- same chunks, same audio, same saved IQ, with one and with several chunks per ring slot."""
    from pysdr_amd import fileio as file_io
    cfg = so.CONFIGS['C3']
    nchunks = 5
    P0 = make_P(cfg, nchunks)
    L = P0.IN_CHUNK_SIZE
    x = so.synth_iq(cfg, nchunks * L + L // 3, 43)          # the file ends inside chunk 5: the replay runs out
    w = file_io.sdr_fileio('raw_iq', 'w', P0, 2, 'RAW_IQ', out_dir=str(tmp_path))
    w.save_data(x)
    w.close()

    def run(pipelined, batch=1):
        P = make_P(cfg, nchunks + 4, audio=2, max_batch_chunks=max(batch, 1))
        P.REPLAY = w.fname
        file_io.open_replay(P)
        P.DURATION = 1e9                                    # the recording, not the clock, ends the run
        ex = executive.SDR_EXECUTIVE(P)
        for i, r in enumerate(cfg['rx']):
            P.rx[i].mode, P.rx[i].af_bw, P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
        seen = []
        if pipelined:
            ex.Run_pipelined(on_chunk=lambda e: seen.append(e.x.copy()), batch_chunks=batch)
        else:
            ex.Run(on_chunk=lambda e: seen.append(e.x.copy()))
        return seen, [pl.rb.pull(pl.rb.nsamps) for pl in P.players]

    sa, aa = run(False)
    assert len(sa) >= 2 and np.array_equal(sa[-1], sa[-2])   # the stale pass: the last chunk twice
    for batch in (1, 3):
        sb, ab = run(True, batch)
        assert len(sb) == len(sa) and all(np.array_equal(u, v) for u, v in zip(sa, sb)), batch
        for u, v in zip(aa, ab):
            assert u.shape == v.shape and np.array_equal(u, v), batch


def test_am_synch_batched_slots_equal_the_chunked_run_within_the_parity_bar():
    """with a serial loop in the chain (AM-Synch carrier PLL) a slot of several chunks runs it
    in segments whose joins are accepted within a tolerance, so Run_pipelined(batch_chunks > 1) equals
    the chunk-by-chunk Run within the 1e-5 bar (not bitwise, unlike the other modes) -- and the oracle's
    serial loop within the same bar."""
    cfg = dict(so.CONFIGS['C1'], ntaps_dec=255, rx=[dict(frq=100e3 - 5.0, mode='AM-Synch', video_bw=10e3, af_bw=5e3)])
    nchunks = 32

    def run(pipelined, dsp=None, batch=1):
        P = make_P(cfg, nchunks, max_batch_chunks=max(batch, 1))
        L = P.IN_CHUNK_SIZE
        P.sdr = stream.SynthSDR(cfg, seed=61, nsamp=(nchunks + 1) * L, read_pattern=(1.0,))
        ex = executive.SDR_EXECUTIVE(P, dsp=dsp)
        P.rx[0].mode, P.rx[0].af_bw = 'AM-Synch', 5e3
        if pipelined:
            ex.Run_pipelined(batch_chunks=batch)
        else:
            ex.Run()
        return P.players[0].rb.pull(P.players[0].rb.nsamps)

    a = run(False)
    b = run(True, batch=32)
    o = run(False, dsp=oracle_dsp)
    assert a.shape == b.shape == o.shape and len(a) >= nchunks * 1023
    assert np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(a))
    assert np.max(np.abs(b - o)) <= 1e-5 * np.max(np.abs(o))
