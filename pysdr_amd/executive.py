"""Headless receiver executive: the caller side of the hot path, with the same names, call
order and per-chunk post-processing as the reference's ``receiver.py`` (``SDR_EXECUTIVE.Run``
:684, ``read_chunk`` :538, ``mode_freq_change`` :633, ``demodulate_data`` :231,
``audio_out`` :153) and the ``am.py`` harness (:54-75), minus everything that needs a
display, a sound card or a radio.  It drives ``dsp.Receiver`` objects -- by default
``pysdr_amd.sig_proc`` (the GPU) -- through exactly the calls the reference makes, so the
parity tests read like the reference's main loop.

MP_SCHEME 1 only (one RX thread, all sub-receivers served per chunk, ``receiver.py:723-725``);
the GPU context shares the chunk between sub-receivers instead of MP_SCHEME 3's process
fan-out (``receiver.py:726-739``)."""
from __future__ import annotations

import numpy as np

from .stream import SOAPY_SDR_CF32, SOAPY_SDR_RX
from .tables import AF_BWs, VIDEO_BWs


class NullPlayer:
    """Stands in for ``audio_io.AudioIO`` (``receiver.py:850``): owns the ring buffer the
    audio would be pulled from; never opens a sound device."""

    def __init__(self, rb, fs, tag=''):
        self.rb = rb
        self.fs = fs
        self.tag = tag
        self.active = False
        self.Start_Time = 0.0

    def start_playback(self, delay, flag):
        self.active = True
        return True

    def pause(self):
        self.active = False

    def resume(self):
        self.active = True

    def stop(self):
        self.active = False


def demodulate_data(P, x, irx):
    """One sub-receiver, one chunk (``receiver.py:231-297``)."""
    rx = P.rx[irx]
    am = rx.demod_data(x)
    return post_demod(P, x, irx, am)


def post_demod(P, x, irx, am):
    """Everything ``demodulate_data`` does behind ``rx.demod_data(x)`` (``receiver.py:238-297``)."""
    rx = P.rx[irx]
    if getattr(P, 'ENABLE_AUTO_MUTE', False):           # receiver.py:238-245
        P.AUTO_MUTED = bool(rx.auto_mute(x))

    # receiver.py:250-252: DC removal, decided by the GLOBAL mode P.MODE.  As in the reference it only
    # reaches the consumers below (AF PSD tap, SAVE_DEMOD): `am` is re-bound to a new array, `rx.am` --
    # what audio_out plays (receiver.py:195) -- keeps what demod_data left there
    # (tests/golden/host_loop_ref.npz: the reference's text executed)
    if P.MODE == 'AM' or P.MODE == 'USB':
        am = am - np.mean(am)

    if P.SHOW_AF_PSD and irx == P.PLOT_RX:              # receiver.py:257-274
        P.rb_af.push(rx.iq if P.PANADAPTOR else am)
    if P.SHOW_BASEBAND_PSD and irx == P.PLOT_RX:        # receiver.py:276-284
        P.rb_baseband.push(rx.iq)
    if P.SAVE_BASEBAND and irx == 0:                    # receiver.py:293-297
        P.baseband_iq_io.save_data(rx.iq)
    if P.SAVE_DEMOD and irx == 0:
        P.demod_io.save_data(am)
    return am


def audio_out(P):
    """Route demodulated audio to the players' ring buffers (``receiver.py:153-225``)."""
    if P.AUDIO_SCHEME == 2:                             # two mono RX per stereo player
        n2 = int((P.NUM_RX + 1) / 2)
        for iplay in range(n2):
            g1 = 0. if P.MUTED[iplay] else pow(10., P.AF_GAIN) - 1
            am1 = P.rx[iplay].am.real
            if iplay + n2 < P.NUM_RX:
                g2 = 0. if P.MUTED[iplay + n2] else pow(10., P.AF_GAIN) - 1
                am2 = P.rx[iplay + n2].am.real
            else:
                g2, am2 = 0., 0
            if P.audio_playback:
                player = P.players[iplay]
                player.rb.push(am1 * g1 + 1j * am2 * g2)
                if not player.active:
                    player.start_playback(P.DELAY, False)
        return
    for irx in range(P.NUM_RX):                         # default: one player per RX
        am = P.rx[irx].am
        player = P.players[irx]
        gain = 0. if (P.MUTED[irx] or P.AUTO_MUTED) else pow(10., P.AF_GAIN) - 1
        if P.audio_playback:
            player.rb.push(am * gain)
            if not player.active:
                player.start_playback(P.DELAY, False)


class SDR_EXECUTIVE:
    def __init__(self, P, GUI=False, dsp=None):
        if dsp is None:
            from . import sig_proc as dsp
        self.dsp = dsp
        self.P = P
        P.SDR_EXEC = self
        P.RX_DONE = False
        P.nchunks = 0
        self.create_Receivers()
        self.create_Audio_Players()
        # buffers: grab enough RF samples to produce one block of OUT_CHUNK_SIZE audio
        self.xold = np.zeros(0, np.complex64)
        self.xx = np.zeros(P.IN_CHUNK_SIZE, np.complex64)
        self.x = np.zeros(P.IN_CHUNK_SIZE, np.complex64)
        self.raw = None
        self.praw = 0

    # -- receiver.py:826-835
    def create_Receivers(self):
        P = self.P
        for irx in range(P.NUM_RX):
            if P.SOURCE[irx] >= 0:
                frq = P.FC[irx] - P.FC[P.SOURCE[irx]]
            else:
                frq = P.FOFFSET + P.FC[irx] - P.FC[0]
            P.rx[irx] = self.dsp.Receiver(P, frq, irx, str(irx + 1), VIDEO_BWs, AF_BWs)

    # -- receiver.py:838-851 without the sound card
    def create_Audio_Players(self):
        P = self.P
        P.players = []
        for irx in range(P.NUM_PLAYERS):
            rb = self.dsp.ring_buffer2('Audio' + str(irx + 1), P.RB_SIZE)
            P.players.append(NullPlayer(rb, P.FS_OUT, 'RX ' + str(irx)))

    # -- receiver.py:504-534
    def Startup(self):
        P = self.P
        if P.REPLAY_MODE:
            # the reference loads the whole recording (receiver.py:526); a fileio reader is
            # streamed through its memory map instead, chunk by chunk
            self.raw = None if hasattr(P.sdr, 'read_chunk') else P.sdr.read_data()
            self.praw = 0
        else:
            P.rxStream = P.sdr.setupStream(SOAPY_SDR_RX, SOAPY_SDR_CF32)
            P.sdr.activateStream(P.rxStream)

    # -- receiver.py:538-631
    def read_chunk(self):
        P = self.P
        n = P.IN_CHUNK_SIZE
        if P.REPLAY_MODE:
            total = len(self.raw) if self.raw is not None else P.sdr.nsamples
            if self.praw + n < total:                   # strict, as receiver.py:543
                x1 = self.raw[self.praw:self.praw + n] if self.raw is not None else P.sdr.read_chunk(n)
                self.praw += n
                lo = getattr(P, 'lo', None)
                self.x = lo.quad_mixer(x1) if (lo is not None and lo.fo != 0) else x1
            else:
                P.RX_DONE = True
            return
        # readStream does not block: keep reading until the chunk is full, carrying the
        # surplus of the last read over to the next chunk
        nn = len(self.xold)
        if nn > 0:
            self.x[0:nn] = self.xold
            self.xold = self.xold[:0]
        n1 = nn
        idle = 0
        while n1 < n and not (P.Stopper and P.Stopper.is_set()):
            try:
                sr = P.sdr.readStream(P.rxStream, [self.xx], n)
                got = sr.ret
            except Exception:                           # receiver.py:603-605: trap, treat as 0
                got = 0
            if got > 0:
                n2 = min(n1 + got, n)
                m = n1 + got - n2
                self.x[n1:n2] = self.xx[0:got - m]
                self.xold = self.xx[got - m:got].copy()
                n1 = n2
                idle = 0
            else:
                idle += 1
                exhausted = getattr(P.sdr, 'exhausted', None)
                if (exhausted is not None and exhausted()) or idle > 1000:
                    P.RX_DONE = True                    # synthetic source ran dry
                    return

    # -- receiver.py:633-650
    def mode_freq_change(self):
        P = self.P
        if P.MODE_CHANGE:
            if P.NEW_MODE == 'FM':
                P.NEW_MODE = 'NFM'
            if P.MODE != P.NEW_MODE:
                P.MODE = P.NEW_MODE
                if getattr(P, 'MP_SCHEME', 1) in (1, 2):    # receiver.py:646-648: in scheme 3 the workers own the receivers
                    P.rx[0].agc.reset()
                    P.rx[0].demod.am_pll.reset()
            P.MODE_CHANGE = False

    # -- receiver.py:684-773
    def Run(self, on_chunk=None):
        P = self.P
        dt = float(P.IN_CHUNK_SIZE) / P.SRATE
        t = 0.0
        self.Startup()
        while not P.RX_DONE:
            t += dt
            P.nchunks += 1
            if P.Stopper and P.Stopper.is_set():
                P.RX_DONE = True
                break
            self.read_chunk()
            if P.RX_DONE and not P.REPLAY_MODE:
                break                                  # the synthetic live source ran dry: self.x is only partly new
            # (a replay that runs out sets RX_DONE in read_chunk and leaves self.x as it was; like the reference
            #  -- receiver.py:543-557,715-740 -- the loop body still runs once more on that stale chunk)
            self.mode_freq_change()
            for irx in range(P.NUM_RX):
                demodulate_data(P, self.x, irx)
            audio_out(P)
            if P.SHOW_RF_PSD:
                P.rb_rf.push(self.x)
            if P.SAVE_IQ:
                P.raw_iq_io.save_data(self.x)
            if on_chunk is not None:
                on_chunk(self)
            P.RX_DONE = P.RX_DONE or t >= P.DURATION
        self.quit_rx()

    # NOTE (AM-Synch / WFM2): a slot of several chunks runs the serial PLL in segments with joins accepted
    # within a tolerance (include/pysdr_hip.h, pysdr_process_batch): batched == chunk by chunk within the
    # 1e-5 parity bar there, bit for bit in every other mode.
    def Run_pipelined(self, on_chunk=None, nslots=3, batch_chunks=1):
        """``Run`` with the ingest ring (N4): ``self.x`` is (a chunk of) a pinned ring slot, the slot
        is submitted asynchronously and its audio is post-processed one slot later, while the next
        chunks are being read.  ``batch_chunks`` > 1 puts that many chunks into one slot = one DMA
        and one launch sequence (throughput, at the price of that many chunks of latency).  Same
        chunks, same order, same results as ``Run`` -- live and in REPLAY_MODE, the stale last pass of a
        replay included (``tests/test_gpu_executive.py``)."""
        from .ingest import IngestRing
        P = self.P
        B = max(1, int(batch_chunks))
        if B > 1 and getattr(P._pysdr_stream, 'max_chunks', 1) < B:
            raise ValueError(f"Run_pipelined(batch_chunks={B}) needs P.MAX_BATCH_CHUNKS >= {B}")
        L = P.IN_CHUNK_SIZE
        ring = IngestRing(P._pysdr_stream, nslots, B)
        dt = float(L) / P.SRATE
        t = 0.0
        self.Startup()
        slot, pending, filled = 0, None, 0
        last = None                                         # the chunk before this one (a view into a ring slot)

        def finish(ps):
            nfill = ps[1]
            cn, pk = ring.chunks(ps[0])          # (before collect: collecting releases the slot)
            res = ring.collect(ps[0])
            xs_all = ring.buffer(ps[0])
            pos = 0
            for k in range(nfill):
                xs = xs_all[k * L:(k + 1) * L]
                n = int(cn[k])
                for irx in range(P.NUM_RX):
                    rx = P.rx[irx]
                    am, iq, _ = res[irx]
                    rx.am, rx.iq, rx.peak_in = am[pos:pos + n], iq[pos:pos + n], float(pk[k])
                    post_demod(P, xs, irx, rx.am)
                pos += n
                audio_out(P)
                if P.SHOW_RF_PSD:
                    P.rb_rf.push(xs.copy())
                if P.SAVE_IQ:
                    P.raw_iq_io.save_data(xs)
                if on_chunk is not None:
                    self.x = xs
                    on_chunk(self)

        def flush():
            nonlocal slot, pending, filled
            if filled == 0:
                return
            ring.submit(slot, filled * L)
            if pending is not None:
                finish(pending)
            pending = (slot, filled)
            slot = (slot + 1) % nslots
            filled = 0

        try:
            while not P.RX_DONE:
                t += dt
                P.nchunks += 1
                if P.Stopper and P.Stopper.is_set():
                    P.RX_DONE = True
                    break
                dst = ring.buffer(slot)[filled * L:(filled + 1) * L]
                self.x = dst
                self.read_chunk()
                if P.RX_DONE:
                    if not P.REPLAY_MODE:
                        break                               # the live source ran dry: the chunk is only partly new
                    # a replay that runs out leaves self.x as it was and the reference's loop body still runs once
                    # more on that stale chunk (receiver.py:543-557,715-740; Run above does the same): here self.x
                    # already points at the next ring slot, so the previous chunk is copied into it
                    dst[:] = last if last is not None else 0
                elif self.x is not dst:                     # replay hands back its own array
                    dst[:] = self.x
                last = dst
                if self.mode_freq_change_pending():
                    flush()                                 # a retune / mode change applies from THIS chunk on
                    dst2 = ring.buffer(slot)[0:L]
                    if dst2 is not dst:
                        dst2[:] = dst
                self.mode_freq_change()
                filled += 1
                if filled == B:
                    flush()
                P.RX_DONE = P.RX_DONE or t >= P.DURATION
            flush()
            if pending is not None:
                finish(pending)
        finally:
            ring.close()
        self.quit_rx()

    def mode_freq_change_pending(self):
        P = self.P
        return bool(getattr(P, 'MODE_CHANGE', False) or getattr(P, 'FREQ_CHANGE', False))

    # -- receiver.py:461-500
    def quit_rx(self):
        P = self.P
        for pl in P.players:
            if pl.active:
                pl.stop()
        if not P.REPLAY_MODE and P.sdr is not None:
            P.sdr.deactivateStream(P.rxStream)
            P.sdr.closeStream(P.rxStream)


def replay_batched(P, batch_chunks=64, on_batch=None, dsp=None):
    """Replay a recording (``P.sdr`` = ``fileio.sdr_fileio`` reader, see ``fileio.open_replay``)
    through the batched device path: ``batch_chunks`` chunks per launch sequence instead of one
    Python round trip per chunk, the results identical to ``SDR_EXECUTIVE.Run`` in REPLAY_MODE
    chunk for chunk (a chunk is still one AGC block; ``receiver.py:541-557`` for the slicing and
    the tuning-offset mixer, ``:250-252`` for the DC removal of the saved demod, ``:293-297`` for the save taps).

    ``on_batch(first_chunk, am, iq, chunk_nout)`` gets, per batch, lists over the sub-receivers of
    the audio / baseband IQ of the whole batch and the per-chunk output counts.  Returns the
    number of chunks processed.  GPU only (``pysdr_amd.sig_proc``)."""
    if dsp is None:
        from . import sig_proc as dsp
    P.MAX_BATCH_CHUNKS = int(batch_chunks)
    ex = SDR_EXECUTIVE(P, dsp=dsp)
    ctx = P._pysdr_stream
    L = P.IN_CHUNK_SIZE
    file_total = (P.sdr.nsamples - 1) // L             # strict '<' of receiver.py:543
    dur_total = int(np.ceil(P.DURATION * P.SRATE / L)) if getattr(P, 'DURATION', None) else None   # t >= DURATION ends Run
    total = file_total if dur_total is None else min(file_total, dur_total)
    ran_out = dur_total is None or dur_total > file_total   # Run would start one more pass and find the file exhausted
    lo = getattr(P, 'lo', None)
    done = 0
    x_last = None
    P.sdr.rewind()

    def run_batch(x, nb):
        ctx.process_batch(x, nb, L)
        ams, iqs, cns = [], [], None
        for irx in range(P.NUM_RX):
            rx = P.rx[irx]
            am, iq, cn, pk = ctx.fetch(rx.irx, nb)
            ams.append(am)                             # the audio: rx.am as demod_data leaves it (receiver.py:195)
            iqs.append(iq)
            cns = cn
            if irx == 0:
                if P.SAVE_BASEBAND:
                    P.baseband_iq_io.save_data(iq)
                if P.SAVE_DEMOD:
                    if P.MODE == 'AM' or P.MODE == 'USB':   # receiver.py:250-252: per chunk, the saved / PSD copy only
                        am = am.copy()
                        pos = 0
                        for c in cn:
                            am[pos:pos + c] -= np.mean(am[pos:pos + c])
                            pos += c
                    P.demod_io.save_data(am)
        if P.SAVE_IQ:
            P.raw_iq_io.save_data(x)
        if on_batch is not None:
            on_batch(done, ams, iqs, cns)

    while done < total:
        nb = min(batch_chunks, total - done)
        x = P.sdr.read_chunk(nb * L)
        if lo is not None and lo.fo != 0:
            x = lo.quad_mixer(x)
        run_batch(x, nb)
        done += nb
        P.nchunks += nb
        x_last = np.array(x[(nb - 1) * L:nb * L])
    if done > 0 and ran_out:
        # the recording ran out (not P.DURATION): SDR_EXECUTIVE.Run of the reference runs its loop body once more
        # on the stale last chunk in the pass that discovers it (receiver.py:543-557,715-740: demodulated, played,
        # tapped and saved again); so does Run above, so does this path
        run_batch(x_last, 1)
        done += 1
        P.nchunks += 1
    ex.quit_rx()
    return done
