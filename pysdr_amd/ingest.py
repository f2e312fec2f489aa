"""Ingest ring: the step in front of ``rx.demod_data(x)`` for a live
device.  The reference reads the radio into ``self.xx`` and assembles one chunk in ``self.x``
(short reads, the ``xold`` carry, ``receiver.py:579-631``; ``soapy.py:33-48`` for the
stream calls); here ``self.x`` IS a pinned host buffer of the ring, so the assembled chunk goes
to the GPU with one asynchronous DMA while the host already assembles the next one, and the
results come back into pinned buffers the same way (``pysdr_ingest_*`` in the C ABI)."""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import check


class IngestRing:
    def __init__(self, ctx, nslots=3, chunks_per_slot=1):
        """``ctx`` = the stream context shared by the sub-receivers (``P._pysdr_stream``).
        ``chunks_per_slot`` > 1: a slot holds that many consecutive chunks and goes through the
        device as ONE batch (same arithmetic; the per-chunk fixed costs are paid once per slot)."""
        self.ctx = ctx
        self.nslots = int(nslots)
        self.chunks_per_slot = int(chunks_per_slot)
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.pysdr_ingest_create_batched(ctx.h, self.nslots, self.chunks_per_slot, C.byref(h)),
              "pysdr_ingest_create_batched")
        self.h = h
        if not hasattr(ctx, '_rings'):
            ctx._rings = []
        ctx._rings.append(self)
        self._nrx = {}               # slot -> sub-receivers at its submit
        self._bufs = []
        for s in range(self.nslots):
            p = C.POINTER(C.c_float)()
            cap = C.c_size_t(0)
            check(self.L.pysdr_ingest_buffer(self.h, s, C.byref(p), C.byref(cap)), "pysdr_ingest_buffer")
            a = np.ctypeslib.as_array(p, shape=(2 * cap.value,)).view(np.complex64)
            self._bufs.append(a)

    def buffer(self, slot):
        """complex64 view of the slot's pinned chunk buffer (IN_CHUNK_SIZE samples)."""
        return self._bufs[slot]

    def submit(self, slot, n=None):
        n = len(self._bufs[slot]) if n is None else int(n)
        for rx in self.ctx.receivers:
            rx._sync_controls()
        check(self.L.pysdr_ingest_submit(self.h, slot, n), "pysdr_ingest_submit")
        self._nrx[slot] = len(self.ctx.receivers)
        self.ctx.seq += max(1, n // max(1, int(self.ctx.cfg.in_chunk)))

    def chunks(self, slot):
        """-> (per-chunk output counts, per-chunk raw peaks max|x|^2) of a submitted slot (waits for it)."""
        n = C.c_int(0)
        cn = np.zeros(self.chunks_per_slot, np.int32)
        pk = np.zeros(self.chunks_per_slot, np.float32)
        check(self.L.pysdr_ingest_chunks(self.h, slot, self.chunks_per_slot, C.byref(n),
                                         cn.ctypes.data_as(C.POINTER(C.c_int)), _lib.as_pf(pk)), "pysdr_ingest_chunks")
        return cn[:n.value].copy(), pk[:n.value].copy()

    def collect(self, slot):
        """-> [(am, iq, peak_in)] per sub-receiver (copies: the slot may be reused at once)."""
        # the sub-receivers the slot was SUBMITTED with (a receiver added since then has no result in it)
        nrx = self._nrx.get(slot, len(self.ctx.receivers))
        outs = (_lib.Out * max(nrx, 1))()
        check(self.L.pysdr_ingest_collect(self.h, slot, outs), "pysdr_ingest_collect")
        res = []
        for r in range(nrx):
            k = outs[r].n_out
            if outs[r].am_is_complex:
                am = np.ctypeslib.as_array(outs[r].am, shape=(2 * max(k, 1),))[:2 * k].copy().view(np.complex64)
            else:
                am = np.ctypeslib.as_array(outs[r].am, shape=(max(k, 1),))[:k].copy()
            iq = np.ctypeslib.as_array(outs[r].iq, shape=(2 * max(k, 1),))[:2 * k].copy().view(np.complex64)
            res.append((am, iq, float(outs[r].peak_in)))
        return res

    def close(self):
        if self.h:
            self.L.pysdr_ingest_destroy(self.h)
            self.h = None
            if self in getattr(self.ctx, '_rings', []):
                self.ctx._rings.remove(self)

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass
