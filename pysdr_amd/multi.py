"""Multi-GPU layout of the receiver hot path: one process per GPU.

The path shards by construction: sub-receivers share only the read-only
wideband chunk, streams share nothing.

* primary  -- shard BY STREAM (config C5: 8 streams x 4 RX on 8 GPUs): rank g owns stream g
  with all its sub-receivers; no collective on the data path, rank 0 only gathers the 48 kHz
  audio (a few KB per chunk).
* secondary -- ONE stream, sub-receivers split over ranks (RX r -> rank r mod G).  The
  analogue of the reference's MP_SCHEME 3 fan-out, where the executive puts the same chunk
  on every worker's queue and waits for all of them (``receiver.py:728-739``): here one
  broadcast of the chunk per step, RCCL over xGMI when the buffers live on GPUs
  (``pysdr_comm_bcast``), gloo when they are host arrays (CPU tests).

``torch.distributed`` is control-plane plumbing only (rendezvous, barrier, gathering audio);
the DSP never touches torch."""
from __future__ import annotations

import ctypes as C

import numpy as np


def partition_streams(nstreams, world):
    """Stream indices owned by each rank (contiguous blocks, sizes differ by at most 1)."""
    base, rem = divmod(nstreams, world)
    out, pos = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append(list(range(pos, pos + n)))
        pos += n
    return out


def partition_rx(nrx, world):
    """Sub-receiver indices owned by each rank when one stream is split (r -> r mod G)."""
    return [[r for r in range(nrx) if r % world == g] for g in range(world)]


def max_over_ranks(dt, dist=None):
    """The benchmark's clock: the slowest rank's time (all ranks get it)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(dt)
    import torch
    t = torch.tensor([float(dt)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_audio(local, dist=None, dst=0):
    """``local`` = {key: ndarray} produced on this rank; returns the merged dict on ``dst``
    (None elsewhere).  Keys are (stream, irx) tuples, so ranks never collide."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object(local, parts, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


def broadcast_chunk_host(x, dist, src=0):
    """Host-array broadcast of the wideband chunk (gloo): every rank passes an array of the
    chunk's shape; non-source contents are overwritten."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.float32))
    dist.broadcast(t, src=src)
    return t.numpy().view(np.complex64)


class RcclBroadcaster:
    """Device-buffer broadcast through the C ABI (``pysdr_comm_*`` = ncclBroadcast on the
    context's stream).  The 128-byte ncclUniqueId travels over the control plane."""

    def __init__(self, ctx, dist=None):
        from . import _lib
        self._lib = _lib
        self.ctx = ctx
        L = _lib.lib()
        rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        uid = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.pysdr_comm_unique_id(uid), "pysdr_comm_unique_id")
        if world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)
            uid = C.create_string_buffer(box[0], 128)
        _lib.check(L.pysdr_comm_init(ctx.h, uid, rank, world), "pysdr_comm_init")

    def bcast(self, dev_ptr, nbytes, root=0):
        self._lib.check(self._lib.lib().pysdr_comm_bcast(self.ctx.h, C.c_void_p(int(dev_ptr)),
                                                         int(nbytes), int(root)), "pysdr_comm_bcast")

    def close(self):
        self._lib.lib().pysdr_comm_destroy(self.ctx.h)


class DeviceRxSplit:
    """One rank of the split-RX layout on GPUs: the wideband chunk (or batch of chunks) lives in a
    device buffer, rank ``root`` uploads it, ``ncclBroadcast`` (RCCL over xGMI, on the context's
    stream) hands it to every other GPU, and this rank's sub-receivers demodulate it where it
    landed -- the executive's ``que[irx].put(('DAT',nchunks,x))`` to every worker followed by the
    wait on every ``rx_ready`` (``receiver.py:728-739``, ``am.py:85-114``), with the copy done by
    the fabric instead of a pickled queue.

    ``ctx`` = the ``_StreamContext`` the rank's ``sig_proc.Receiver`` objects share (a context
    without receivers still takes part in the broadcast)."""

    def __init__(self, ctx, max_samples, dist=None, root=0):
        from . import _lib
        self._lib, self.ctx, self.root = _lib, ctx, root
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.device = int(ctx.cfg.device)
        self.cap = int(max_samples)
        self.d_buf = C.c_void_p()
        _lib.check(_lib.lib().pysdr_dev_alloc(self.device, self.cap * 8, C.byref(self.d_buf)), "pysdr_dev_alloc")
        self.bc = RcclBroadcaster(ctx, dist)

    def step(self, x, nchunks, chunk_len):
        """``x`` (complex64, only read on the root) -> broadcast -> demodulate this rank's RX."""
        n = int(nchunks) * int(chunk_len)
        if n > self.cap:
            raise ValueError(f"DeviceRxSplit: {n} samples > capacity {self.cap}")
        L = self._lib.lib()
        if self.rank == self.root:
            x = np.ascontiguousarray(x, np.complex64)
            # the context's stream is non-blocking, so the (null-stream) upload is not ordered against the
            # kernels of the previous step that still read d_buf: wait for them here rather than rely on
            # the caller having fetched (a root without local receivers, a PSD consumer, ...)
            self._lib.check(L.pysdr_sync(self.ctx.h), "pysdr_sync")
            self._lib.check(L.pysdr_dev_upload(self.device, self.d_buf, C.c_void_p(x.ctypes.data), n * 8),
                            "pysdr_dev_upload")
        self.bc.bcast(self.d_buf.value, n * 8, self.root)
        if self.ctx.receivers:
            self.ctx.process_batch(self.d_buf.value, nchunks, chunk_len, on_device=True)

    def fetch(self, irx_local, nchunks):
        return self.ctx.fetch(irx_local, nchunks)

    def close(self):
        self._lib.check(self._lib.lib().pysdr_sync(self.ctx.h), "pysdr_sync")
        self.bc.close()
        self._lib.lib().pysdr_dev_free(self.device, self.d_buf)


def run_sharded(streams, make_rx, chunk_len, nchunks, dist=None, mode="stream", nrx=None, device_split=None):
    """Process ``streams`` (list of complex64 arrays, all ranks hold the list; only the owner
    touches its entries) for ``nchunks`` chunks and return {(stream, irx): audio} on rank 0.

    ``make_rx(stream_index, rx_indices)`` -> list of receiver objects with ``demod_data``
    (``pysdr_amd.sig_proc.Receiver`` on a GPU box, the oracle in CPU tests).
    mode "stream": shard by stream.  mode "rx": ONE stream (streams[0]) of ``nrx`` sub-receivers
    split across ranks, the chunk broadcast from rank 0 each step: over RCCL into device buffers
    when ``device_split(rank_rx_objects) -> DeviceRxSplit`` is given (GPU boxes), as a host array
    over gloo otherwise (CPU tests; the receivers then take host arrays)."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    local = {}
    if mode == "stream":
        for si in partition_streams(len(streams), world)[rank]:
            rxs = make_rx(si, None)
            acc = [[] for _ in rxs]
            for k in range(nchunks):
                x = streams[si][k * chunk_len:(k + 1) * chunk_len]
                for i, rx in enumerate(rxs):
                    acc[i].append(np.array(rx.demod_data(x)))
            for i in range(len(rxs)):
                local[(si, i)] = np.concatenate(acc[i])
    elif mode == "rx":
        if nrx is None:
            raise ValueError("run_sharded(mode='rx') needs nrx, the number of sub-receivers of the stream")
        idx = partition_rx(int(nrx), world)[rank]
        rxs = make_rx(0, idx)
        acc = [[] for _ in idx]
        split = device_split(rxs) if device_split is not None else None
        for k in range(nchunks):
            x = streams[0][k * chunk_len:(k + 1) * chunk_len] if rank == 0 else None
            if split is not None:
                split.step(x, 1, chunk_len)
                for j in range(len(idx)):
                    acc[j].append(np.array(split.fetch(j, 1)[0]))
                continue
            x = np.ascontiguousarray(x, np.complex64) if rank == 0 else np.zeros(chunk_len, np.complex64)
            if world > 1:
                x = broadcast_chunk_host(x, dist, src=0)
            for j, rx in enumerate(rxs):
                acc[j].append(np.array(rx.demod_data(x)))
        if split is not None:
            split.close()
        for j, i in enumerate(idx):
            local[(0, i)] = np.concatenate(acc[j]) if acc[j] else np.zeros(0, np.float32)
    else:
        raise ValueError(mode)
    return gather_audio(local, dist, dst=0)
