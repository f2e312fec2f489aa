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


def run_sharded(streams, make_rx, chunk_len, nchunks, dist=None, mode="stream"):
    """Process ``streams`` (list of complex64 arrays, all ranks hold the list; only the owner
    touches its entries) for ``nchunks`` chunks and return {(stream, irx): audio} on rank 0.

    ``make_rx(stream_index, rx_indices)`` -> list of receiver objects with ``demod_data``
    (``pysdr_amd.sig_proc.Receiver`` on a GPU box, the oracle in CPU tests).
    mode "stream": shard by stream.  mode "rx": ONE stream (streams[0]), sub-receivers split
    across ranks, the chunk broadcast from rank 0 each step."""
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
        mine = None
        for k in range(nchunks):
            if rank == 0:
                x = np.ascontiguousarray(streams[0][k * chunk_len:(k + 1) * chunk_len], np.complex64)
            else:
                x = np.zeros(chunk_len, np.complex64)
            if world > 1:
                x = broadcast_chunk_host(x, dist, src=0)
            if mine is None:
                nrx = len(make_rx.rx_modes)
                idx = partition_rx(nrx, world)[rank]
                mine = (idx, make_rx(0, idx), [[] for _ in idx])
            idx, rxs, acc = mine
            for j, rx in enumerate(rxs):
                acc[j].append(np.array(rx.demod_data(x)))
        if mine is not None:
            idx, rxs, acc = mine
            for j, i in enumerate(idx):
                local[(0, i)] = np.concatenate(acc[j]) if acc[j] else np.zeros(0, np.float32)
    else:
        raise ValueError(mode)
    return gather_audio(local, dist, dst=0)
