"""PCIe-inclusive rate of the host-fed path (never the bench `value`): C3, 4 RX, chunks handed over
as host arrays.  (a) synchronous pysdr_process per chunk, (b) the ingest ring with slots of B chunks
(pinned slots, async H2D / one launch sequence per slot / async D2H, results collected one slot late),
with the host memcpy that stands for readStream() filling the slot, and without it (a radio DMAs
straight into the pinned slot).  Prints one JSON line at the end."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_receivers
from pysdr_amd.ingest import IngestRing
from pysdr_amd.synth import CONFIGS, synth_iq

cfg = CONFIGS['C3']
BMAX = 64
P, rxs = build_receivers(cfg, 0, BMAX)
ctx = P._pysdr_stream
L = P.IN_CHUNK_SIZE
x = synth_iq(cfg, 8 * L, 10)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
res = {}

for _ in range(10):
    ctx.process_chunk(x[:L])
t0 = time.perf_counter()
for k in range(N // 4):
    ctx.process_chunk(x[(k % 8) * L:(k % 8 + 1) * L])
dt = (time.perf_counter() - t0) / (N // 4)
print(f"sync pysdr_process        : {L / dt / 1e6:9.1f} MS/s  ({dt * 1e3:.3f} ms per chunk)")
res["sync_ms_per_chunk"] = dt * 1e3

for B in (1, 4, 16, 64):
    for fill in (True, False):
        ring = IngestRing(ctx, 3, B)
        for s in range(3):                                    # slots pre-filled: the no-memcpy case reuses them
            for k in range(B):
                ring.buffer(s)[k * L:(k + 1) * L] = x[(k % 8) * L:(k % 8 + 1) * L]
        nslots = max(6, N // B)
        slot, pending = 0, None
        t0 = time.perf_counter()
        for j in range(nslots):
            if fill:
                buf = ring.buffer(slot)
                for k in range(B):                            # stands for readStream() filling the slot
                    buf[k * L:(k + 1) * L] = x[(k % 8) * L:(k % 8 + 1) * L]
            ring.submit(slot, B * L)
            if pending is not None:
                ring.collect(pending)
            pending, slot = slot, (slot + 1) % 3
        ring.collect(pending)
        dt = (time.perf_counter() - t0) / (nslots * B)
        print(f"ingest ring, {B:2d} chunks/slot, {'host memcpy into the slot' if fill else 'slot filled by DMA (no copy)'}: "
              f"{L / dt / 1e6:9.1f} MS/s  ({dt * 1e3:.4f} ms per chunk)")
        res[f"ring_b{B}_{'memcpy' if fill else 'dma'}_ms_per_chunk"] = dt * 1e3
        ring.close()
print(json.dumps(res))
