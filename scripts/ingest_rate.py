"""PCIe-inclusive rate of the chunk-by-chunk host path (never the bench `value`): C3, 4 RX,
chunks handed over as host arrays.  (a) synchronous pysdr_process per chunk, (b) the ingest
ring (pinned slots, async H2D / kernels / D2H, results collected one chunk late)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_receivers
from pysdr_amd.ingest import IngestRing
from pysdr_amd.synth import CONFIGS, synth_iq

cfg = CONFIGS['C3']
P, rxs = build_receivers(cfg, 0, 1)
ctx = P._pysdr_stream
L = P.IN_CHUNK_SIZE
x = synth_iq(cfg, 8 * L, 10)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400

for _ in range(10):
    ctx.process_chunk(x[:L])
t0 = time.perf_counter()
for k in range(N):
    ctx.process_chunk(x[(k % 8) * L:(k % 8 + 1) * L])
dt = time.perf_counter() - t0
print(f"sync pysdr_process : {N * L / dt / 1e6:9.1f} MS/s  ({dt / N * 1e3:.3f} ms per chunk)")

ring = IngestRing(ctx, 3)
slot, pending = 0, None
t0 = time.perf_counter()
for k in range(N):
    ring.buffer(slot)[:] = x[(k % 8) * L:(k % 8 + 1) * L]     # stands for readStream() filling the slot
    ring.submit(slot, L)
    if pending is not None:
        ring.collect(pending)
    pending, slot = slot, (slot + 1) % 3
ring.collect(pending)
dt = time.perf_counter() - t0
print(f"ingest ring        : {N * L / dt / 1e6:9.1f} MS/s  ({dt / N * 1e3:.3f} ms per chunk, incl. the host memcpy into the slot)")
ring.close()
