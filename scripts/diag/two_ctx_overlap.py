"""What would overlapping stage 2 of call k with the front end of call k+1 buy?  Upper-bound experiment with NO library
change: two contexts (= two HIP streams) in one process work through the SAME resident batch, their calls enqueued
alternately, so the front end of one runs beside the audio-rate stages (PLL walks, AF FIR, AGC) of the other whenever
the hardware lets them.  Aggregate rate of the pair against one context alone = what a two-stream pipeline inside one
context could reach (VERDICT r4 "Next round" 2 and 5: measure the co-residency with the real kernels).

    python scripts/diag/two_ctx_overlap.py [workload ...]     (default: c1 c1synch c2 rx6 c4mono c4)
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pysdr_amd import _lib  # noqa: E402

lib = _lib.lib()
_lib.require_gpu()


def run(w, nctx, steps):
    args = bench.parse(["--workload", w])
    cfg = bench.workload_cfg(args)
    B = bench.DEFAULT_CHUNKS[w]
    ctxs = []
    for _ in range(nctx):
        P, rxs = bench.build_receivers(cfg, 0, B)
        ctxs.append((P, rxs, P._pysdr_stream))
    L = ctxs[0][0].IN_CHUNK_SIZE
    nsamp = B * L
    nloop = 1700000 if 'wfm' in cfg else 8 * L
    xu = bench.synth_batch(cfg, nloop, 10)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, nsamp * 8, C.byref(d_x)), "alloc")
    for off in range(0, nsamp, nloop):
        n = min(nloop, nsamp - off)
        _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + off * 8), C.c_void_p(xu.ctypes.data), n * 8), "up")

    def sync():
        for _, _, c in ctxs:
            _lib.check(lib.pysdr_sync(c.h), "sync")
    for _ in range(max(8, steps // 4)):
        for _, _, c in ctxs:
            c.process_batch(d_x.value, B, L, on_device=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for _, _, c in ctxs:
            c.process_batch(d_x.value, B, L, on_device=True)
    sync()
    dt = time.perf_counter() - t0
    for _, _, c in ctxs:
        c.close()
    lib.pysdr_dev_free(0, d_x)
    return nctx * steps * nsamp / dt / 1e9, dt / (nctx * steps) * 1e3


for w in (sys.argv[1:] or ["c1", "c1synch", "c2", "rx6", "c4mono", "c4"]):
    steps = bench.MIN_STEPS.get(w, 30)
    one, ms1 = run(w, 1, steps)
    two, ms2 = run(w, 2, steps // 2)
    print(f"{w:8s} one context {one:7.1f} GS/s ({ms1:.3f} ms per call)   two contexts interleaved {two:7.1f} GS/s ({ms2:.3f} ms per call)   x{two / one:.3f}", flush=True)
