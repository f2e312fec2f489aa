"""Throughput of the broadcast-FM path (config C4: 10 MS/s IQ, 1 RX, WFM2 stereo or WFM mono)
in batches resident in HBM; per-stage kernel times from the context's event ring."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
from pysdr_amd.synth import synth_wfm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.lib()
for mode in ('WFM', 'WFM2'):
    fs, L = 10e6, 213333
    P = RunTimeParams(fs=fs, fc=[98.1e6], mode=mode, nfilt=255, foffset=300e3, vid_bw=200e3, max_batch_chunks=B)
    g = sig_proc.Receiver(P, 300e3, 0, '1')
    ctx = P._pysdr_stream
    xu = synth_wfm(fs, 4 * L, 4)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d_x)), "alloc")
    for k in range(0, B, 4):
        n = min(4, B - k) * L
        _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + k * L * 8), C.c_void_p(xu.ctypes.data), n * 8), "up")
    for _ in range(2):
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    _lib.check(lib.pysdr_set_profile(ctx.h, 1), "prof")
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    dt = (time.perf_counter() - t0) / K
    ms = C.c_float()
    parts = []
    for which in (0, 1):
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, which, 0, C.byref(ms)), "el")
        parts.append(ms.value)
    print(f"{mode}: {B} chunks x {L}: {dt * 1e3:.3f} ms per batch = {B * L / dt / 1e9:.2f} GS/s; front+disc+pll+resample {parts[0]:.3f} ms, stage2 {parts[1]:.3f} ms")
    _lib.check(lib.pysdr_dev_free(0, d_x), "free")
    ctx.close()
