"""mix+decimate time against the number of sub-receivers sharing one 8 MS/s stream (255-tap
prototype, 1024 chunks resident in HBM): the input is read once whatever the count."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pysdr_amd import _lib
from pysdr_amd.synth import CONFIGS, synth_iq
B = 1024
lib = _lib.lib()
base = CONFIGS['C3']
for nrx in (1, 2, 4, 6, 8):
    rx = [dict(base['rx'][i % 4], frq=base['rx'][i % 4]['frq'] + 20e3 * (i // 4)) for i in range(nrx)]
    cfg = dict(base, rx=rx)
    P, rxs = bench.build_receivers(cfg, 0, B)
    ctx = P._pysdr_stream
    L = P.IN_CHUNK_SIZE
    xu = synth_iq(cfg, 8 * L, 3)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d_x)), "alloc")
    for k in range(0, B, 8):
        _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + k * L * 8), C.c_void_p(xu.ctypes.data), 8 * L * 8), "up")
    for _ in range(2):
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync"); _lib.check(lib.pysdr_set_profile(ctx.h, 1), "prof")
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    dt = (time.perf_counter() - t0) / 5
    ms = C.c_float(); _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, 0, C.byref(ms)), "el")
    ms2 = C.c_float(); _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 1, 0, C.byref(ms2)), "el")
    nb = B * L * 8 + nrx * (B * L * 3 // 500) * 8
    print(f"{nrx} RX: {B*L/dt/1e9:6.1f} GS/s; mixdec {ms.value:.3f} ms = {nb/ms.value/1e6:.0f} GB/s ({nb/ms.value/1e6/80:.0f} % of HBM); stage 2 {ms2.value:.3f} ms")
    _lib.check(lib.pysdr_dev_free(0, d_x), "free")
    ctx.close()
