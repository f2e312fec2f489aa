"""Batch throughput of configuration C1 (am.py path: 2.048 MS/s, 1 RX AM, the reference's
default 1001-tap prototype) resident in HBM."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
from pysdr_amd.synth import CONFIGS, synth_iq
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.lib()
cfg = CONFIGS['C1']
P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode='AM', nfilt=cfg['ntaps_dec'], max_batch_chunks=B)
g = sig_proc.Receiver(P, 100e3, 0, '1')
ctx = P._pysdr_stream
L = P.IN_CHUNK_SIZE
xu = synth_iq(cfg, 8 * L, 3)
d_x = C.c_void_p()
_lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d_x)), "alloc")
for k in range(0, B, 8):
    n = min(8, B - k) * L
    _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + k * L * 8), C.c_void_p(xu.ctypes.data), n * 8), "up")
for _ in range(2):
    ctx.process_batch(d_x.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
_lib.check(lib.pysdr_set_profile(ctx.h, 1), "prof")
t0 = time.perf_counter()
for _ in range(5):
    ctx.process_batch(d_x.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
dt = (time.perf_counter() - t0) / 5
ms = C.c_float()
_lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, 0, C.byref(ms)), "el")
print(f"C1: {B} chunks x {L}: {dt*1e3:.3f} ms per batch = {B*L/dt/1e9:.1f} GS/s; mixdec {ms.value:.3f} ms = {B*L*8/ms.value/1e6:.0f} GB/s")
