"""Batch throughput with one sub-receiver in AM-Synch (serial carrier PLL at FS_OUT)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
from pysdr_amd.synth import CONFIGS, synth_iq
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.lib()
cfg = CONFIGS['C1']
P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode='AM', nfilt=255, max_batch_chunks=B)
g = sig_proc.Receiver(P, 100e3, 0, '1')
g.mode = 'AM-Synch'
ctx = P._pysdr_stream
L = P.IN_CHUNK_SIZE
xu = synth_iq(cfg, 4 * L, 3)
d_x = C.c_void_p()
_lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d_x)), "alloc")
for k in range(0, B, 4):
    n = min(4, B - k) * L
    _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + k * L * 8), C.c_void_p(xu.ctypes.data), n * 8), "up")
ctx.process_batch(d_x.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
t0 = time.perf_counter()
for _ in range(3):
    ctx.process_batch(d_x.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
dt = (time.perf_counter() - t0) / 3
nout = B * L * P.UP // P.DOWN
print(f"AM-Synch: {B} chunks x {L} @ {cfg['fs']/1e6} MS/s: {dt*1e3:.2f} ms per batch = {B*L/dt/1e9:.3f} GS/s; {dt/nout*1e9:.1f} ns per output sample")
