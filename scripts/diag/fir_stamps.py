"""Diagnostic build only (python -m pysdr_amd.build --diag; PYSDR_USE_DIAG_LIB=1): phase stamps of one
workgroup of demod_fir_kernel in the middle of a C3 / C2 batch."""
import ctypes as C, os, sys
os.environ["PYSDR_USE_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bench import build_receivers, synth_batch
from pysdr_amd import _lib
from pysdr_amd.synth import CONFIGS
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = CONFIGS[wl]
B = 2048
P, rxs = build_receivers(cfg, 0, B)
ctx = P._pysdr_stream
L = P.IN_CHUNK_SIZE
lib = _lib.lib()
x = synth_batch(cfg, 8 * L, 10)
d = C.c_void_p()
_lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d)), "alloc")
for k in range(0, B, 8):
    _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d.value + k * L * 8), C.c_void_p(x.ctypes.data), 8 * L * 8), "up")
for _ in range(3):
    ctx.process_batch(d.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
st = (C.c_ulonglong * 64)()
f = lib.pysdr_diag_fir_stamps
f.restype = C.c_int
assert f(st) == 0
t = np.array(list(st), dtype=np.int64).reshape(8, 8)
names = ["start", "staged (loads+detect+LDS writes)", "barrier", "FIR loop", "peaks/atomics", "transposed stores"]
nw = 4
for i in range(1, 6):
    print("%-36s wave0 +%7d   all waves: %s" % (names[i], t[i, 0] - t[i - 1, 0], [int(t[i, w] - t[0, w]) for w in range(nw)]))
