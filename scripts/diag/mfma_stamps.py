"""Where a tile period of the matrix-core mix+decimate kernel (mixdec_mfma.hip) goes: s_memtime stamps from the
DIAGNOSTIC build (python -m pysdr_amd.build --diag).
    PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=256 python scripts/diag/mfma_stamps.py [chunks]
Stamps (lane 0 of every wave of workgroups 3 and 131, first 24 tiles): 0 loop top, 1 after the wait for the tile's
copies (producers), 2 after the barrier, 3 after issuing the next tile's copies, 4 after the epilogue of the previous
tile (producers), 5 after the MFMA chain (consumers), 6 end of the trip; 7 = HW_ID (SIMD the wave runs on)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for _ in range(3):
    ctx.process_batch(d_x.value, B, L, on_device=True)
_lib.check(lib.pysdr_sync(ctx.h), "sync")
st = np.zeros((2, 16, 24, 8), dtype=np.uint64)
fn = lib.pysdr_diag_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p]
_lib.check(fn(ctx.h, st.ctypes.data), "stamps")
st = st.astype(np.int64)
for wg in range(2):
    live = [w for w in range(16) if st[wg][w, 5, 0] != 0]
    s = st[wg][live][:, 4:23, :]
    per = s[:, 1:, 0] - s[:, :-1, 0]
    print(f"workgroup {('3', '131')[wg]}: {len(live)} waves, tile period {per.mean():.0f} cycles (min {per.min()}, max {per.max()})")
    hw = st[wg][live][:, 5, 7]
    print("   SIMD of wave:", " ".join(str((int(v) >> 4) & 3) for v in hw), "  CU:", sorted(set((int(v) >> 8) & 15 for v in hw)))
    names = ["0->1 wait copies", "1->2 barrier", "2->3 issue copies", "3->4 epilogue", "2->5 MFMA chain", "..->6 rest"]
    def col(a, b):
        return (s[:, :, b] - s[:, :, a])
    rows = [col(0, 1), col(1, 2), col(2, 3), col(3, 4), col(2, 5), None]
    for nm, d in zip(names, rows):
        if d is None:
            continue
        print(f"   {nm:18s} per wave: " + " ".join(f"{v:6.0f}" for v in d.mean(axis=1)))
    print(f"   {'2->6 whole trip':18s} per wave: " + " ".join(f"{v:6.0f}" for v in col(2, 6).mean(axis=1)))
    arrive = s[:, :, 1]
    print(f"   spread of arrivals at the barrier (max - min over waves), mean over tiles: {(arrive.max(axis=0) - arrive.min(axis=0)).mean():.0f}")
