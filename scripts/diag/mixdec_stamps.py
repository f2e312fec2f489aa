"""Where a tile period of the mix+decimate kernel goes: s_memtime stamps of the tile loop's phases from the
DIAGNOSTIC build (python -m pysdr_amd.build --diag; PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=256).
    PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=256 python scripts/diag/mixdec_stamps.py [c1|c2|c3] [chunks]
Stamps (lane 0 of every wave of workgroups 3 and 131, first 24 tiles): 0 loop top, 1 after the wait for the tile's
copies, 2 after the barrier, 3 after issuing the next tile's copies, 4 after the raw-peak scan, 5 after the dot
products, 6 after the (occasional) flush of the output stage."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
from pysdr_amd.synth import CONFIGS, synth_iq
wl = sys.argv[1] if len(sys.argv) > 1 else 'c1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (4096 if wl in ('c1', 'test2rx') else 2048)
lib = _lib.lib()
if wl == 'c1':
    cfg = CONFIGS['C1']
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode='AM', nfilt=cfg['ntaps_dec'], max_batch_chunks=B)
    g = sig_proc.Receiver(P, 100e3, 0, '1')
elif wl in ('ft8tri', 'test2rx'):
    cfg = CONFIGS[wl.upper()]
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[14e6] * len(cfg['rx']), mode=cfg['rx'][0]['mode'], nfilt=cfg['ntaps_dec'], max_batch_chunks=B)
    gs = []
    for i, r in enumerate(cfg['rx']):
        P.VIDEO_BW = r['video_bw']
        gs.append(sig_proc.Receiver(P, r['frq'], i, str(i + 1)))
        gs[-1].mode = r['mode']
    g = gs[0]
elif wl == 'c3':
    cfg = CONFIGS['C3']
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[146e6], mode='USB', nfilt=255, max_batch_chunks=B)
    gs = [sig_proc.Receiver(P, f, i, str(i + 1)) for i, f in enumerate((200e3, -310e3, 455e3, -1.2e6))]
    for g_, m in zip(gs, ('USB', 'CW', 'NFM', 'AM')):
        g_.mode = m
    g = gs[0]
else:
    cfg = CONFIGS['C2']
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[146e6], mode='NFM', nfilt=255, max_batch_chunks=B)
    g = sig_proc.Receiver(P, 455e3, 0, '1')
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
names = ["wait for copies", "barrier", "issue next copies", "peak scan", "dot products", "flush"]
for wg in range(2):
    s = st[wg][:, 4:23, :]                       # waves x tiles x stamps
    per = (s[:, 1:, 0] - s[:, :-1, 0])           # tile period per wave
    print(f"workgroup {('3', '131')[wg]}: tile period {per.mean():.0f} cycles (min {per.min()}, max {per.max()})")
    d = np.diff(s[:, :, :7], axis=2)             # waves x tiles x 6 phases
    tail = s[:, 1:, 0] - s[:, :-1, 6]
    for k, nm in enumerate(names):
        print(f"   {nm:18s} mean {d[:, :, k].mean():8.0f}   per wave: " + " ".join(f"{v:6.0f}" for v in d[:, :, k].mean(axis=1)))
    print(f"   {'loop back':18s} mean {tail.mean():8.0f}")
    # how far apart the waves arrive at the barrier, and the span of the dot-product phase over the workgroup
    arrive = s[:, :, 1]
    print(f"   spread of arrivals at the barrier (max - min over waves), mean over tiles: {(arrive.max(axis=0) - arrive.min(axis=0)).mean():.0f}")
    done = s[:, :, 5]
    print(f"   spread of dot-product completion over waves: {(done.max(axis=0) - done.min(axis=0)).mean():.0f}")
