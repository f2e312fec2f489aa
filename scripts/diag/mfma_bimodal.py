"""Why the matrix-core front end of C1 runs in two modes on one box (0.274 or 0.285-0.297 ms per 4096 chunks):
where every workgroup ran (HW_REG_XCC_ID / HW_REG_HW_ID), how long it took on the shader clock (s_memtime) and on the
constant 100 MHz clock (s_memrealtime) -> the shader clock each workgroup really had, per launch, over several contexts
(fresh allocations of the input batch) in ONE process, next to the launch's HIP-event time.  DIAGNOSTIC build:
    python -m pysdr_amd.build --diag
    PYSDR_TUNING=1 PYSDR_USE_DIAG_LIB=1 PYSDR_DEBUG_FLAGS=512 python scripts/diag/mfma_bimodal.py [contexts] [launches]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
from pysdr_amd.synth import CONFIGS, synth_iq

NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 6
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 4096
lib = _lib.lib()
cfg = CONFIGS['C1']
fn = lib.pysdr_diag_mfma_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p]
keep = []                                   # some contexts keep their batch allocated: the next one lands elsewhere
for ic in range(NCTX):
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode='AM', nfilt=cfg['ntaps_dec'], max_batch_chunks=B)
    g = sig_proc.Receiver(P, 100e3, 0, '1')
    ctx = P._pysdr_stream
    L = P.IN_CHUNK_SIZE
    xu = synth_iq(cfg, 8 * L, 3)
    pad = (ic * 3) % 5                      # different offsets of the batch inside its allocation (2 MB steps)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, B * L * 8 + pad * (2 << 20), C.byref(d_x)), "alloc")
    base = d_x.value + pad * (2 << 20)
    for k in range(0, B, 8):
        _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(base + k * L * 8), C.c_void_p(xu.ctypes.data), 8 * L * 8), "up")
    _lib.check(lib.pysdr_set_profile(ctx.h, 1), "profile")
    times = []
    for it in range(NL):
        ctx.process_batch(base, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    ms = C.c_float(0)
    for back in range(min(NL, 32)):
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, back, C.byref(ms)), "elapsed")
        times.append(ms.value)
    st = np.zeros((1024, 24), dtype=np.uint64)
    _lib.check(fn(ctx.h, st.ctypes.data), "stamps")
    st = st[:256].astype(np.int64)
    xcc = st[:, 0]
    hw = st[:, 1]
    cu, se = (hw >> 8) & 15, (hw >> 13) & 7
    dt_clk = st[:, 3] - st[:, 2]
    dt_real = (st[:, 5] - st[:, 4]) * 10e-9           # 100 MHz ticks -> seconds
    ghz = dt_clk / np.maximum(dt_real, 1e-12) / 1e9
    t0 = st[:, 4] - st[:, 4].min()
    print(f"ctx {ic}: batch at 0x{base:x} (offset {pad} x 2 MB); front end ms: median {np.median(times):.4f} min {min(times):.4f} max {max(times):.4f}")
    print(f"   last launch: workgroup i on XCC {xcc[:16].tolist()} ... ; per XCC: " +
          " ".join(f"{x}:{int((xcc == x).sum())}" for x in range(8)))
    print(f"   workgroup duration (100 MHz clock) us: min {dt_real.min() * 1e6:.1f} median {np.median(dt_real) * 1e6:.1f} max {dt_real.max() * 1e6:.1f};"
          f" start skew max {t0.max() * 10e-3:.1f} us; shader clock GHz: min {ghz.min():.3f} median {np.median(ghz):.3f} max {ghz.max():.3f}")
    per_xcc = [f"{x}: {np.median(dt_real[xcc == x]) * 1e6:.0f} us @ {np.median(ghz[xcc == x]):.2f} GHz" for x in range(8) if (xcc == x).any()]
    print("   per XCC median: " + "; ".join(per_xcc))
    # where the waves stand: cycles at the tile loop's barrier as a fraction of the workgroup's run (C1: waves 0-7 consumers,
    # 8-13 copies, 14-15 epilogue)
    bar = st[:, 6:22]
    frac_b = (bar & 0xFFFFFFFF) / np.maximum(dt_clk, 1)[:, None]
    frac_w = (bar >> 32) / np.maximum(dt_clk, 1)[:, None]
    print("   barrier wait, fraction of the run, mean over workgroups: consumers " + " ".join(f"{v:.2f}" for v in frac_b[:, :8].mean(axis=0)) +
          " | copy waves " + " ".join(f"{v:.2f}" for v in frac_b[:, 8:14].mean(axis=0)) + " (+ waiting for their copies " +
          " ".join(f"{v:.2f}" for v in frac_w[:, 8:14].mean(axis=0)) + ") | epilogue " + " ".join(f"{v:.2f}" for v in frac_b[:, 14:16].mean(axis=0)))
    distinct = len({(int(a), int(b), int(c_)) for a, b, c_ in zip(xcc, se, cu)})
    print(f"   distinct (XCC, SE, CU) of the 256 workgroups: {distinct}")
    if ic % 2 == 0:
        keep.append((ctx, d_x))
    else:
        ctx.close()
        lib.pysdr_dev_free(0, d_x)

# ---- per-launch: event time against the shader clock the launch really had (one context, a sync + a stamp read per launch;
#      `pace` = seconds of host sleep between launches: a GPU that idles between launches changes its power state)
import time
ctx, d_x = keep[0]
base = d_x.value
rows = []
for pace in (0.0, 0.0, 0.002, 0.02):
    for it in range(24):
        ctx.process_batch(base, B, L, on_device=True)
        _lib.check(lib.pysdr_sync(ctx.h), "sync")
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, 0, C.byref(ms)), "elapsed")
        st = np.zeros((1024, 24), dtype=np.uint64)
        _lib.check(fn(ctx.h, st.ctypes.data), "stamps")
        st = st[:256].astype(np.int64)
        real = (st[:, 5] - st[:, 4]) * 10e-9
        ghz = (st[:, 3] - st[:, 2]) / np.maximum(real, 1e-12) / 1e9
        rows.append((pace, ms.value, float(np.median(real)) * 1e6, float(np.median(ghz))))
        if pace:
            time.sleep(pace)
rows = np.array(rows)
print("per launch (one context, sync after every launch): pace s | launches | event ms min / median / max | shader GHz min / median / max | corr(ms, GHz)")
for pace in (0.0, 0.002, 0.02):
    r = rows[rows[:, 0] == pace]
    cc = np.corrcoef(r[:, 1], r[:, 3])[0, 1]
    print(f"   {pace:5.3f} | {len(r):3d} | {r[:, 1].min():.4f} / {np.median(r[:, 1]):.4f} / {r[:, 1].max():.4f} | "
          f"{r[:, 3].min():.3f} / {np.median(r[:, 3]):.3f} / {r[:, 3].max():.3f} | {cc:+.2f}")
lo = rows[rows[:, 1] <= np.percentile(rows[:, 1], 25)]
hi = rows[rows[:, 1] >= np.percentile(rows[:, 1], 75)]
print(f"   fastest quarter of the launches: {lo[:, 1].mean():.4f} ms at {lo[:, 3].mean():.3f} GHz; slowest quarter: {hi[:, 1].mean():.4f} ms at {hi[:, 3].mean():.3f} GHz")
