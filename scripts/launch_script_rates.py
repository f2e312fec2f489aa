"""The mix + decimate front end at the operating points of the reference's own launch scripts and rate tables: the SDRplay rates
of Tables.py:45 (1 .. 10 MS/s) x 1 .. 6 USB sub-receivers on the DEFAULT 1001-tap prototype (params.py:134) -> 48 kHz, batch
resident in HBM; front-end kernel time by HIP events, fraction of the 8 TB/s HBM roofline by SURVEY 8(d)'s bytes
(8 + R (UP/DOWN) 8 per input sample), and which instantiation ran.
    python scripts/launch_script_rates.py [fs_MHz nrx]          (profiles/r06_launch_script_rates.txt)
Launch scripts: FT8:42,70 (1 MS/s, 1 RX), FT8FT4:22,34 (1 MS/s, 2 RX), FT8dual:30,43 (5 MS/s, 2 RX), TEST:13-32 (4 MS/s, 2 RX),
FT8follow:10,18 / FT8tri:47-74 (8 MS/s, 3 RX), FT8FT4dual:20,32 (8 MS/s, 4 RX), FT8FT4tri:20,32 (8 MS/s, 6 RX)."""
import ctypes as C, json, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

POINTS = [(1e6, 1, "FT8"), (1e6, 2, "FT8FT4"), (2e6, 1, ""), (2e6, 2, ""), (3e6, 2, ""), (4e6, 2, "TEST"), (5e6, 2, "FT8dual"), (6e6, 2, ""),
          (6e6, 3, ""), (7e6, 3, ""), (8e6, 1, "FT8rtl-style single"), (8e6, 2, ""), (8e6, 3, "FT8tri, FT8follow"), (8e6, 4, "FT8FT4dual"),
          (8e6, 6, "FT8FT4tri"), (9e6, 3, ""), (10e6, 3, ""), (10e6, 4, "")]


def one(fs, nrx):
    from pysdr_amd import _lib, sig_proc
    from pysdr_amd.params import RunTimeParams
    from pysdr_amd.synth import synth_iq
    lib = _lib.lib()
    frqs = [(-0.35 + 0.7 * (i + 0.5) / nrx) * fs for i in range(nrx)]
    cfg = dict(fs=fs, fs_out=48e3, ntaps_dec=1001, noise=2e-3, carriers=[dict(f=f, kind='usb', amp=0.1, tone=1000.0 + 100 * i) for i, f in enumerate(frqs)],
               rx=[dict(frq=f, mode='USB', video_bw=45e3, af_bw=5e3) for f in frqs])
    P = RunTimeParams(fs=fs, fsout=48e3, fc=[7e6] * nrx, mode='USB', nfilt=1001, max_batch_chunks=1)
    L = P.IN_CHUNK_SIZE
    B = int(340e6 // L) if fs >= 4e6 else int(170e6 // L)
    P = RunTimeParams(fs=fs, fsout=48e3, fc=[7e6] * nrx, mode='USB', nfilt=1001, max_batch_chunks=B)
    for i, f in enumerate(frqs):
        P.VIDEO_BW = 45e3
        rx = sig_proc.Receiver(P, f, i, str(i + 1))
        rx.mode, rx.af_bw = 'USB', 5e3
    ctx = P._pysdr_stream
    xu = synth_iq(cfg, 8 * L, 3)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, B * L * 8, C.byref(d_x)), "alloc")
    for k in range(0, B, 8):
        n = min(8, B - k) * L
        _lib.check(lib.pysdr_dev_upload(0, C.c_void_p(d_x.value + k * L * 8), C.c_void_p(xu.ctypes.data), n * 8), "up")
    for _ in range(3):
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    _lib.check(lib.pysdr_set_profile(ctx.h, 1), "prof")
    fr, tot = [], []
    for _ in range(24):                                   # back to back, as a replay runs them (a sync per call lets the clocks sag: -10 %)
        ctx.process_batch(d_x.value, B, L, on_device=True)
    _lib.check(lib.pysdr_sync(ctx.h), "sync")
    for back in range(12):
        ms = C.c_float()
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, back, C.byref(ms)), "el")
        fr.append(ms.value)
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 3, back, C.byref(ms)), "el")      # period between consecutive calls
        tot.append(ms.value)
    ms, mt = float(np.median(fr)), float(np.median(tot))
    n_in = B * L
    nbytes = n_in * 8 + nrx * (n_in * P.UP // P.DOWN) * 8
    return dict(fs=fs, nrx=nrx, up=P.UP, down=P.DOWN, chunks=B, in_chunk=L, front_ms=ms, call_ms=mt, frac=nbytes / (ms * 1e-3) / 8e12,
                gsps=n_in / (mt * 1e-3) / 1e9, job_frac=(n_in * 8 + nrx * (n_in * P.UP // P.DOWN) * 12) / (mt * 1e-3) / 8e12)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        print(json.dumps(one(float(sys.argv[1]) * 1e6, int(sys.argv[2]))))
        sys.exit(0)
    print("# fs, RX: UP/DOWN, taps per branch | front end ms, fraction of 8 TB/s | whole call GS/s, job fraction | launch script")
    for fs, nrx, what in POINTS:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(fs / 1e6), str(nrx)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            a = json.loads(p.stdout.decode().strip().splitlines()[-1])
            print(f"{fs / 1e6:5.1f} MS/s x {nrx} RX: {a['up']}/{a['down']}, {-(-1001 // a['up'])} per branch | {a['front_ms']:.4f} ms  {a['frac']:.3f} | "
                  f"{a['gsps']:.0f} GS/s  {a['job_frac']:.3f} | {what}", flush=True)
        except Exception as e:
            print(f"{fs / 1e6:5.1f} MS/s x {nrx} RX: FAILED {e} {p.stderr.decode()[-300:]}", flush=True)
