"""One sub-receiver with the reference's default 1001-tap prototype (params.py:134) at the rates of Tables.py:44-45 that have
a matrix-core shape (mixdec_mfma.hip: 1.024, 1.536, 1.792, 1.92, 2.048, 2.56 MS/s -> 48 kHz), batch resident in HBM: front-end kernel time and
fraction of the 8 TB/s HBM roofline, matrix-core form and (PYSDR_TUNING=1 PYSDR_MIXDEC_MFMA=0, in a child process) vector form.
    python scripts/long_prototype_rates.py"""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(fs):
    from pysdr_amd import _lib, sig_proc
    from pysdr_amd.params import RunTimeParams
    from pysdr_amd.synth import CONFIGS, synth_iq
    lib = _lib.lib()
    cfg = dict(CONFIGS['C1'], fs=fs, carriers=[dict(f=0.05 * fs, kind='am', amp=0.3, tone=1000.0, depth=0.5)])
    P = RunTimeParams(fs=fs, fsout=48e3, fc=[7e6], mode='AM', nfilt=1001, max_batch_chunks=1)
    L = P.IN_CHUNK_SIZE
    B = int(170e6 // L)
    P = RunTimeParams(fs=fs, fsout=48e3, fc=[7e6], mode='AM', nfilt=1001, max_batch_chunks=B)
    g = sig_proc.Receiver(P, 0.05 * fs, 0, '1')
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
    ms_all = []
    for _ in range(8):
        ctx.process_batch(d_x.value, B, L, on_device=True)
        _lib.check(lib.pysdr_sync(ctx.h), "sync")
        ms = C.c_float()
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, 0, C.byref(ms)), "el")
        ms_all.append(ms.value)
    ms = float(np.median(ms_all))
    nbytes = B * L * 8 + (B * L * P.UP // P.DOWN) * 8
    return dict(fs=fs, up=P.UP, down=P.DOWN, chunks=B, in_chunk=L, front_ms=ms, frac=nbytes / (ms * 1e-3) / 8e12)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(json.dumps(one(float(sys.argv[1]))))
        sys.exit(0)
    for fs in (1.024e6, 1.536e6, 1.792e6, 1.92e6, 2.048e6, 2.56e6):
        row = {}
        for name, env in (("matrix cores", {}), ("vector form", {"PYSDR_TUNING": "1", "PYSDR_MIXDEC_MFMA": "0"})):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(fs)], env=dict(os.environ, **env), stdout=subprocess.PIPE)
            row[name] = json.loads(p.stdout.decode().strip().splitlines()[-1])
        a, b = row["matrix cores"], row["vector form"]
        print(f"{fs / 1e6:5.3f} MS/s  {a['up']}/{a['down']}  {a['chunks']} chunks x {a['in_chunk']}:  matrix cores {a['front_ms']:.4f} ms = {a['frac']:.3f} of HBM,"
              f"  vector form {b['front_ms']:.4f} ms = {b['frac']:.3f}")
