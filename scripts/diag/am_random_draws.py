# which start paths (linear / walked) and how many patched segments the draws of tests/test_gpu_pll.py::test_am_synch_random_carriers_... exercise (GPU box)
import sys; sys.path.insert(0,'.')
import numpy as np
from tests import test_gpu_pll as T
from oracle import sdr_oracle as so
# replicate the draws and print which paths ran
import ctypes as C
from pysdr_amd import _lib, sig_proc
from pysdr_amd.params import RunTimeParams
for seed in range(10):
    rng = np.random.default_rng(1000 + seed)
    cfg = dict(so.CONFIGS['C1']); cfg['ntaps_dec']=255
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    B = int(rng.choice([13, 24, 40, 61, 97])); f_off=float(rng.uniform(-35,35)); noise=float(rng.choice([1e-3,1e-2,5e-2,0.15])); depth=float(rng.uniform(0.3,0.97))
    n=3*B*L
    jumps = tuple((int(rng.integers(L, n - L)), float(rng.uniform(-3.0, 3.0))) for _ in range(int(rng.integers(0, 3))))
    x = T._carrier_stream(n, cfg['fs'], f_off, noise, jumps, seed=50 + seed, depth=depth)
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode='AM-Synch', nfilt=255, max_batch_chunks=B); P.VIDEO_BW=10e3
    g = sig_proc.Receiver(P, 100e3, 0, '1'); g.mode, g.af_bw = 'AM-Synch', 5e3
    ctx = P._pysdr_stream; info=[]
    for h in range(3):
        ctx.process_batch(x[h*B*L:(h+1)*B*L], B, L, on_device=False); ctx.fetch(0,B)
        info.append((T.pll_stats(ctx), T._linear_starts(ctx)))
    ctx.close()
    print(seed, "B", B, "f_off %.1f noise %.3g depth %.2f jumps %d" % (f_off, noise, depth, len(jumps)), "(segments, patched), linear:", info)
