"""Diagnostic: GPU vs float32 and float64 oracle errors per chunk (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import sdr_oracle as so
from tests.test_gpu_parity import make_gpu_receivers, relerr

def run(cfg, L, n, seed):
    x = so.synth_iq(cfg, n * L, seed)
    P, g = make_gpu_receivers(cfg)
    o32 = so.make_receivers(cfg, np.float32)
    o64 = so.make_receivers(cfg, np.float64)
    for k in range(n):
        xc = x[k * L:(k + 1) * L]
        for i in range(len(g)):
            a = g[i].demod_data(xc); b = o32[i].demod_data(xc); c = o64[i].demod_data(xc)
            print(k, o32[i].mode, "iq g-32 %.2e g-64 %.2e 32-64 %.2e | am g-32 %.2e g-64 %.2e 32-64 %.2e" % (
                relerr(g[i].iq, o32[i].iq), relerr(g[i].iq, o64[i].iq), relerr(o32[i].iq, o64[i].iq),
                relerr(a, b), relerr(a, c), relerr(b, c)))

cfg = dict(so.CONFIGS['C2'], fs=10e6, ntaps_dec=1001, carriers=[dict(f=455e3, kind='fm', amp=0.3, tone=1000.0, dev=3000.0)])
run(cfg, 213333, 3, 9)
run(so.CONFIGS['C2'], 170666, 2, 2)
run(so.CONFIGS['C3'], 170666, 2, 3)
