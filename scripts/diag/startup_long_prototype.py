import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import sdr_oracle as so
from tests.test_gpu_parity import make_gpu_receivers
cfg = dict(so.CONFIGS['C2'], fs=10e6, ntaps_dec=1001, carriers=[dict(f=455e3, kind='fm', amp=0.3, tone=1000.0, dev=3000.0)])
L = so.chunk_sizes(10e6, 48e3)[3]
x = so.synth_iq(cfg, 2 * L, 9)
P, g = make_gpu_receivers(cfg)
o32 = so.make_receivers(cfg, np.float32)[0]; o64 = so.make_receivers(cfg, np.float64)[0]
ag, yg, a32, y32, a64, y64 = [], [], [], [], [], []
for k in range(2):
    xc = x[k*L:(k+1)*L]
    ag.append(g[0].demod_data(xc).copy()); yg.append(g[0].iq.copy())
    a32.append(o32.demod_data(xc)); y32.append(o32.iq.copy())
    a64.append(o64.demod_data(xc)); y64.append(o64.iq.copy())
ag, yg, a32, y32, a64, y64 = map(np.concatenate, (ag, yg, a32, y32, a64, y64))
pk = np.abs(a64).max(); ypk = np.abs(y64).max()
dg, do = np.abs(ag-a32)/pk, np.abs(a32-a64)/pk
print("audio gpu-o32: count>1e-5", int((dg>1e-5).sum()), "last", np.nonzero(dg>1e-5)[0][-1:] , "max", dg.max())
print("audio o32-o64: count>1e-5", int((do>1e-5).sum()), "last", np.nonzero(do>1e-5)[0][-1:], "max", do.max())
print("iq gpu-o32 max", (np.abs(yg-y32)/ypk).max(), " |y|^2/peak first 6:", (np.abs(y64[:6])**2/ypk**2))
print("iq o32-o64 first 4 rel to own magnitude:", np.abs(y32[:4]-y64[:4])/np.abs(y64[:4]), " gpu:", np.abs(yg[:4]-y64[:4])/np.abs(y64[:4]))
from tests.test_gpu_parity import nfm_rounding_allowance
allow = nfm_rounding_allowance(o64, y64)
err = np.abs(ag - a32)
ntaps = o64.demod.ntaps
ybuf = np.concatenate((np.zeros(ntaps + 1, np.complex128), y64))
scale = o64.demod.fs_out / (2 * np.pi * so.NFM_FULL_SCALE_DEV)
d = so.nfm_discriminator(ybuf, np.float64) * scale
big = np.nonzero(np.abs(d) > 4)[0]
print("detector samples > 4 full scales: idx (in d)", big[:10], "values", d[big][:10])
print("steady peak", np.abs(a64[400:]).max(), "transient peak", np.abs(a64[:300]).max())
k = np.argmax(err - allow)
print("worst excess at", k, "err", err[k], "allow", allow[k], "a64", a64[k], "ratio err/allow over first 256: max", np.max(err[:256] / np.maximum(allow[:256], 1e-30)))
print("err/|a64| first 256 max", np.max(err[:256] / np.abs(a64[:256])))
