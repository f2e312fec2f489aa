"""Where do the start-up samples of NFM / WFM really differ between the GPU and the float32 oracle,
and is that where the float32 and float64 oracles differ (ill-conditioning), or a bug hiding
behind a wholesale skip?  """
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import sdr_oracle as so
from oracle import wfm_oracle as wo
from tests.test_gpu_parity import make_gpu_receivers

cfg = so.CONFIGS['C2']
L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
x = so.synth_iq(cfg, 2 * L, 2)
P, g = make_gpu_receivers(cfg)
o32 = so.make_receivers(cfg, np.float32)[0]
o64 = so.make_receivers(cfg, np.float64)[0]
ag = np.concatenate([g[0].demod_data(x[k * L:(k + 1) * L]).copy() for k in range(2)])
yg = None
a32, y32 = [], []
for k in range(2):
    a32.append(o32.demod_data(x[k * L:(k + 1) * L])); y32.append(o32.iq.copy())
a32, y32 = np.concatenate(a32), np.concatenate(y32)
a64 = np.concatenate([o64.demod_data(x[k * L:(k + 1) * L]) for k in range(2)])
pk = np.max(np.abs(a64))
dg, do = np.abs(ag - a32) / pk, np.abs(a32 - a64) / pk
print("NFM: gpu-vs-o32 > 1e-5 at", np.nonzero(dg > 1e-5)[0][:20], "count", int((dg > 1e-5).sum()), "max", dg.max(), "argmax", int(dg.argmax()))
print("NFM: o32-vs-o64 max", do.max())
print("NFM: |y|^2 first 6 rel to peak", (np.abs(y32[:6]) ** 2 / np.max(np.abs(y32) ** 2)))
print("NFM: dg first 8", dg[:8], " dg[250:262]", dg[250:262])

from pysdr_amd import sig_proc
from pysdr_amd.params import RunTimeParams
fs, L4 = 10e6, 213333
xw = wo.synth_wfm(fs, 3 * L4, 4)
for stereo in (False, True):
    Pw = RunTimeParams(fs=fs, fc=[98.1e6], mode='WFM2' if stereo else 'WFM', nfilt=255, foffset=300e3, vid_bw=200e3)
    gw = sig_proc.Receiver(Pw, 300e3, 0, '1')
    w32 = wo.WfmReceiver(fs, 48e3, 300e3, stereo=stereo, ntaps_dec=255, dtype=np.float32)
    w64 = wo.WfmReceiver(fs, 48e3, 300e3, stereo=stereo, ntaps_dec=255, dtype=np.float64)
    bg = np.concatenate([gw.demod_data(xw[k * L4:(k + 1) * L4]).copy() for k in range(3)])
    b32 = np.concatenate([w32.demod_data(xw[k * L4:(k + 1) * L4]) for k in range(3)])
    b64 = np.concatenate([w64.demod_data(xw[k * L4:(k + 1) * L4]) for k in range(3)])
    pk = np.max(np.abs(b64))
    dg, do = np.abs(bg - b32) / pk, np.abs(b32 - b64) / pk
    bad = np.nonzero(dg > 1e-5)[0]
    print("WFM stereo=%d: gpu-vs-o32 > 1e-5: count %d, last index %s, max %.3g; o32-vs-o64 > 1e-5: count %d last %s max %.3g"
          % (stereo, len(bad), bad[-1] if len(bad) else None, dg.max(), int((do > 1e-5).sum()),
             np.nonzero(do > 1e-5)[0][-1] if (do > 1e-5).any() else None, do.max()))
