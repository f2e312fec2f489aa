import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import sdr_oracle as so
from pysdr_amd import executive, stream, sig_proc
from tests.test_executive_host import make_P
cfg = so.CONFIGS['C1']
P = make_P(cfg, 4)
L = P.IN_CHUNK_SIZE
P.sdr = stream.SynthSDR(cfg, seed=31, nsamp=5 * L)
ex = executive.SDR_EXECUTIVE(P, dsp=None)
for i, r in enumerate(cfg['rx']):
    P.rx[i].mode, P.rx[i].af_bw, P.rx[i].bfo = r['mode'], r.get('af_bw'), r.get('bfo', 0.0)
def show(e):
    rx = P.rx[0]
    st = rx.agc
    print("chunk", P.nchunks, "x", np.abs(e.x).max(), "iq", np.abs(rx.iq).max() if len(rx.iq) else None, "am", np.abs(rx.am).max(), len(rx.am),
          "agc", st.agc, st.gain, st.maxbuf, "peak_in", rx.peak_in, "muted", P.MUTED[0], P.AUTO_MUTED)
ex.Run(on_chunk=show)
out = P.players[0].rb.pull(P.players[0].rb.nsamps)
print("player", np.abs(out).max(), len(out))
