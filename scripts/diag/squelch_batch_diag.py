"""Where a batch with the squelch armed departs from the chunk-by-chunk loop (tiny chunks): first differing output, its
block, the gate history.   python scripts/diag/squelch_batch_diag.py L B thresh"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from oracle import sdr_oracle as so
from test_gpu_parity import make_gpu_receivers

L, B, th = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
cfg = so.CONFIGS['C2']
x = so.synth_iq(cfg, B * L, 14)
noise_only = so.synth_iq(dict(cfg, carriers=[]), B * L, 15)
a, b = (B // 3) * L, (B // 3 + B // 4) * L
x[a:b] = noise_only[a:b]
P1, g1 = make_gpu_receivers(cfg)
if th > 0: g1[0].squelch = th
am1, gate1, lvl1, iq1, gain1 = [], [], [], [], []
for k in range(B):
    am1.append(g1[0].demod_data(x[k * L:(k + 1) * L]).copy())
    iq1.append(g1[0].iq.copy()); gain1.append(g1[0].agc.gain)
    st = g1[0].squelch_state
    gate1.append(st[1]); lvl1.append(st[0])
P2, g2 = make_gpu_receivers(cfg, max_batch_chunks=B)
if th > 0: g2[0].squelch = th
ctx = P2._pysdr_stream
ctx.process_batch(x, B, L, on_device=False)
am, iq, cn, pk = ctx.fetch(0, B)
ref = np.concatenate(am1)
print("counts equal", list(cn) == [len(v) for v in am1], "final state", g2[0].squelch_state, g1[0].squelch_state)
d = np.nonzero(am.view(np.uint32) != ref.view(np.uint32))[0]
print("differing outputs", len(d), "of", len(ref))
iqr = np.concatenate(iq1)
print("baseband IQ equal:", np.array_equal(iq.view(np.uint32), iqr.view(np.uint32)), "differing", int(np.count_nonzero(iq != iqr)), "of", len(iqr))
if len(d):
    edges = np.cumsum([0] + [len(v) for v in am1])
    blocks = sorted(set(int(np.searchsorted(edges, i, side='right') - 1) for i in d))
    print("blocks", blocks[:40], "...", len(blocks))
    for i in d[:8]:
        k = int(np.searchsorted(edges, i, side='right') - 1)
        print("  out", int(i), "block", k, "batch", am[i], "chunked", ref[i], "gate", gate1[k], "lvl", lvl1[k], "prev gate", gate1[k - 1] if k else None)
