"""Latency of ONE live call -- rx.demod_data(x) on a host chunk, results back in host arrays (receiver.py:235) -- per mode,
at the am.py rate with the default 1001-tap prototype.  A live call is always the single-stream form and, for the serial
loops, one segment from the true state.   python scripts/diag/live_latency.py [tree]   (tree: another checkout to import from)"""
import os
import sys
import time

import numpy as np

root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from pysdr_amd import sig_proc  # noqa: E402
from pysdr_amd.params import RunTimeParams  # noqa: E402
from pysdr_amd.synth import CONFIGS, synth_iq  # noqa: E402

cfg = CONFIGS['C1']
for mode in ('AM', 'AM-Synch', 'NFM', 'USB'):
    P = RunTimeParams(fs=cfg['fs'], fsout=48e3, fc=[7e6], mode=mode, nfilt=1001)
    g = sig_proc.Receiver(P, 100e3, 0, '1')
    g.mode, g.af_bw = mode, 5e3
    L = P.IN_CHUNK_SIZE
    x = synth_iq(cfg, 8 * L, 3)
    for k in range(16):
        g.demod_data(x[(k % 8) * L:(k % 8 + 1) * L])
    ts = []
    for k in range(200):
        xc = x[(k % 8) * L:(k % 8 + 1) * L]
        t0 = time.perf_counter()
        g.demod_data(xc)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"{os.path.basename(root):12s} {mode:9s} one {L}-sample chunk: median {np.median(ts):7.1f} us  min {ts.min():7.1f}  p90 {np.percentile(ts, 90):7.1f}   (real time: {L / cfg['fs'] * 1e6:.0f} us per chunk)", flush=True)
    P._pysdr_stream.close()
