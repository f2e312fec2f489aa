"""Adversarial sweep: a batch of chunks whose output counts are odd / ragged against the chunk-by-chunk loop, bit for bit, for
the modes and front ends the identity tests do not cover (IQ / LSB / RTTY on the matrix-core front end, broadcast FM mono).
    python scripts/diag/odd_counts_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import sdr_oracle as so
from test_gpu_parity import make_gpu_receivers


def case(name, cfg, L, B, cuts):
    x = so.synth_iq(cfg, B * L, 21)
    P1, g1 = make_gpu_receivers(cfg)
    am1, iq1 = [[] for _ in g1], [[] for _ in g1]
    for k in range(B):
        for i, rx in enumerate(g1):
            am1[i].append(np.array(rx.demod_data(x[k * L:(k + 1) * L])).copy()); iq1[i].append(rx.iq.copy())
    odd = sum(len(a) & 1 for a in am1[0])
    P2, g2 = make_gpu_receivers(cfg, max_batch_chunks=B)
    ctx = P2._pysdr_stream
    bad = []
    lo = 0
    for hi in cuts + [B]:
        ctx.process_batch(x[lo * L:hi * L], hi - lo, L, on_device=False)
        for i in range(len(g2)):
            am, iq, cn, pk = ctx.fetch(i, hi - lo)
            ra, ri = np.concatenate(am1[i][lo:hi]), np.concatenate(iq1[i][lo:hi])
            if list(cn) != [len(a) for a in am1[i][lo:hi]]: bad.append((i, lo, 'counts'))
            elif not np.array_equal(iq.view(np.uint32), ri.view(np.uint32)): bad.append((i, lo, 'iq', int(np.count_nonzero(iq != ri))))
            elif am.dtype != ra.dtype or not np.array_equal(am.view(np.uint32), ra.view(np.uint32)): bad.append((i, lo, 'audio', int(np.count_nonzero(am != ra)), len(ra)))
        lo = hi
    ctx.close()
    print('%-44s L %6d B %4d odd-count chunks %4d: %s' % (name, L, B, odd, 'OK' if not bad else bad))


base = so.CONFIGS['C1']
for mode, af in (('IQ', 10e3), ('LSB', 3e3), ('RTTY', 3e3), ('AM', 5e3), ('CW', 1e3)):
    cfg = dict(base, rx=[dict(frq=100e3, mode=mode, video_bw=20e3, af_bw=af, bfo=700.0 if mode == 'CW' else 0.0)])
    for L in (43690 // 16, 999):
        case('2.048 MS/s 1001 taps (matrix cores) ' + mode, cfg, L, 120, [37, 80])
c2 = so.CONFIGS['C2']
for mode, af in (('IQ', 10e3), ('USB', 3e3)):
    cfg = dict(c2, rx=[dict(frq=-1.1e6, mode=mode, video_bw=20e3, af_bw=af)])
    case('8 MS/s 255 taps ' + mode, cfg, 1500, 150, [51])
# six and eight receivers on one stream (the tasks of the vector kernel are dealt out by (branch, RX half) above four)
modes = [('USB', 3e3), ('CW', 1e3), ('NFM', 0.0), ('AM', 5e3), ('LSB', 3e3), ('IQ', 10e3), ('AM', 5e3), ('NFM', 0.0)]
for n in (2, 3, 5, 6, 7, 8):
    cfg = dict(c2, rx=[dict(frq=-1.5e6 + 0.4e6 * i, mode=m, video_bw=20e3, af_bw=af, bfo=700.0 if m == 'CW' else 0.0) for i, (m, af) in enumerate(modes[:n])])
    case('8 MS/s 255 taps, %d RX' % n, cfg, 2500, 60, [23])

# broadcast FM mono (no pilot loop: every stage is a fixed-order sum): ragged chunk lengths, odd IF / audio counts
from oracle import wfm_oracle as wo
from pysdr_amd import sig_proc
from pysdr_amd.params import RunTimeParams
for L, B, cuts in ((20001, 60, [17, 40]), (3333, 200, [77])):
    x = wo.synth_wfm(10e6, B * L, 4)
    P1 = RunTimeParams(fs=10e6, fc=[98.1e6], mode='WFM', nfilt=255, foffset=300e3, vid_bw=200e3)
    g1 = sig_proc.Receiver(P1, 300e3, 0, '1')
    am1 = [np.array(g1.demod_data(x[k * L:(k + 1) * L])).copy() for k in range(B)]
    P2 = RunTimeParams(fs=10e6, fc=[98.1e6], mode='WFM', nfilt=255, foffset=300e3, vid_bw=200e3, max_batch_chunks=B)
    g2 = sig_proc.Receiver(P2, 300e3, 0, '1')
    ctx = P2._pysdr_stream
    bad, lo = [], 0
    for hi in cuts + [B]:
        ctx.process_batch(x[lo * L:hi * L], hi - lo, L, on_device=False)
        am, iq, cn, pk = ctx.fetch(0, hi - lo, want_iq=False)
        ra = np.concatenate(am1[lo:hi])
        if list(cn) != [len(a) for a in am1[lo:hi]]: bad.append((lo, 'counts'))
        elif not np.array_equal(am.view(np.uint32), ra.view(np.uint32)): bad.append((lo, 'audio', int(np.count_nonzero(am != ra)), len(ra)))
        lo = hi
    print('%-44s L %6d B %4d odd-count chunks %4d: %s' % ('10 MS/s broadcast FM mono', L, B, sum(len(a) & 1 for a in am1), 'OK' if not bad else bad))
