"""AM-Synch carrier loop: the start state of a segment by ONE LINEAR SOLVE over the warm-up window instead of walking it.

The loop is linear in the phase domain as long as the detector does not wrap.  Around a straight line
g[j] = a + j G over the window (a = block mean of the signal's own phase at the window start, G = the integrator at the
start of the call, in words of 2^32 per sample), with u[j] = wrap(phi[j] - g[j]) taken per sample (no continuity of phi
needed: noise wraps exactly as in the loop), eps = theta - g, V = W - G:
    eps[j+1] = (1 - kp - ki) eps[j] + V[j-1] + (kp + ki) u[j]
    V[j]     = V[j-1] - ki eps[j] + ki u[j]
-- a constant 2x2 matrix A and an input: x[N] = sum_k A^(N-1-k) b[k] (+ A^N x[0], forgotten: 16 tau).  On the device a
lane takes the samples 64 j + l of the window (coalesced), Horner in A^64, then A^(63-l) per lane and a wave sum.

Question: how far is (a + N G + eps[N], (G + V[N-1]) / R) from the state the exact walk has at that sample?

    python scripts/experiments/am_linear_seed.py
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from am_pll_sweeps import F, R2W, W2R, loop_consts, phase_words, signals, sweep_walk, wdiff  # noqa: E402


def wrap32(v):
    return (np.asarray(v, np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31


def linear_seed(phi, wb, s0, w_call, kp, ki, lanes=True):
    """state in front of sample s0 from the window [wb, s0)"""
    G = int(np.rint(F(w_call * R2W)))
    f0 = int(phi[wb])
    dev = wrap32(phi[wb:wb + 64].astype(np.int64) - f0 - np.arange(64) * G).astype(F)
    a = (f0 + int(np.rint(np.sum(dev, dtype=F) / F(64)))) % 2 ** 32
    N = s0 - wb
    j = np.arange(N, dtype=np.int64)
    u = wrap32(phi[wb:s0].astype(np.int64) - a - j * G).astype(np.float64)
    kpd, kid = float(kp), float(ki)
    # in words: the integrator's gain acts on e in words and yields words per sample
    A = np.array([[1.0 - kpd - kid, 1.0], [-kid, 1.0]])
    b = np.stack([(kpd + kid) * u, kid * u])
    if not lanes:
        x = np.zeros(2)
        for k in range(N):
            x = A @ x + b[:, k]
    else:
        # the device's order: lane l owns samples 64 j + l; Horner in A^64; A^(63 - l); sum over lanes
        A64 = np.linalg.matrix_power(A, 64)
        m = N // 64
        c = np.zeros((2, 64))
        for jj in range(m):
            c = A64 @ c + b[:, 64 * jj:64 * jj + 64]
        x = np.zeros(2)
        for l in range(64):
            x += np.linalg.matrix_power(A, 63 - l) @ c[:, l]
    ph = (a + N * G + int(np.rint(x[0]))) % 2 ** 32
    # V = W - frac: the integrator in words per sample is G + V
    w = F((G + x[1]) / float(R2W))
    return ph, w, float(np.max(np.abs(u))) / 2 ** 32


def main():
    kp, ki, tau = loop_consts()
    n = 160000
    for name, y in signals(n).items():
        yr, yi = y.real.astype(F), y.imag.astype(F)
        phi = phase_words(yr, yi)
        # the exact walk, with its integrator in front of every block of 64
        PH = np.empty(n, np.uint32); Wb = {}
        ph0, w0 = 0, F(0)
        for i0 in range(0, n, 64):
            Wb[i0] = (ph0, w0)
            p, ph0, w0 = sweep_walk(phi[i0:i0 + 64], ph0, w0, kp, ki)
            PH[i0:i0 + 64] = p
        print("==", name)
        w_call = Wb[8000 // 64 * 64][1]          # "the integrator as the call began": early, barely settled
        for Wt in (6, 8, 10, 12, 16):
            Wn = (int(math.ceil(Wt * tau)) + 63) & ~63
            er, ew, um = [], [], []
            for s0 in range(20032, n - 100, 4096):
                ph, w, umax = linear_seed(phi, s0 - Wn, s0, w_call, kp, ki)
                tp, tw = Wb[s0]
                er.append(int(wdiff(np.array([ph], np.uint32), np.array([tp], np.uint32))[0]))
                ew.append(abs(float(w) - float(tw)))
                um.append(umax)
            print("  window %2d tau = %5d samples: phase off by max %d words (median %d), integrator max %.2g rad/sample; max |u| %.3f rev"
                  % (Wt, Wn, max(er), np.median(er), max(ew), max(um)))
        # the two orders agree
        s0 = 20032 + 4096
        Wn = (int(math.ceil(16 * tau)) + 63) & ~63
        p1 = linear_seed(phi, s0 - Wn, s0, w_call, kp, ki, lanes=True)
        p2 = linear_seed(phi, s0 - Wn, s0, w_call, kp, ki, lanes=False)
        print("  lane order vs sample order: %d words, %.2g" % (wdiff(np.array([p1[0]], np.uint32), np.array([p2[0]], np.uint32))[0], abs(float(p1[1]) - float(p2[1]))))


if __name__ == "__main__":
    main()
