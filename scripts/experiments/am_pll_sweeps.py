"""Model of the AM-Synch carrier loop (oracle/sdr_oracle.py CarrierPLL; rx.demod.am_pll, receiver.py:649) as
block fixed-point sweeps on a 32-bit phase accumulator -- what round 5 builds in stage2.hip (am_pll_sweep_*).

The detector e = atan2(Im v, Re v), v = y exp(-j theta), is wrap(arg y - theta): the loop is LINEAR in the phase
domain.  So (i) phi = arg y is computed once per sample, in parallel, off the chain; (ii) on a 2^32 phase word the
wrap is the integer overflow; (iii) a block of 64 samples is solved by sweeps: from a guess of the 64 phases every
lane computes e_j, the integrator is w0 + ki * inclusive_scan(e), the increment rint((w_j + kp e_j) * 2^32/2pi), the
phases theta0 + exclusive_scan(increment) -- sample 0 is exact from the start, sweep k makes samples 0..k exact.

Questions answered here (NumPy model of the same sweeps; build container only):
  1. how far is the integer-phase recursion from the oracle's float32 walk (audio, phase)?
  2. how many sweeps per block until a sweep reproduces its input bit for bit?  what does a cap of N leave?
  3. how long must a warm-up be from the guess theta = phi[start] for the join tolerance?

    python scripts/experiments/am_pll_sweeps.py
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

F = np.float32
R2W = F(2 ** 32 / (2 * math.pi))
W2R = F(2 * math.pi / 2 ** 32)


def loop_consts(fs=48000.0, bw=50.0, zeta=0.7071):
    wn = 2 * math.pi * bw / fs
    return F(2 * zeta * wn), F(wn * wn), fs / (zeta * wn * fs)


def serial_float(yr, yi, th, w, kp, ki):
    """the oracle's recursion (float32 walk)"""
    n = len(yr)
    TH = np.empty(n, F); VR = np.empty(n, F)
    pi, twopi = F(math.pi), F(2 * math.pi)
    for i in range(n):
        TH[i] = th
        c, s = F(np.cos(th)), F(np.sin(th))
        vr = yr[i] * c + yi[i] * s
        vi = yi[i] * c - yr[i] * s
        e = F(np.arctan2(vi, vr))
        w = F(w + ki * e)
        th = F(th + F(w + kp * e))
        if th >= pi: th = F(th - twopi)
        elif th < -pi: th = F(th + twopi)
        VR[i] = vr
    return TH, VR, th, w


def phase_words(yr, yi):
    return np.rint(np.arctan2(yi, yr).astype(F) * R2W).astype(np.int64).astype(np.uint32)


def serial_int(phi, ph, w, kp, ki):
    """the same loop on a phase word, sample by sample"""
    n = len(phi)
    PH = np.empty(n, np.uint32)
    ph = int(ph)
    for i in range(n):
        PH[i] = ph
        d = (int(phi[i]) - ph + 2 ** 31) % 2 ** 32 - 2 ** 31
        e = F(F(d) * W2R)
        w = F(w + ki * e)
        inc = int(np.rint(F(F(w + kp * e) * R2W)))
        ph = (ph + inc) % 2 ** 32
    return PH, ph, w


def sweep_block(phi, ph0, w0, kp, ki, cap, check=True, stats=None):
    """one block (<= 64 samples) by sweeps; returns phases in front of every sample, end state, sweeps used"""
    n = len(phi)
    j = np.arange(n, dtype=np.int64)
    inc0 = int(np.rint(F(w0 * R2W)))
    ph = (int(ph0) + j * inc0) % 2 ** 32
    it = 0
    while True:
        d = ((phi.astype(np.int64) - ph + 2 ** 31) % 2 ** 32 - 2 ** 31)
        e = (d.astype(F) * W2R).astype(F)
        S = np.cumsum(e, dtype=F)
        wj = (ki * S + w0).astype(F)
        corr = np.rint(((kp * e + wj).astype(F) * R2W).astype(F)).astype(np.int64)
        tot = np.cumsum(corr)
        new = (int(ph0) + tot - corr) % 2 ** 32
        it += 1
        same = np.array_equal(new, ph)
        ph = new
        if (check and same) or it >= cap:
            break
    if stats is not None:
        stats.append(it)
    return ph, (int(ph0) + int(tot[-1])) % 2 ** 32, F(wj[-1]), it


def sweep_walk(phi, ph0, w0, kp, ki, cap=66, check=True, stats=None):
    n = len(phi)
    PH = np.empty(n, np.uint32)
    for i0 in range(0, n, 64):
        p, ph0, w0, _ = sweep_block(phi[i0:i0 + 64], ph0, w0, kp, ki, cap, check, stats)
        PH[i0:i0 + 64] = p
    return PH, ph0, w0


def wdiff(a, b):
    return np.abs((a.astype(np.int64) - b.astype(np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31)


def signals(n, fs=48000.0):
    rng = np.random.default_rng(7)
    t = np.arange(n) / fs
    nz = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / math.sqrt(2)
    out = {}
    out["AM 50% + 7 Hz off, noise -40 dBc"] = 0.3 * (1 + 0.5 * np.sin(2 * np.pi * 1000 * t)) * np.exp(1j * (2 * np.pi * 7.0 * t + 0.7)) + 3e-3 * nz
    out["AM 90% + 40 Hz off, noise -20 dBc"] = 0.3 * (1 + 0.9 * np.sin(2 * np.pi * 400 * t)) * np.exp(1j * (2 * np.pi * 40.0 * t - 2.0)) + 3e-2 * nz
    out["noise only"] = 0.05 * nz
    return out


def main():
    kp, ki, tau = loop_consts()
    print("carrier loop: kp %.6g ki %.6g, tau = %.0f samples; 64 kp = %.2f" % (kp, ki, tau, 64 * kp))
    n = 120000
    for name, y in signals(n).items():
        yr, yi = y.real.astype(F), y.imag.astype(F)
        print("==", name)
        TH, VR, _, _ = serial_float(yr, yi, F(0), F(0), kp, ki)
        phi = phase_words(yr, yi)
        PHs, _, _ = serial_int(phi, 0, F(0), kp, ki)
        st = []
        PHw, _, _ = sweep_walk(phi, 0, F(0), kp, ki, stats=st)
        # 1. deviation from the oracle
        thw = np.rint(TH.astype(np.float64) * 2 ** 32 / (2 * math.pi)).astype(np.int64) % 2 ** 32
        lo = 8000
        d_or = wdiff(PHs[lo:], thw[lo:].astype(np.uint32))
        print("  integer serial vs float32 oracle walk: max %.3g rad (median %.3g)" % (d_or.max() * W2R, np.median(d_or) * W2R))
        rev = PHs.astype(np.int32).astype(F) * F(1.0 / 2 ** 32)
        c, s = np.cos(2 * np.pi * rev.astype(np.float64)).astype(F), np.sin(2 * np.pi * rev.astype(np.float64)).astype(F)
        vr = yr * c + yi * s
        print("  audio Re(v): max |diff| / max |v| = %.3g" % (np.max(np.abs(vr[lo:] - VR[lo:])) / np.max(np.abs(VR[lo:]))))
        d_sw = wdiff(PHw, PHs)
        print("  sweeps to the bit-stable fixed point vs integer serial: max %d words; sweeps per block mean %.2f max %d"
              % (d_sw.max(), np.mean(st), max(st)))
        # 2. caps
        for cap in (4, 5, 6, 7, 8):
            P2, _, _ = sweep_walk(phi, 0, F(0), kp, ki, cap=cap, check=False)
            dd = wdiff(P2[lo:], PHw[lo:])
            print("  cap %d: max %d words from the fixed-point walk (%.2g rad)" % (cap, dd.max(), dd.max() * W2R))
        # 3. warm-ups from theta = phi[start], w = 0 / the true w far away
        rng = np.random.default_rng(3)
        for Wt in (8, 10, 12, 14, 16):
            Wn = (int(math.ceil(Wt * tau)) + 63) & ~63
            er, ew = [], []
            for s0 in range(20000, n - 100, 4099):
                a = s0 - Wn
                _, ph, w = sweep_walk(phi[a:s0], int(phi[a]), F(0), kp, ki, cap=6, check=False)
                er.append(int(wdiff(np.array([ph], np.uint32), PHw[s0:s0 + 1])[0]))
            print("  warm-up %2d tau = %5d samples from theta = phi[start], w = 0: join off by max %d words = %.2g rad (median %d)"
                  % (Wt, Wn, max(er), max(er) * W2R, np.median(er)))


if __name__ == "__main__":
    main()
