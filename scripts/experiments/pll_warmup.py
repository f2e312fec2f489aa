"""How fast do the two PLLs forget their start state?  (the warm-up length of the
time-parallel kernels.)  Restates the two recursions in C float32 (gcc, compiled into a temp dir),
runs the SERIAL loop over a long record for the reference trajectory, then restarts it W samples
in front of many points from a random / an extrapolated state and reports what is left of the
difference at the point.  Build container only (imports the oracle for the broadcast-FM front end).

    python scripts/experiments/pll_warmup.py
Round-2 output (excerpt):
  pilot PLL (30 Hz at 250 kHz), random start phase:  W 16384 -> 1.0e6 words of 2^32, 24576 -> 14473, 32768 -> 99
                               extrapolated start :  W 16384 -> 63489, 24576 -> 256
  carrier PLL (50 Hz at 48 kHz), random start      :  W 2048 -> 2.2e-4 rad, 3072 -> 2.9e-6, 4096 -> 2.4e-7 (median 0: identical floats)
"""
import ctypes as C
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SRC = r'''
#include <math.h>
#include <stdint.h>
void wfm_pll(const float* m, int n, uint32_t* ph_io, float* w_io, float kp, float ki, float norm, float rad2word,
             uint32_t fword0, uint32_t* ph_out, float* w_out) {
  uint32_t ph = *ph_io; float w = *w_io;
  for (int i = 0; i < n; ++i) {
    if (ph_out) { ph_out[i] = ph; w_out[i] = w; }
    float rev = (float)(int32_t)ph * (1.0f / 4294967296.0f);
    float c = (float)cos(2.0 * M_PI * (double)rev);
    float e = (m[i] * c) * norm;
    w = w + ki * e;
    float t = (w + kp * e) * rad2word;
    ph = ph + fword0 + (uint32_t)(int32_t)rintf(t);
  }
  *ph_io = ph; *w_io = w;
}
void am_pll(const float* yr, const float* yi, int n, float* th_io, float* w_io, float kp, float ki,
            float* th_out, float* w_out) {
  float th = *th_io, w = *w_io;
  const float pi = (float)M_PI, twopi = (float)(2 * M_PI);
  for (int i = 0; i < n; ++i) {
    if (th_out) { th_out[i] = th; w_out[i] = w; }
    float c = (float)cos((double)th), s = (float)sin((double)th);
    float vr = yr[i] * c + yi[i] * s, vi = yi[i] * c - yr[i] * s;
    float e = (float)atan2((double)vi, (double)vr);
    w = w + ki * e;
    th = th + (w + kp * e);
    if (th >= pi) th -= twopi; else if (th < -pi) th += twopi;
  }
  *th_io = th; *w_io = w;
}
'''
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "pll.c"), "w").write(SRC)
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(tmp, "pll.c"), "-o",
                       os.path.join(tmp, "libpll.so"), "-lm"])
lib = C.CDLL(os.path.join(tmp, "libpll.so"))
fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
rng = np.random.default_rng(1)


def pilot():
    from oracle import wfm_oracle as wo
    from pysdr_amd.synth import synth_wfm
    fs, L, nch = 10e6, 213333, 80
    x = synth_wfm(fs, nch * L, 4)
    rx = wo.WfmReceiver(fs, 48e3, 300e3, stereo=False, ntaps_dec=255)
    mp, orig = [], rx.audio.process
    rx.audio.process = lambda w: (mp.append(np.asarray(w).real.astype(np.float32).copy()), orig(w))[1]
    for k in range(nch):
        rx.demod_data(x[k * L:(k + 1) * L])
    m = np.concatenate(mp)
    n, fs1 = len(m), 250e3
    wn = 2 * math.pi * 30 / fs1
    kp, ki = np.float32(2 * 0.7071 * wn), np.float32(wn * wn)
    norm, R = np.float32(20.0), np.float32(2 ** 32 / (2 * math.pi))
    fw0 = int(round(19000.0 / fs1 * 2 ** 32))

    def run(seg, ph, w, trace=False):
        seg = np.ascontiguousarray(seg, np.float32)
        phc, wc = C.c_uint32(ph), C.c_float(w)
        po = np.empty(len(seg), np.uint32) if trace else None
        wo_ = np.empty(len(seg), np.float32) if trace else None
        lib.wfm_pll(seg.ctypes.data_as(fp), len(seg), C.byref(phc), C.byref(wc), C.c_float(kp), C.c_float(ki), C.c_float(norm),
                    C.c_float(R), C.c_uint32(fw0), po.ctypes.data_as(up) if trace else None,
                    wo_.ctypes.data_as(fp) if trace else None)
        return phc.value, wc.value, po, wo_
    _, _, P, Wt = run(m, 0, 0.0, True)
    g0 = []
    print("pilot PLL, %d IF samples; tau = %.0f samples" % (n, fs1 / (0.7071 * 2 * math.pi * 30)))
    # the mean increment of a past stretch (exact: unwrapped phase advance / samples) as the predictor
    base, span = 40000, 60000
    adv = np.cumsum(((P[base + 1:base + span + 1].astype(np.int64) - P[base:base + span].astype(np.int64)) % 2 ** 32))
    slope = float(adv[-1]) / span
    wmean = float(np.mean(Wt[base:base + span]))
    for W in (4096, 16384, 20480, 22528, 24576, 32768):
        er, ee, es, ws = [], [], [], []
        for s in range(60000, n - 1000, 7919):
            ph, _, _, _ = run(m[s - W:s], int(rng.integers(0, 2 ** 32)), float(Wt[0]))
            er.append(abs((int(ph) - int(P[s]) + 2 ** 31) % 2 ** 32 - 2 ** 31))
            base = 50000
            inc = fw0 + int(np.rint(np.float32(Wt[base]) * R))
            ph, _, _, _ = run(m[s - W:s], (int(P[base]) + (s - W - base) * inc) % 2 ** 32, float(Wt[base]))
            ee.append(abs((int(ph) - int(P[s]) + 2 ** 31) % 2 ** 32 - 2 ** 31))
            g = (int(P[base]) + int(round((s - W - base) * slope))) % 2 ** 32
            if W == 4096:
                g0.append(abs((g - int(P[s - W]) + 2 ** 31) % 2 ** 32 - 2 ** 31))
            ph, wv, _, _ = run(m[s - W:s], g, wmean)
            es.append(abs((int(ph) - int(P[s]) + 2 ** 31) % 2 ** 32 - 2 ** 31))
            ws.append(abs(float(wv) - float(Wt[s])))
        if W == 4096:
            print("  mean-increment guess itself is off by max %d words (median %d)" % (max(g0), np.median(g0)))
        print("  W %5d: random start -> max %d words (median %d); extrapolated start -> max %d; mean-increment start -> max %d (median %d), integrator off by max %.2g"
              % (W, max(er), np.median(er), max(ee), max(es), np.median(es), max(ws)))


def carrier():
    fs, n = 48000.0, 400000
    t = np.arange(n) / fs
    y = 0.3 * (1 + 0.5 * np.sin(2 * np.pi * 1000 * t)) * np.exp(1j * (2 * np.pi * 3.0 * t + 0.7)) \
        + 2e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / math.sqrt(2)
    yr, yi = np.ascontiguousarray(y.real, np.float32), np.ascontiguousarray(y.imag, np.float32)
    wn = 2 * math.pi * 50 / fs
    kp, ki = np.float32(2 * 0.7071 * wn), np.float32(wn * wn)

    def run(a, b, th, w, trace=False):
        thc, wc = C.c_float(th), C.c_float(w)
        to = np.empty(b - a, np.float32) if trace else None
        wo_ = np.empty(b - a, np.float32) if trace else None
        lib.am_pll(yr[a:b].ctypes.data_as(fp), yi[a:b].ctypes.data_as(fp), b - a, C.byref(thc), C.byref(wc), C.c_float(kp),
                   C.c_float(ki), to.ctypes.data_as(fp) if trace else None, wo_.ctypes.data_as(fp) if trace else None)
        return thc.value, wc.value, to, wo_
    _, _, TH, WW = run(0, n, 0.0, 0.0, True)
    print("carrier PLL; tau = %.0f samples" % (fs / (0.7071 * 2 * math.pi * 50)))
    for W in (1024, 2048, 3072, 4096):
        er = []
        for s in range(20000, n - 10, 3571):
            th, _, _, _ = run(s - W, s, float(rng.uniform(-math.pi, math.pi)), 0.0)
            er.append(abs((th - TH[s] + math.pi) % (2 * math.pi) - math.pi))
        print("  W %5d: random start -> max %.3g rad (median %.3g)" % (W, max(er), np.median(er)))


if __name__ == "__main__":
    carrier()
    pilot()
