"""Generate tests/golden/peaks_ref.npz by EXECUTING the reference's peak-pick statements on crafted lines (build container
only: /root/reference does not travel).

From /root/reference/Plotting.py as they stand, picked by line number: :587 `bkgnd = np.median(PSD2)`, :594
`dist = self.P.PEAK_DIST/self.psd.df` and :596 `peaks, _ = signal.find_peaks(PSD2,distance=dist,height=bkgnd+10)`.  The
lines (PSD2) are made here: noise with carriers, flat tops of 2 - 7 bins (SciPy takes the midpoint), flat stretches touching
either end of the line (no peaks), peaks in the first and last interior bins, EQUAL heights further apart than the distance
(both stay) and closer than it.  The last kind is where SciPy itself is not a function of its input: it ranks equal heights
with an unstable `np.argsort`, so which of two equal peaks closer than `dist` survives depends on NumPy's sort.  Those
cases are marked (`tie{k} = 1`); the device kernel's rule for them (the higher index outranks, as a stable sort would
give) is checked against the sequential greedy walk written out in tests/test_gpu_parity.py instead.  The fixture holds
lines, distances and SciPy's indices: data, none of the reference's text.

    python tests/golden/make_peaks_ref_golden.py
"""
import os
import textwrap
import types

import numpy as np
from scipy import signal

REF = "/root/reference/Plotting.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def lines_and_dists():
    rng = np.random.default_rng(29)
    out = []
    # 0: the kind of line the display sees -- noise floor, carriers of different heights
    x = (rng.standard_normal(4096) * 2 - 95).astype(np.float32)
    for i, h in ((300, 40), (310, 25), (1200, 33.5), (1203, 33.0), (2500, 50), (4000, 21)):
        x[i] += h
    out.append((x, 2.0, 0.125, 0))
    # 1: flat tops (2 .. 7 bins wide), all levels distinct
    x = (rng.standard_normal(1024) * 0.5 - 90).astype(np.float32)
    for j, w in enumerate((2, 3, 4, 5, 6, 7)):
        x[100 + 120 * j:100 + 120 * j + w] = -60.0 + j
    out.append((x, 1.0, 0.125, 0))
    # 2: flat stretches touching both ends, peaks in bins 1 and n - 2
    x = (rng.standard_normal(512) - 90).astype(np.float32)
    x[:5] = -50.0
    x[-7:] = -49.0
    x[1] = -40.0
    x[-2] = -41.0
    x[0] = -45.0
    out.append((x, 0.5, 0.125, 0))
    # 3: equal heights FURTHER apart than the distance: both stay
    x = (rng.standard_normal(2048) - 90).astype(np.float32)
    x[500] = x[540] = x[1000] = -55.0
    x[1500] = -55.0
    out.append((x, 4.0, 0.125, 0))
    # 4: a comb closer than the distance with strictly different heights: the greedy walk in full
    x = (rng.standard_normal(8192) - 90).astype(np.float32)
    hs = rng.permutation(400).astype(np.float32) * 0.05 - 70.0
    x[100:100 + 4 * 400:4] = hs
    out.append((x, 5.0, 0.125, 0))
    # 5: a descending staircase of peaks 3 bins apart (one kept peak per round of the kernel)
    x = np.full(4096, -90.0, np.float32) + rng.standard_normal(4096).astype(np.float32) * 0.01
    x[50:50 + 3 * 600:3] = -40.0 - 0.01 * np.arange(600, dtype=np.float32)
    out.append((x, 2.0, 0.125, 0))
    # 6: a 64k line (the RF PSD's size), noise only: ~20 k local maxima above the median + 10?  no: + 1 dB of headroom -> few
    x = (rng.standard_normal(65536) * 3 - 100).astype(np.float32)
    x[rng.integers(0, 65536, 200)] += 30
    out.append((x, 10.0, 8000.0 / 65536, 0))
    # 7, 8: EQUAL heights closer than the distance (SciPy's answer depends on np.argsort's handling of ties)
    x = (rng.standard_normal(1024) - 90).astype(np.float32)
    x[300] = x[304] = x[308] = -50.0
    out.append((x, 1.0, 0.125, 1))
    x = np.round(rng.standard_normal(4096) * 7).astype(np.float32) - 60.0
    out.append((x, 1.0, 0.125, 1))
    return out


def main():
    L = open(REF).read().splitlines()
    pick = lambda *nums: textwrap.dedent("\n".join(L[i - 1] for i in nums))
    text = pick(587) + "\n" + pick(594) + "\n" + pick(596)
    for frag in ("bkgnd = np.median(PSD2)", "dist = self.P.PEAK_DIST/self.psd.df", "signal.find_peaks(PSD2,distance=dist,height=bkgnd+10)"):
        assert frag in text, frag
    code = compile(text, "Plotting.py:587,594,596", "exec")
    out = {}
    cases = lines_and_dists()
    for k, (x, peak_dist, df, tie) in enumerate(cases):
        self = types.SimpleNamespace(P=types.SimpleNamespace(PEAK_DIST=peak_dist), psd=types.SimpleNamespace(df=df))
        env = dict(self=self, np=np, signal=signal, PSD2=x.astype(np.float64))
        exec(code, env)
        out[f"line{k}"] = x
        out[f"peak_dist{k}"] = peak_dist
        out[f"df{k}"] = df
        out[f"bk{k}"] = float(env["bkgnd"])
        out[f"peaks{k}"] = np.array(env["peaks"], np.int64)
        out[f"tie{k}"] = tie
        print(k, len(x), "dist", env["dist"], "peaks", len(env["peaks"]), "tie" if tie else "")
    out["ncases"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "peaks_ref.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "peaks_ref.npz")))


if __name__ == "__main__":
    main()
