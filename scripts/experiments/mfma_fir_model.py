#!/usr/bin/env python3
"""Index-exact NumPy model of the shifted-tap (Toeplitz) MFMA form of the mix + decimate kernel
(pysdr_amd/csrc/mixdec_mfma.hip), written BEFORE the kernel to shake out the index arithmetic on the CPU
(there is no GPU in the authoring container).

Formulation ("Next round" 1).  Outputs m = UP*(S*G + t) + c  (G = super-group, t < S shift,
c < UP position in the group of UP outputs that share DOWN inputs) use the newest sample
    n_m = P*G + t*DOWN + off_c,   off_c = floor(c*DOWN/UP),   P = S*DOWN,   branch p_c = (c*DOWN) mod UP
    y[m] = sum_k g[p_c][k] * x[n_m - k],  k < KT
One MFMA row = one super-group G: its window is the K' = KT + (S-1)*DOWN + off_{UP-1} samples that start at
L(G) = P*G - (KT-1); window index j = n - L(G).  Column (t, c, part) of the 16-wide B operand holds branch
p_c's taps displaced to j = KT-1 + t*DOWN + off_c - k and zero elsewhere:
    chain 1:  A = Re x,  B1[j][2*(t*UP+c)+0] = Re g,  B1[j][..+1] = Im g
    chain 2:  A = Im x,  B2[j][2*(t*UP+c)+0] = -Im g, B2[j][..+1] = Re g
so that acc[row][2*(t*UP+c)] = Re y, [..+1] = Im y with no cross-lane step.  v_mfma_f32_16x16x4_f32 lane map
(guide): lane l holds A[row l&15][k l>>4], B[k l>>4][col l&15], C/D col l&15, rows 4*(l>>4)+reg.

LDS image of a tile (NB row blocks of 16 rows): the samples from  origin = L(G_tile) - d  (d in {0,1} makes
the origin even relative to the call's first sample, so that every 16-byte DMA element is an aligned pair) in
SEGMENTS of P samples, each followed by one 16-byte pad: rows are P samples apart = a multiple of the 256
bytes the LDS serves per clock, the pad moves consecutive rows 16 bytes further -> the 16 rows of a k-step
(the 16 lanes the LDS serves together) sit on 16 different bank quads.
    byte address of window sample j of row i = i*SEGB + (d+j)*8 + 16*((d+j) // P),  SEGB = 8P+16
    DMA slot q (16 bytes) of the image <-> seg = q // (P/2+1), w = q % (P/2+1); w == P/2 is the pad,
    otherwise the pair of samples origin + seg*P + 2w, +1.

The K' steps are cut into WK contiguous slices (one per wave of a row block); a wave sums its slice into two
accumulators (chain 1 / chain 2), adds them, and the WK partial tiles are added in slice order.  The model
does exactly that in float32 (k-ordered fma chains are modelled as sequential float32 multiply-adds, which
is not bit-exact fma but the same order) and compares with the direct polyphase sum in float64.
"""
import numpy as np


def taps_c(h, up, fword_rev, kpad):
    """g[p][k] = h[p+up*k] * exp(-j*2pi*frev*k) (api.hip build_taps)"""
    g = np.zeros((up, kpad), np.complex128)
    for p in range(up):
        for k in range(kpad):
            j = p + up * k
            if j < len(h):
                g[p, k] = h[j] * np.exp(-2j * np.pi * fword_rev * k)
    return g


class Geo:
    def __init__(self, UP, DOWN, S, KT, NB, WK):
        self.UP, self.DOWN, self.S, self.KT, self.NB, self.WK = UP, DOWN, S, KT, NB, WK
        self.P = S * DOWN
        assert self.P % 4 == 0 and 2 * S * UP <= 16
        self.off = [(c * DOWN) // UP for c in range(UP)]
        self.br = [(c * DOWN) % UP for c in range(UP)]
        self.KP = KT + (S - 1) * DOWN + self.off[UP - 1]          # K'
        self.nsteps = (self.KP + 3) // 4
        self.spw = (self.nsteps + WK - 1) // WK                   # steps of the longest slice
        base, extra = divmod(self.nsteps, WK)                     # the first `extra` slices are one step longer
        self.slice_first = [q * base + min(q, extra) for q in range(WK)]
        self.slice_steps = [base + (1 if q < extra else 0) for q in range(WK)]
        self.SEGB = 8 * self.P + 16
        self.rows = 16 * NB
        self.tile_samples = self.rows * self.P                    # samples a tile advances by
        self.img_samples = (self.rows - 1) * self.P + 4 * self.nsteps + 1   # d + j max + 1
        nseg_full, tail = divmod(self.img_samples, self.P)
        self.img_slots = nseg_full * (self.P // 2 + 1) + (tail + 1) // 2
        self.img_bytes = self.img_slots * 16

    def bmat(self, g):
        """B1, B2 [4*spw*WK][16] float32"""
        n = 4 * self.nsteps
        B1 = np.zeros((n, 16), np.float32)
        B2 = np.zeros((n, 16), np.float32)
        for t in range(self.S):
            for c in range(self.UP):
                col = 2 * (t * self.UP + c)
                for k in range(self.KT):
                    j = self.KT - 1 + t * self.DOWN + self.off[c] - k
                    gg = g[self.br[c], k]
                    B1[j, col] = gg.real
                    B1[j, col + 1] = gg.imag
                    B2[j, col] = -gg.imag
                    B2[j, col + 1] = gg.real
        return B1, B2


def run_call(geo, g, x_abs_fn, S0, n, hist_len, lds_model=True):
    """One call over absolute samples [S0, S0+n).  x_abs_fn(a) -> complex sample at absolute index a (for a >=
    S0 - hist_len).  Returns (m0, y[n_out]) computed the kernel's way."""
    UP, DOWN, S, KT, P = geo.UP, geo.DOWN, geo.S, geo.KT, geo.P
    S1 = S0 + n
    m0 = (S0 * UP + DOWN - 1) // DOWN
    m1 = (S1 * UP + DOWN - 1) // DOWN
    n_out = m1 - m0
    y = np.zeros(n_out, np.complex64)
    B1, B2 = geo.bmat(g)
    US = UP * S
    G_first = m0 // US
    # d: origin even relative to S0
    L0 = P * G_first - (KT - 1)
    d = (L0 - S0) % 2
    c0 = KT - 1 + d
    B0 = P * G_first - c0                      # absolute origin of tile 0's image
    assert B0 <= S0, "ownership of the first samples"
    ntiles = max(1, -(-(S1 - B0) // geo.tile_samples))
    owned = np.zeros(n, np.int32)
    for tau in range(ntiles):
        Gt = G_first + tau * geo.rows
        origin = P * Gt - c0                   # absolute sample of image slot 0
        assert (origin - S0) % 2 == 0
        # ---- DMA: image slots
        img = np.zeros(geo.img_bytes // 8, np.complex64)      # 8-byte units; stale = 0 here
        img[:] = np.nan if False else 0
        for q in range(geo.img_slots):
            seg, w = divmod(q, P // 2 + 1)
            if w == P // 2:
                continue
            a = origin + seg * P + 2 * w
            rel = a - S0
            if rel >= -hist_len and rel + 1 < n:
                img[2 * q] = x_abs_fn(a)
                img[2 * q + 1] = x_abs_fn(a + 1)
            elif rel >= -hist_len and rel < n:   # the odd last sample of the call
                img[2 * q] = x_abs_fn(a)
            elif rel + 1 >= -hist_len and rel + 1 < n:
                raise AssertionError("pair straddles the start of the history")
        # ownership (raw peak): the full segments of the image
        lo = max(origin - S0, 0)
        hi = min(origin + geo.tile_samples - S0, n)
        if tau == ntiles - 1:
            assert hi == n, (hi, n)
        if hi > lo:
            owned[lo:hi] += 1
        # ---- MFMA
        for b in range(geo.NB):
            part = np.zeros((geo.WK, 16, 16), np.float32)       # [slice][row][col]
            for q in range(geo.WK):
                acc1 = np.zeros((16, 16), np.float32)
                acc2 = np.zeros((16, 16), np.float32)
                for ls in range(geo.slice_steps[q]):
                    j0 = 4 * (geo.slice_first[q] + ls)
                    for kk in range(4):
                        j = j0 + kk
                        A = np.zeros(16, np.complex64)
                        for i in range(16):
                            byte = (b * 16 + i) * geo.SEGB + (d + j) * 8 + 16 * ((d + j) // P)
                            assert byte % 8 == 0 and byte + 8 <= geo.img_bytes, (byte, geo.img_bytes)
                            A[i] = img[byte // 8]
                        acc1 += np.outer(A.real.astype(np.float32), B1[j])
                        acc2 += np.outer(A.imag.astype(np.float32), B2[j])
                part[q] = acc1 + acc2
            tot = part[0].copy()
            for q in range(1, geo.WK):
                tot = tot + part[q]
            for i in range(16):
                G = Gt + b * 16 + i
                for t in range(S):
                    for c in range(UP):
                        m = UP * (S * G + t) + c
                        if m0 <= m < m1:
                            col = 2 * (t * UP + c)
                            y[m - m0] = tot[i, col] + 1j * tot[i, col + 1]
    assert np.all(owned == 1), "every sample owned by exactly one tile"
    return m0, y


def direct(g, UP, DOWN, KT, xfn, m):
    n_m = (m * DOWN) // UP
    p = (m * DOWN) % UP
    acc = 0j
    for k in range(KT):
        acc += g[p, k] * xfn(n_m - k)
    return acc


def check(UP, DOWN, S, KT, NB, WK, ntaps, cuts, seed=0):
    rng = np.random.default_rng(seed)
    geo = Geo(UP, DOWN, S, KT, NB, WK)
    h = rng.standard_normal(ntaps) / ntaps
    kpad = (KT + 15) // 16 * 16
    g = taps_c(h, UP, 0.0123, kpad).astype(np.complex64).astype(np.complex128)
    total = sum(cuts)
    hist_len = kpad + 2
    xs = (rng.standard_normal(total + hist_len) + 1j * rng.standard_normal(total + hist_len)).astype(np.complex64)
    xs[:hist_len] = 0

    def xfn(a):      # absolute index a >= -hist_len
        return xs[a + hist_len] if a + hist_len >= 0 else 0

    S0 = 0
    worst = 0.0
    nout = 0
    for n in cuts:
        m0, y = run_call(geo, g, xfn, S0, n, hist_len)
        for i in range(0, len(y), max(1, len(y) // 40)):
            ref = direct(g, UP, DOWN, KT, xfn, m0 + i)
            worst = max(worst, abs(y[i] - ref) / (abs(ref) + 1e-3))
        nout += len(y)
        S0 += n
    print(f"UP {UP} DOWN {DOWN} S {S} KT {KT} NB {NB} WK {WK}: K' {geo.KP} steps {geo.nsteps} (x{geo.spw} per wave), "
          f"image {geo.img_bytes} B, useful MACs {S*UP*KT*4/(32*4*geo.nsteps):.3f}, outputs {nout}, worst rel {worst:.2e}")
    assert worst < 1e-5


if __name__ == "__main__":
    # C1: 2.048 MS/s, 1001 taps -> 334 per branch; odd cuts exercise d = 1
    check(3, 128, 2, 334, 1, 8, 1001, [43690, 4097, 12345, 1, 2, 9000])
    check(3, 128, 2, 334, 2, 4, 1001, [43690, 4097, 12345])
    # C4 IF decimator: 10 MS/s / 40, 255 taps
    check(1, 40, 8, 255, 1, 8, 255, [21333, 21333, 7, 15000])
