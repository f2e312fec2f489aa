"""NumPy model of the single-workgroup 64k PSD (scripts/experiments/psd_one.hip): the index algebra of
every pass, checked against np.fft before any of it is written in HIP.

frame xw[0..32768) (windowed), zero-padded to N = 65536.  Bins 4k + r (r = 0..3) are the 16384-point
DFT of   y_r[n] = W_N^(r n) (xw[n] + (-j)^r xw[n + 16384]),  n < 16384          (decimation in frequency)
A thread t < 1024 holds n = t + 1024 i, i < 16.  The 16384-point DFT of y (k = output index):
  pass A  (registers)  over i:            Z[ka][t]       = sum_i y[t + 1024 i] W_16^(i ka),  * W_16384^(t ka)
  pass B  (LDS)        t = tl + 64 th:    over th:   U[ka][kb][tl] = sum_th Z[ka][tl + 64 th] W_16^(th kb), * W_1024^(tl kb)
  pass C  (LDS)        tl = u + 4 s:      over s:    V[ka][kb][kc][u] = sum_s U[ka][kb][u + 4 s] W_16^(s kc), * W_64^(u kc)
  pass D  (LDS)                           over u:    X[k] = sum_u V[ka][kb][kc][u] W_4^(u kd)
  with k = ka + 16 (kb + 16 (kc + 16 kd))   ... derived below and asserted numerically.
"""
import numpy as np

N, M, Q = 65536, 32768, 16384
rng = np.random.default_rng(1)
xw = (rng.standard_normal(M) + 1j * rng.standard_normal(M))
ref = np.fft.fft(np.concatenate((xw, np.zeros(M))))

W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)
out = np.empty(N, complex)
t = np.arange(1024)
for r in range(4):
    n = np.arange(Q)
    y = W(N, r * n) * (xw[:Q] + (-1j) ** r * xw[Q:])
    # pass A: y[t + 1024 i] -> Z[ka][t]
    Y = y.reshape(16, 1024)                                     # [i][t]
    Z = np.einsum('it,ik->kt', Y, W(16, np.outer(np.arange(16), np.arange(16))))      # [ka][t]
    Z = Z * W(Q, np.outer(np.arange(16), t))                    # W_Q^(t ka)
    # pass B: t = tl + 64 th
    Zr = Z.reshape(16, 16, 64)                                  # [ka][th][tl]
    U = np.einsum('ahl,hb->abl', Zr, W(16, np.outer(np.arange(16), np.arange(16))))   # [ka][kb][tl]
    U = U * W(1024, np.outer(np.arange(16), np.arange(64)))[None, :, :]               # W_1024^(tl kb)
    # pass C: tl = u + 4 s
    Ur = U.reshape(16, 16, 16, 4)                               # [ka][kb][s][u]
    V = np.einsum('absu,sc->abcu', Ur, W(16, np.outer(np.arange(16), np.arange(16)))) # [ka][kb][kc][u]
    V = V * W(64, np.outer(np.arange(16), np.arange(4)))[None, None, :, :]            # W_64^(u kc)
    # pass D: over u
    X4 = np.einsum('abcu,ud->abcd', V, W(4, np.outer(np.arange(4), np.arange(4))))    # [ka][kb][kc][kd]
    ka, kb, kc, kd = np.meshgrid(np.arange(16), np.arange(16), np.arange(16), np.arange(4), indexing='ij')
    k = ka + 16 * (kb + 16 * (kc + 16 * kd))
    Xr = np.empty(Q, complex)
    Xr[k.ravel()] = X4.ravel()
    out[4 * np.arange(Q) + r] = Xr
err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
print("max rel err vs np.fft:", err)
assert err < 1e-12
