"""N3: the RTTY filterbank.  CPU: the oracle's parameter arithmetic against the values the
reference prints for FS_OUT = 48 kHz (rtty.py:376-404) and its sliding behaviour; GPU: the
batched device filterbank against the oracle."""
import numpy as np
import pytest

from oracle import rtty_oracle as ro


def test_params_for_48k():
    p = ro.RttyParams(48000)
    assert (p.N, p.NFFT, p.NSTART, p.M) == (1056, 2048, [0, 264, 528, 792], 30)
    assert p.NBINS == round(170 / (48000 / 2048)) == 7
    assert len(p.frq) == 2048 and p.frq[1024] == 0.0
    q = ro.RttyParams(44100)                       # non-integer quarter step
    assert (q.N, q.NFFT, q.NSTART) == (970, 1024, [0, 243, 485, 728])


def test_lines_formula_flip_and_streaming():
    fs = 48000
    x, bits = ro.synth_rtty(fs, 40, 2000.0, seed=2)
    fb = ro.RttyFilterbank(fs)
    lines = fb.push(x)
    assert lines.shape == (4 * 39, 2048)           # first symbol primes `prev`
    # line 4 = quarter 0 of symbol 2 = samples [N, 2N) (prev = symbol 1 ... rtty.py:831-840)
    N = fb.p.N
    xx = x[0:N].astype(np.complex128) * np.kaiser(N, 8.6)
    X = np.fft.fftshift(np.fft.fft(xx, 2048))
    want = np.flipud(10 * np.log10(X.real ** 2 + X.imag ** 2))
    assert np.allclose(lines[0], want, rtol=0, atol=1e-9)
    # the flip puts +f at index NFFT-1-(NFFT/2 + k): the mark tone (2 kHz) peaks there
    k = int(round(2000.0 / (fs / 2048)))
    peak = int(np.argmax(lines.mean(axis=0)))
    assert abs(peak - (2047 - (1024 + k))) <= 8
    # ragged pushes produce the same lines
    fb2 = ro.RttyFilterbank(fs)
    parts = [fb2.push(c) for c in np.array_split(x, 17)]
    assert np.array_equal(np.concatenate([p for p in parts if len(p)]), lines)
    # mark/space discriminate the bits: symbol s drives lines 4(s-1)..4(s-1)+3
    mb = 2047 - (1024 + k)
    mark, space = ro.mark_space(lines, mb, fb.p.NBINS)
    sig = (mark - space).reshape(-1, 4)[:, 0]      # quarter 0 = exactly one symbol
    assert np.array_equal(sig > 0, bits[:39] > 0)


@pytest.mark.gpu
def test_device_filterbank_matches_oracle():
    from pysdr_amd import rtty
    fs = 48000
    x, bits = ro.synth_rtty(fs, 300, -3000.0, seed=5, noise=3e-3)
    o = ro.RttyFilterbank(fs)
    g = rtty.RTTY_Filterbank(fs, max_symbols=128)
    assert (g.RTTY.N, g.RTTY.NFFT, g.RTTY.NSTART, g.RTTY.NBINS, g.RTTY.M) == (1056, 2048, [0, 264, 528, 792], 7, 30)
    want = o.push(x)
    got = np.concatenate([g.push(c) for c in np.array_split(x, 7)])      # batches of <= 128 symbols, ragged pushes
    assert got.shape == want.shape == (4 * 299, 2048)
    # every bin of every line in linear power: 1e-5 of the line's peak and the per-bin bound of a
    # float32 transform (the window is not normalised: psd_check scales by the peak)
    from tests.test_gpu_parity import psd_check
    for j in range(len(want)):
        psd_check(got[j], want[j])
    k = int(round(-3000.0 / (fs / 2048)))
    mb = 2047 - (1024 + k)
    mg, sg = g.mark_space(got, mb)
    mo, so_ = ro.mark_space(want, mb, 7)
    assert np.array_equal((mg - sg) > 0, (mo - so_) > 0)
    g.close()


def _ref():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rtty_ref.npz"))
    return g["x"], g["lines"], [int(v) for v in g["params"]]


def test_oracle_filterbank_equals_the_executed_reference_text():
    """tests/golden/rtty_ref.npz holds lines made by EXECUTING rtty.py:376-404 (RTTY_Params) and
    rtty.py:807,831,837-843 (window, prev|cur, slice, FFT, fftshift, 10 log10, flipud) as they stand
    (tests/golden/make_rtty_ref_golden.py, build container only): the oracle's restatement gives the
    same numbers and the same parameters."""
    x, lines, prm = _ref()
    fb = ro.RttyFilterbank(48000)
    assert [fb.p.N, fb.p.NFFT, fb.p.NBINS, fb.p.M] + list(fb.p.NSTART) == prm
    got = fb.push(x)
    assert got.shape == lines.shape == (24, 2048)
    assert np.allclose(got, lines, rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_device_filterbank_equals_the_executed_reference_text():
    from pysdr_amd import rtty
    from tests.test_gpu_parity import psd_check
    x, lines, prm = _ref()
    g = rtty.RTTY_Filterbank(48000, max_symbols=4)
    assert [g.RTTY.N, g.RTTY.NFFT, g.RTTY.NBINS, g.RTTY.M] + list(g.RTTY.NSTART) == prm
    got = np.concatenate([g.push(c) for c in np.array_split(x, 5)])
    assert got.shape == lines.shape
    for j in range(len(lines)):
        psd_check(got[j], lines[j])
    g.close()
