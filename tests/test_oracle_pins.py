"""The oracle against every in-tree pin of the reference and
against scipy's independent resampler.  CPU only."""
import numpy as np
import pytest
from scipy import signal

from oracle import sdr_oracle as so

# srates.py:35-74 -- the 39 known answers (fs MHz, fs_out kHz, UP, DOWN), pasted stdout
SRATES_KAT = """
0.25 48 24 125|0.25 96 48 125|0.25 192 96 125|0.5 48 12 125|0.5 96 24 125|0.5 192 48 125|
1 48 6 125|1 96 12 125|1 192 24 125|2 48 3 125|2 96 6 125|2 192 12 125|
2.048 48 3 128|2.048 96 3 64|2.048 192 3 32|3 48 2 125|3 96 4 125|3 192 8 125|
4 48 3 250|4 96 3 125|4 192 6 125|5 48 6 625|5 96 12 625|5 192 24 625|
6 48 1 125|6 96 2 125|6 192 4 125|7 48 6 875|7 96 12 875|7 192 24 875|
8 48 3 500|8 96 3 250|8 192 3 125|9 48 2 375|9 96 4 375|9 192 8 375|
10 48 3 625|10 96 6 625|10 192 12 625
"""


def _kat_rows():
    rows = []
    for item in SRATES_KAT.replace("\n", "").split("|"):
        f1, f2, up, dn = item.split()
        rows.append((int(float(f1) * 1e6), int(float(f2) * 1e3), int(up), int(dn)))
    return rows


def test_up_dn_known_answers():
    rows = _kat_rows()
    assert len(rows) == 39
    for fs1, fs2, up, dn in rows:
        assert so.up_dn(fs1, fs2) == (up, dn), (fs1, fs2)


def test_chunk_sizes_params_py():
    # params.py:405-406,444 ; SURVEY 8 sizes
    assert so.chunk_sizes(2.048e6, 48e3) == (3, 128, 48000, 43690)
    assert so.chunk_sizes(8e6, 48e3) == (3, 500, 48000, 170666)
    assert so.chunk_sizes(10e6, 48e3) == (3, 625, 48000, 213333)


def test_adjust_foffset_and_rb_size():
    # utils.py:277-289 ; params.py:456-468
    rb = so.rb_size(1, 'sdrplay', 48000)
    assert rb == 32 * 1024
    assert so.rb_size(4, 'sdrplay', 48000) == 4 * 32 * 1024
    assert so.rb_size(1, 'rtlsdr', 96000) == 32 * 1024 * 2 * 2
    fo = so.adjust_foffset(100e3, 2.048e6, rb)
    m = fo * rb / 2.048e6
    assert abs(m - round(m)) < 1e-9 and abs(fo - 100e3) <= 0.5 * 2.048e6 / rb


def test_af_gain_slider():
    assert so.af_gain(0.5) == pytest.approx(10 ** 0.5 - 1)      # receiver.py:200


def test_nfm_discriminator_matches_octave_formula():
    # sigs/nfm.m:124-127 evaluated literally
    rng = np.random.default_rng(5)
    y = (rng.standard_normal(64) + 1j * rng.standard_normal(64))
    IQ = y[2:]
    d = IQ - y[:-2]
    y1 = y[1:-1]
    fm = y1.real * d.imag - y1.imag * d.real
    got = so.nfm_discriminator(y, np.float64)
    assert np.allclose(got, fm / (2 * np.abs(y1) ** 2 + 1e-20), rtol=1e-12)
    # a pure tone of normalised frequency w gives sin(w)
    w = 0.3
    t = np.exp(1j * w * np.arange(50))
    assert np.allclose(so.nfm_discriminator(t, np.float64), np.sin(w), atol=1e-12)


def test_agc_decay_is_the_pinned_loop_filter():
    # sigs/agc.m:6-12: y = filter(beta, [1 beta-1], x), beta = .1
    agc = so.AGC(np.float64)
    agc.update(1.0, True)                      # attack: env jumps to the peak
    x = np.full(50, 0.2)
    env = []
    for v in x:
        agc.update(v, True)
        env.append(agc.agc)
    ref, _ = signal.lfilter([0.1], [1, 0.1 - 1], x, zi=[(1 - 0.1) * 1.0])
    assert np.allclose(env, ref, rtol=1e-12)
    assert agc.gain == pytest.approx(so.AGC_REF / env[-1])
    assert agc.maxbuf == 0.2 and agc.err == pytest.approx(so.AGC_REF - agc.gain * 0.2)


def test_psd_formula_matches_rtty_py():
    # rtty.py:839-841: fftshift(fft(x*window, NFFT)) -> 10*log10(re^2+im^2)
    rng = np.random.default_rng(7)
    n, nfft = 512, 1024
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    sp = so.Spectrum(48.0, n, nfft, 0.0, dtype=np.float64)
    got = sp.periodogram(x, True)
    X = np.fft.fftshift(np.fft.fft(x * so.psd_window(n), nfft))
    want = 10 * np.log10(np.square(X.real) + np.square(X.imag) + so.PSD_FLOOR)
    assert np.allclose(got, want, atol=1e-9)
    assert len(sp.frq) == nfft and sp.frq[nfft // 2] == 0
    # real input -> positive half only
    assert len(sp.periodogram(x.real, True)) == nfft // 2
    # unit tone reads 0 dB at its bin (unit coherent gain)
    k = 100
    tone = np.exp(2j * np.pi * k * np.arange(n) / nfft)
    p = so.Spectrum(48.0, n, nfft, 0.0, dtype=np.float64).periodogram(tone)
    assert np.argmax(p) == nfft // 2 + k and abs(p.max()) < 1e-6


@pytest.mark.parametrize("up,down,ntaps", [(3, 500, 255), (3, 128, 1001), (24, 125, 301), (1, 125, 255)])
def test_decimator_equals_scipy_upfirdn(up, down, ntaps):
    rng = np.random.default_rng(up * 1000 + down)
    n = 20000
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    h = so.dec_filter_bank(down * 16e3, up, up * 16e3, ntaps)[2]
    dec = so.RationalDecimator(h, up, down, np.float64)
    y = dec.process(x)
    ref = signal.upfirdn(h, x, up, down)
    assert len(y) == -(-n * up // down)
    assert np.allclose(y, ref[:len(y)], atol=1e-12)


@pytest.mark.parametrize("chunks", [[43690] * 3, [170666, 1, 7, 170666], [1000, 213333, 999]])
def test_chunked_equals_one_shot(chunks):
    # sigs/iir.py:83-125: processing in chunks with carried state == one shot
    cfg = so.CONFIGS['C3']
    n = sum(chunks)
    x = so.synth_iq(cfg, n, 11)
    one = so.make_receivers(cfg, np.float64)
    many = so.make_receivers(cfg, np.float64)
    for a, b in zip(one, many):
        # AGC blocks follow chunk boundaries, so compare the pre-AGC signal path
        a.agc.ref = b.agc.ref = np.float64(so.AGC_REF)
        ya = a.dec.process(a.lo.quad_mixer(x))
        parts, pos = [], 0
        for c in chunks:
            parts.append(b.dec.process(b.lo.quad_mixer(x[pos:pos + c])))
            pos += c
        yb = np.concatenate(parts)
        assert len(ya) == len(yb)
        assert np.max(np.abs(ya - yb)) < 1e-12
        da = a.demod.process(ya, a.mode, a.bfo)
        db = np.concatenate([b.demod.process(p, b.mode, b.bfo) for p in parts])
        assert np.max(np.abs(da - db)) < 1e-12


def test_tone_in_tone_out():
    # AM carrier with a 1 kHz tone -> audio dominated by 1 kHz; NFM 3 kHz dev -> 0.6 FS
    for name, idx, f_audio in (('C1', 0, 1000.0), ('C2', 0, 1000.0)):
        cfg = so.CONFIGS[name]
        L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
        x = so.synth_iq(cfg, 12 * L, 1)
        rx = so.make_receivers(cfg, np.float64)[idx]
        am = np.concatenate([rx.demod_data(x[i * L:(i + 1) * L]) for i in range(12)])[4096:]
        am = am - am.mean()
        spec = np.abs(np.fft.rfft(am * np.hanning(len(am))))
        fpk = np.argmax(spec) * 48000.0 / len(am)
        assert abs(fpk - f_audio) < 20.0, (name, fpk)
        if name == 'C2':
            assert abs(np.max(np.abs(am)) - 3000.0 / so.NFM_FULL_SCALE_DEV) < 0.05


def test_float32_mirror_tracks_float64_master():
    cfg = so.CONFIGS['C3']
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    x = so.synth_iq(cfg, 4 * L, 3)
    r32, r64 = so.make_receivers(cfg, np.float32), so.make_receivers(cfg, np.float64)
    for a, b in zip(r32, r64):
        for i in range(4):
            ya = a.demod_data(x[i * L:(i + 1) * L])
            yb = b.demod_data(x[i * L:(i + 1) * L])
        assert np.max(np.abs(ya - yb)) <= 2e-5 * np.max(np.abs(yb)), a.mode


def test_waterfall_numerics_plotting_py():
    # Plotting.py:385,539-547,583-587,618-626,689-695
    nfft, ncol = 64, 100
    wf = -1e38 * np.ones((nfft, ncol))
    rng = np.random.default_rng(0)
    cnt = 0
    for _ in range(5):
        wf = so.waterfall_push(wf, rng.standard_normal(nfft))
        cnt += 1
    assert wf.shape == (nfft, ncol) and np.all(wf[:, :-5] == -1e38)
    short = so.waterfall_push(wf, np.zeros(10))
    assert np.all(short[10:, -1] == -1e38)
    img, bk, psd2 = so.waterfall_image(wf, cnt, 60.0)
    assert bk == np.median(np.mean(wf[:, -cnt:], 1))
    assert img.min() >= np.nanmax(wf - bk) - 60.0 - 1e-9
    rolled, fc = so.waterfall_roll(wf, 0.0, 3.2, 1.0)
    assert fc == 3.2 and np.array_equal(rolled, np.roll(wf, -3, axis=0))


def _replay_waterfall_fixture(push, roll, image):
    """Drive `push(line, flip)`, `roll(fc, df)`, `image(pan_dr, npsd) -> (img, bk, psd2)` through the
    sequence of tests/golden/waterfall_ref.npz and yield (k, got, fixture) at its snapshots."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "waterfall_ref.npz"))
    retune = dict(zip([int(k) for k in g["retune_k"]], [float(f) for f in g["retune_fc"]]))
    snaps = set(int(k) for k in g["snap_k"])
    df, flips = float(g["df"]), set(int(k) for k in g["flip_k"])
    for k in range(len(g["lines"])):
        n = int(g["lens"][k])
        if k in retune:
            roll(retune[k], df)
        push(g["lines"][k][:n], k in flips)
        if k in snaps:
            yield k, image(float(g["pan_dr"]), n), g


def test_oracle_waterfall_equals_the_executed_reference_text():
    """tests/golden/waterfall_ref.npz = Plotting.py:385-388,536-548,583-626,689-695 EXECUTED as they
    stand on 130 pushed lines (half-length lines, one RIG_IF<0 flip, two retunes, history wrapped):
    tests/golden/make_waterfall_ref_golden.py, build container only.  The oracle's restatement gives
    the same image, background, averaged PSD and peaks."""
    st = dict(wf=-1e38 * np.ones((512, 100)), cnt=0, fc=0.0)

    def push(line, flip):
        line = np.asarray(line, np.float64)
        st["wf"] = so.waterfall_push(st["wf"], line[::-1] if flip else line)
        st["cnt"] = min(st["cnt"] + 1, 100)

    def roll(fc, df):
        st["wf"], st["fc"] = so.waterfall_roll(st["wf"], st["fc"], fc, df)

    seen = 0
    for k, (img, bk, psd2), g in _replay_waterfall_fixture(push, roll, lambda dr, n: so.waterfall_image(st["wf"], st["cnt"], dr, n)):
        assert st["cnt"] == int(g[f"cnt{k}"])
        assert img.shape == g[f"img{k}"].shape and np.array_equal(img.astype(np.float32), g[f"img{k}"])
        assert bk == float(g[f"bk{k}"]) and np.array_equal(psd2, g[f"psd2_{k}"])
        assert np.array_equal(so.find_peaks_db(psd2, bk, float(g["peak_dist"]) / float(g["df"])), g[f"peaks{k}"])
        seen += 1
    assert seen == 4


def test_written_out_find_peaks_equals_the_executed_reference_statement():
    """`oracle.find_peaks_greedy` (the checker of the device peak pick) against tests/golden/peaks_ref.npz: the reference's
    `signal.find_peaks(PSD2, distance=dist, height=bkgnd+10)` (Plotting.py:587,594,596 executed as they stand,
    tests/golden/make_peaks_ref_golden.py) on every tie-free line, index for index; and against SciPy itself here on random
    lines whose heights are all distinct (flat tops of every width included)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks_ref.npz"))
    seen = 0
    for k in range(int(g["ncases"])):
        if int(g[f"tie{k}"]):
            continue
        x, bk = g[f"line{k}"].astype(np.float64), float(g[f"bk{k}"])
        assert abs(bk - float(np.median(x))) == 0.0
        got = so.find_peaks_greedy(x, bk + 10.0, float(g[f"peak_dist{k}"]) / float(g[f"df{k}"]))
        assert np.array_equal(got, g[f"peaks{k}"]), k
        seen += 1
    assert seen >= 7
    rng = np.random.default_rng(31)
    for trial in range(60):
        n = int(rng.choice([3, 8, 17, 100, 1000, 8192]))
        m = max(1, n // int(rng.choice([1, 2, 5])))
        x = np.repeat(rng.permutation(m).astype(np.float32) * 0.25 - 3.0, rng.integers(1, 6, m))[:n]
        if len(x) < n:
            x = np.r_[x, rng.permutation(n - len(x)).astype(np.float32) + 1e4]
        h, dist = float(np.median(x)), float(rng.choice([1.0, 2.5, 7.0, 33.3]))
        want, _ = signal.find_peaks(x, distance=dist, height=h)
        assert np.array_equal(so.find_peaks_greedy(x, h, dist), want), (trial, n, dist)
