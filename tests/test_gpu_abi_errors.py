"""Error behaviour of the C ABI on a live device: every misuse returns a negative pysdr_status
with a message in pysdr_last_error(), nothing aborts, and the context keeps working afterwards
(the reference's convention is "print + carry on", receiver.py:603-605)."""
import ctypes as C

import numpy as np
import pytest

from oracle import sdr_oracle as so
from pysdr_amd import _lib, design

pytestmark = pytest.mark.gpu


def _ctx(max_chunks=1, in_chunk=170666, ntaps=255):
    lib = _lib.lib()
    cfg = _lib.Cfg(8e6, 3, 500, in_chunk, max_chunks, ntaps, 255, 0, 0)
    h = C.c_void_p()
    _lib.check(lib.pysdr_create(C.byref(cfg), C.byref(h)), "create")
    return lib, h


def _taps():
    hdec = np.ascontiguousarray(design.decimator_bank(8e6, 3, 48000, 255)[0], np.float64)
    af = np.ascontiguousarray(design.af_bank_real(48000, 255)[3].astype(np.complex128)).view(np.float64)
    return hdec, af


def test_misuse_returns_status_and_context_survives():
    lib, h = _ctx()
    hdec, af = _taps()
    irx = C.c_int(-1)
    x = so.synth_iq(so.CONFIGS['C2'], 170666, 1)
    outs = (_lib.Out * 1)()
    # no receivers yet
    assert lib.pysdr_process(h, _lib.as_pf(x.view(np.float32)), len(x), outs) < 0
    assert b"no receivers" in lib.pysdr_last_error()
    # bad mode / null taps
    assert lib.pysdr_rx_add(h, 99, -455e3, _lib.as_pd(hdec), _lib.as_pd(af), 0.0, C.byref(irx)) < 0
    assert lib.pysdr_rx_add(h, 9, -455e3, None, _lib.as_pd(af), 0.0, C.byref(irx)) < 0
    for k in range(8):
        assert lib.pysdr_rx_add(h, 9, -455e3 + 1e3 * k, _lib.as_pd(hdec), _lib.as_pd(af), 0.0, C.byref(irx)) == 0
    assert irx.value == 7
    assert lib.pysdr_rx_add(h, 9, 0.0, _lib.as_pd(hdec), _lib.as_pd(af), 0.0, C.byref(irx)) < 0      # PYSDR_MAX_RX
    # setters on a sub-receiver that does not exist, wrong tap counts
    fa = C.c_double()
    assert lib.pysdr_set_lo(h, 8, 1.0, C.byref(fa)) < 0 and lib.pysdr_set_lo(h, -1, 1.0, C.byref(fa)) < 0
    assert lib.pysdr_set_dec_taps(h, 0, _lib.as_pd(hdec), 254) < 0
    assert lib.pysdr_set_mode(h, 0, 9, _lib.as_pd(af), 17, 0.0) < 0
    # more samples than the context was created for
    big = np.zeros(2 * 170666 + 5, np.complex64)
    outs8 = (_lib.Out * 8)()
    assert lib.pysdr_process(h, _lib.as_pf(big.view(np.float32)), len(big), outs8) < 0
    assert b"capacity" in lib.pysdr_last_error()
    assert lib.pysdr_process_batch(h, C.c_void_p(big.ctypes.data), 2, 170666, 0) < 0
    # output buffers too small
    am = np.empty(16, np.float32)
    outs8[0].am = _lib.as_pf(am)
    outs8[0].cap = 16
    assert lib.pysdr_process(h, _lib.as_pf(x.view(np.float32)), len(x), outs8) < 0
    # ... and after all that a proper call works
    for r in range(8):
        outs8[r].am, outs8[r].iq, outs8[r].cap = None, None, 0
    assert lib.pysdr_process(h, _lib.as_pf(x.view(np.float32)), len(x), outs8) == 0
    assert outs8[0].n_out in (1023, 1024, 1025) and outs8[0].peak_in > 0
    lib.pysdr_destroy(h)


def test_spectrum_waterfall_ingest_misuse():
    lib, h = _ctx()
    sp = C.c_void_p()
    win = np.ones(1024, np.float32)
    assert lib.pysdr_spectrum_create(0, 1024, 512, 1, _lib.as_pf(win), C.byref(sp)) < 0       # nfft < chunk
    assert lib.pysdr_spectrum_create(0, 1024, 2048, 0, _lib.as_pf(win), C.byref(sp)) < 0      # no frames
    _lib.check(lib.pysdr_spectrum_create(0, 1024, 2048, 4, _lib.as_pf(win), C.byref(sp)), "spectrum_create")
    d = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(0, 1 << 20, C.byref(d)), "alloc")
    assert lib.pysdr_spectrum_batch(sp, d, 5, 1024, d) < 0                                    # > max_frames
    assert lib.pysdr_spectrum_order(sp, h, 3) < 0
    assert lib.pysdr_spectrum_order(sp, h, 0) == 0 and lib.pysdr_spectrum_order(sp, h, 2) == 0
    ing = C.c_void_p()
    assert lib.pysdr_ingest_create(h, 1, C.byref(ing)) < 0                                    # needs >= 2 slots
    _lib.check(lib.pysdr_ingest_create(h, 2, C.byref(ing)), "ingest_create")
    outs = (_lib.Out * 1)()
    assert lib.pysdr_ingest_collect(ing, 0, outs) < 0                                         # never submitted
    assert lib.pysdr_ingest_submit(ing, 0, 170666) < 0                                        # no receivers
    assert lib.pysdr_ingest_submit(ing, 5, 10) < 0 and lib.pysdr_ingest_submit(ing, 0, 10 ** 9) < 0
    lib.pysdr_ingest_destroy(ing)
    lib.pysdr_dev_free(0, d)
    lib.pysdr_spectrum_destroy(sp)
    lib.pysdr_destroy(h)
    assert lib.pysdr_strerror(-5) and lib.pysdr_strerror(-99)


def test_setters_from_a_second_thread_while_processing():
    """The GUI thread retunes / swaps filters / changes mode while the RX thread demodulates
    (gui.py:1713,1938 vs receiver.py:684-725): setters take effect at a chunk boundary, nothing
    crashes or returns garbage."""
    import threading
    from tests.test_gpu_parity import make_gpu_receivers
    cfg = so.CONFIGS['C3']
    P, g = make_gpu_receivers(cfg)
    L = P.IN_CHUNK_SIZE
    x = so.synth_iq(cfg, 4 * L, 3)
    stop = threading.Event()
    errors = []

    def gui():
        k = 0
        try:
            while not stop.is_set():
                g[2].lo.change_freq(-455e3 - 50.0 * (k % 7))
                g[0].dec.h = g[0].dec.filter_bank[3 + (k % 4)]
                g[1].bfo = 650.0 + 10.0 * (k % 5)
                g[3].af_bw = (3e3, 5e3)[k % 2]
                if k % 11 == 0:
                    g[0].agc.reset()
                k += 1
        except Exception as e:          # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=gui)
    t.start()
    try:
        for k in range(40):
            xc = x[(k % 4) * L:(k % 4 + 1) * L]
            for rx in g:
                am = rx.demod_data(xc)
                assert len(am) in (1023, 1024, 1025) and np.all(np.isfinite(am)) and np.all(np.isfinite(rx.iq))
    finally:
        stop.set()
        t.join()
    assert not errors
    assert abs(g[2].lo.fo) > 0


def test_raw_abi_set_mode_and_set_lo_race_pysdr_process():
    """pysdr_set_mode (-> AM-Synch, which needs a buffer the context allocates lazily)
    and pysdr_set_lo called straight through the C ABI from a second thread WHILE pysdr_process
    runs (ctypes drops the GIL for the call).  One process call must see one consistent set of
    mode / NCO word / taps / buffers: no device fault, status OK, finite output."""
    import threading
    lib, h = _ctx()
    hdec, af = _taps()
    irx = C.c_int(-1)
    for k in range(3):
        assert lib.pysdr_rx_add(h, 0, -455e3 + 2e3 * k, _lib.as_pd(hdec), _lib.as_pd(af), 0.0,
                                C.byref(irx)) == 0
    x = so.synth_iq(so.CONFIGS['C2'], 170666, 1)
    stop = threading.Event()
    bad = []

    def setter():
        modes = (1, 0, 9, 5, 1, 3, 6)          # AM-Synch, AM, NFM, CW, AM-Synch, USB, IQ
        k = 0
        fa = C.c_double()
        while not stop.is_set():
            r = k % 3
            if lib.pysdr_set_mode(h, r, modes[k % len(modes)], _lib.as_pd(af), 255, 700.0) != 0:
                bad.append("set_mode")
            if lib.pysdr_set_lo(h, (r + 1) % 3, -455e3 - 25.0 * (k % 9), C.byref(fa)) != 0:
                bad.append("set_lo")
            k += 1

    t = threading.Thread(target=setter)
    t.start()
    try:
        am = [np.empty(2 * 1100, np.float32) for _ in range(3)]
        iq = [np.empty(2 * 1100, np.float32) for _ in range(3)]
        outs = (_lib.Out * 3)()
        for r in range(3):
            outs[r].am, outs[r].iq, outs[r].cap = _lib.as_pf(am[r]), _lib.as_pf(iq[r]), 1100
        for _ in range(150):
            rc = lib.pysdr_process(h, _lib.as_pf(x.view(np.float32)), len(x), outs)
            assert rc == 0, lib.pysdr_last_error()
            for r in range(3):
                n = outs[r].n_out
                assert n in (1023, 1024, 1025)
                k = 2 * n if outs[r].am_is_complex else n
                assert np.all(np.isfinite(am[r][:k])) and np.all(np.isfinite(iq[r][:2 * n]))
    finally:
        stop.set()
        t.join()
    assert not bad
    lib.pysdr_destroy(h)


def test_batched_ingest_ring_misuse():
    """Slots of several chunks (pysdr_ingest_create_batched / pysdr_ingest_chunks): capacity and call
    order violations return a status, and a proper submit / chunks / collect still works afterwards."""
    lib, h = _ctx(max_chunks=4)
    hdec, af = _taps()
    irx = C.c_int(-1)
    assert lib.pysdr_rx_add(h, 9, -455e3, _lib.as_pd(hdec), _lib.as_pd(af), 0.0, C.byref(irx)) == 0
    ing = C.c_void_p()
    assert lib.pysdr_ingest_create_batched(h, 3, 5, C.byref(ing)) < 0          # > max_chunks
    assert b"max_chunks" in lib.pysdr_last_error()
    assert lib.pysdr_ingest_create_batched(h, 3, 0, C.byref(ing)) < 0
    _lib.check(lib.pysdr_ingest_create_batched(h, 3, 4, C.byref(ing)), "create_batched")
    n = C.c_int(0)
    cn = np.zeros(4, np.int32)
    pk = np.zeros(4, np.float32)
    pcn, ppk = cn.ctypes.data_as(C.POINTER(C.c_int)), _lib.as_pf(pk)
    assert lib.pysdr_ingest_chunks(ing, 0, 4, C.byref(n), pcn, ppk) < 0         # never submitted
    p = C.POINTER(C.c_float)()
    cap = C.c_size_t(0)
    _lib.check(lib.pysdr_ingest_buffer(ing, 0, C.byref(p), C.byref(cap)), "buffer")
    assert cap.value == 4 * 170666
    buf = np.ctypeslib.as_array(p, shape=(2 * cap.value,)).view(np.complex64)
    buf[:] = so.synth_iq(so.CONFIGS['C2'], cap.value, 7)
    assert lib.pysdr_ingest_submit(ing, 0, cap.value + 1) < 0                   # more than the slot holds
    _lib.check(lib.pysdr_ingest_submit(ing, 0, 3 * 170666), "submit")           # three whole chunks
    assert lib.pysdr_ingest_submit(ing, 0, 170666) < 0                          # already in flight
    assert lib.pysdr_ingest_chunks(ing, 0, 2, C.byref(n), pcn, ppk) < 0         # cap < 3 chunks
    _lib.check(lib.pysdr_ingest_chunks(ing, 0, 4, C.byref(n), pcn, ppk), "chunks")
    assert n.value == 3 and all(c in (1023, 1024, 1025) for c in cn[:3]) and np.all(pk[:3] > 0)
    outs = (_lib.Out * 1)()
    _lib.check(lib.pysdr_ingest_collect(ing, 0, outs), "collect")
    assert outs[0].n_out == int(cn[:3].sum())
    assert lib.pysdr_ingest_chunks(ing, 0, 4, C.byref(n), pcn, ppk) < 0         # collected: released
    # a slot that is not a whole number of chunks is ONE (short) chunk
    _lib.check(lib.pysdr_ingest_submit(ing, 1, 100000), "submit short")
    _lib.check(lib.pysdr_ingest_chunks(ing, 1, 4, C.byref(n), pcn, ppk), "chunks")
    assert n.value == 1
    _lib.check(lib.pysdr_ingest_collect(ing, 1, outs), "collect")
    lib.pysdr_ingest_destroy(ing)
    lib.pysdr_destroy(h)


@pytest.mark.parametrize("ntaps_af", [2048, 1993, 1000])
def test_af_filter_length_up_to_the_bound_pysdr_create_accepts(ntaps_af):
    """pysdr_create takes ntaps_af <= 2048, and from 1993 taps on the AF FIR kernel asks for
    more than 64 KB of dynamic LDS (66.8 KB at 2048): every length the context accepts must run, and
    give the oracle's audio."""
    import numpy as np
    from oracle import sdr_oracle as so
    from tests.test_gpu_parity import make_gpu_receivers, relerr
    cfg = dict(so.CONFIGS['C1'], ntaps_dec=255,
               rx=[dict(frq=100e3, mode='AM', video_bw=10e3, af_bw=5e3), dict(frq=100e3, mode='IQ', video_bw=10e3, af_bw=5e3)])
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    x = so.synth_iq(cfg, 4 * L, 3)
    P, g = make_gpu_receivers(cfg, af_filt_len=ntaps_af)
    o = so.make_receivers(cfg, np.float32, ntaps_af=ntaps_af)
    for k in range(4):
        for rg, ro in zip(g, o):
            a, b = rg.demod_data(x[k * L:(k + 1) * L]), ro.demod_data(x[k * L:(k + 1) * L])
            assert relerr(a, b) <= 1e-5, (ntaps_af, ro.mode, k)
    # one past the bound is refused at creation, with a message
    from pysdr_amd import _lib
    cfgc = _lib.Cfg(2.048e6, 3, 128, L, 1, 255, 2049, 0, 0)
    h = C.c_void_p()
    assert _lib.lib().pysdr_create(C.byref(cfgc), C.byref(h)) != 0
