"""The drop-in boundary against the reference's own text: every `dsp.<name>`, `rx.<attr>...`,
`psd.<attr>`, ring-buffer method and `lo.<attr>` that /root/reference uses exists on the objects
pysdr_amd.sig_proc provides (scripts/check_surface.py: `ast` only, nothing of the reference is
imported).  Build-container only -- the reference does not travel to the GPU box."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (GPU box)")
def test_every_sig_proc_name_the_reference_uses_exists():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_surface
    rows, missing = check_surface.check()
    assert len(rows) >= 45
    kinds = {r[0] for r in rows}
    assert {"sig_proc", "Receiver", "spectrum", "ring_buffer2/3", "signal_generator", "convolver"} <= kinds
    names = {(r[0], r[1]) for r in rows}
    # spot checks that the collector really saw the call sites SURVEY 2.2 lists
    for want in [("Receiver", "demod_data"), ("Receiver", "lo.change_freq"), ("Receiver", "dec.h"),
                 ("Receiver", "demod.am_pll.reset"), ("Receiver", "agc.err"), ("spectrum", "periodogram"),
                 ("spectrum", "new_samps"), ("ring_buffer2/3", "push_zeros"), ("signal_generator", "quad_mixer")]:
        assert want in names, want
    assert not missing, missing


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (GPU box)")
def test_every_device_method_the_reference_calls_exists_on_the_synthetic_sdr():
    """`P.sdr.<method>(...)` / `sdr.<method>(...)` as the reference calls them -- the RX thread and the GUI's message pump in
    receiver.py, setupSDR / check_sdr_settings in utils.py:292-460, the canonical loop of soapy.py -- collected with `ast`
    (nothing of the reference is imported): the synthetic SoapySDR-shaped device answers every one of them, with the
    arities those call sites use, so that the reference's own device set-up could be pointed at it."""
    import ast
    from pysdr_amd.stream import SOAPY_SDR_RX, SynthSDR
    from pysdr_amd.synth import CONFIGS
    calls = {}
    for fn, lo, hi in (("receiver.py", 1, 10 ** 6), ("utils.py", 292, 460), ("soapy.py", 1, 10 ** 6)):
        tree = ast.parse(open(os.path.join("/root/reference", fn)).read())
        for n in ast.walk(tree):
            if not (isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and lo <= n.lineno <= hi):
                continue
            v = n.func.value
            if (isinstance(v, ast.Attribute) and v.attr == "sdr") or (isinstance(v, ast.Name) and v.id == "sdr"):
                calls.setdefault(n.func.attr, set()).add(len(n.args))
    assert len(calls) >= 30 and {"readStream", "getGainRange", "hasGainMode", "getGainMode", "setupStream"} <= set(calls)
    dev = SynthSDR(CONFIGS['C2'], nsamp=4096)
    missing = [m for m in calls if not callable(getattr(dev, m, None))]
    assert not missing, missing
    import inspect
    for m, arities in calls.items():
        sig = inspect.signature(getattr(dev, m))
        for k in arities:
            sig.bind(*range(k))                      # raises TypeError if that many positional arguments do not fit
    # the range object as receiver.py:328-330 reads it, and as utils.py:406 prints it
    r = dev.getGainRange(SOAPY_SDR_RX, 0, 'TUNER')
    assert (r.minimum(), r.maximum(), r.step()) == tuple(r) and r.minimum() <= r.maximum()
    dev.setGainMode(SOAPY_SDR_RX, 0, True)
    assert dev.hasGainMode(SOAPY_SDR_RX, 0) and dev.getGainMode(SOAPY_SDR_RX, 0) is True
    dev.setBandwidth(SOAPY_SDR_RX, 0, 1.5e6)
    assert dev.getBandwidth(SOAPY_SDR_RX, 0) == 1.5e6
    dev.activateStream(dev.setupStream(SOAPY_SDR_RX, 'CF32'))
    got = dev.readStreamRTL(0, 1000)                 # receiver.py:598-600: an array, as long as what was ready
    assert got.dtype.name == 'complex64' and 0 < len(got) <= 1000
