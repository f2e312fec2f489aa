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
