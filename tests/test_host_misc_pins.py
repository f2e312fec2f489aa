"""The product's small host logic against tests/golden/host_misc_ref.json, a fixture made by EXECUTING
the reference's own text (Tables.py:34-62, params.py:291-329, receiver.py:633-651,826-835;
tests/golden/make_host_misc_ref_golden.py, build container only).  CPU only."""
import json
import os
import types

import numpy as np

from pysdr_amd import executive, tables
from pysdr_amd.params import RunTimeParams

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_misc_ref.json")))


def test_tables_and_find_filter():
    t = G["tables"]
    assert tables.MODES == t["MODES"] and tables.AF_BWs == t["AF_BWs"] and tables.VIDEO_BWs == t["VIDEO_BWs"]
    assert tables.RTLsrates == t["RTLsrates"] and tables.SDRplaysrates == t["SDRplaysrates"]
    for row in G["find_filter"]:
        if row["video"] is not None:
            assert tables.find_filter(row["max_bw"], tables.VIDEO_BWs) == row["video"], row
        if row["af"] is not None:
            assert tables.find_filter(row["max_bw"], tables.AF_BWs) == row["af"], row


def test_run_time_params_defaults():
    for r in G["params"]:
        P = RunTimeParams(fs=2.048e6, fc=r["fc"], mode=r["mode"], foffset=r["foffset"], vid_bw=r["vid_bw"] * 1e3, bfo=r["bfo"],
                          audio=r["audio"], src=r["src"] if r["src"] else None)
        assert [int(v) for v in P.SOURCE] == r["SOURCE"], r
        assert P.NUM_PLAYERS == r["NUM_PLAYERS"] and P.BFO == r["BFO"] and P.VIDEO_BW == r["VIDEO_BW"], r
        if r["foffset"] == 0.0:
            # params.py:311-314 centres the LO; the snap to M*SRATE/RB_SIZE (utils.py:277-289) comes on top
            m = round(P.RB_SIZE * r["FOFFSET"] / P.SRATE)
            assert P.FOFFSET == m * P.SRATE / P.RB_SIZE, r


def test_create_receivers_offsets_and_mode_change():
    for r in G["create_Receivers"]:
        made = []
        dsp = types.SimpleNamespace(Receiver=lambda P, frq, irx, name, vb, ab: made.append((float(frq), name)) or types.SimpleNamespace())
        P = types.SimpleNamespace(FOFFSET=r["foffset"], NUM_RX=len(r["fc"]), SOURCE=np.array(r["source"]), FC=np.array(r["fc"]),
                                  rx=[None] * len(r["fc"]))
        ex = types.SimpleNamespace(P=P, dsp=dsp)
        executive.SDR_EXECUTIVE.create_Receivers(ex)
        assert [m[0] for m in made] == r["frq"] and [m[1] for m in made] == r["names"], r
    for r in G["mode_change"]:
        calls = []
        rx0 = types.SimpleNamespace(agc=types.SimpleNamespace(reset=lambda: calls.append("agc")),
                                    demod=types.SimpleNamespace(am_pll=types.SimpleNamespace(reset=lambda: calls.append("pll"))))
        P = types.SimpleNamespace(MODE_CHANGE=True, MODE=r["old"], NEW_MODE=r["new"], MP_SCHEME=r["mp_scheme"], FREQ_CHANGE=False, rx=[rx0])
        executive.SDR_EXECUTIVE.mode_freq_change(types.SimpleNamespace(P=P))
        assert (P.MODE, P.NEW_MODE, bool(P.MODE_CHANGE)) == (r["MODE"], r["NEW_MODE"], r["MODE_CHANGE"]), r
        assert calls == r["resets"], r
