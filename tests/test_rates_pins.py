"""The PRODUCT's rate / size arithmetic (pysdr_amd/rates.py, what `sig_proc.up_dn` and the run-time
parameter object hand to params.py:405-472's callers) and the oracle's restatement of it, bit for
bit against tests/golden/rates_ref.json: 780 rows produced by EXECUTING the reference's own text
(params.py:405-406,440-449,456-472, utils.py:277-289; `up_dn` from the 39 answers of srates.py:35-74;
generator: tests/golden/make_rates_golden.py, build container only).  CPU only."""
import json
import os

import pytest

from oracle import sdr_oracle as so
from pysdr_amd import rates
from pysdr_amd.params import RunTimeParams

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rates_ref.json")
ROWS = json.load(open(GOLD))["rows"]


def test_fixture_is_the_references_table():
    assert len(ROWS) == 39 * 2 * 2 * 5
    assert len({(r["SRATE"], r["FS_OUT_REQ"]) for r in ROWS}) == 39
    # the three BASELINE rates: 3/128 43690, 3/500 170666, 3/625 213333
    pick = {(r["SRATE"], r["FS_OUT_REQ"]): (r["UP"], r["DOWN"], r["IN_CHUNK_SIZE"]) for r in ROWS}
    assert pick[(2048000, 48000)] == (3, 128, 43690) and pick[(8000000, 48000)] == (3, 500, 170666)
    assert pick[(10000000, 48000)] == (3, 625, 213333)


def test_product_rates_equal_the_executed_reference():
    import pysdr_amd.sig_proc as dsp                 # `from sig_proc import up_dn` (params.py:405)
    for r in ROWS:
        fs, fso = r["SRATE"], r["FS_OUT_REQ"]
        assert dsp.up_dn(fs, fso) == rates.up_dn(float(fs), float(fso)) == (r["UP"], r["DOWN"]), (fs, fso)
        d = rates.derive(float(fs), float(fso))
        assert (d["UP"], d["DOWN"], d["FS_OUT"], d["IN_CHUNK_SIZE"]) == (r["UP"], r["DOWN"], r["FS_OUT"], r["IN_CHUNK_SIZE"])
        assert type(d["FS_OUT"]) is int and type(d["IN_CHUNK_SIZE"]) is int
        rb = rates.ring_buffer_size(r["NUM_RX"], r["SDR_TYPE"], d["FS_OUT"])
        assert rb == r["RB_SIZE"]
        assert rates.adjust_foffset(r["FOFFSET_IN"], float(fs), rb) == r["FOFFSET"]        # same float, bit for bit


def test_run_time_params_object_equals_the_executed_reference():
    for r in ROWS:
        if r["FOFFSET_IN"] == 0.0:
            continue                                 # FOFFSET == 0 takes the params.py:311-316 branch first
        P = RunTimeParams(fs=r["SRATE"], fsout=r["FS_OUT_REQ"], fc=[7.0e6] * r["NUM_RX"], foffset=r["FOFFSET_IN"],
                          sdr_type=r["SDR_TYPE"])
        got = (P.UP, P.DOWN, P.FS_OUT, P.IN_CHUNK_SIZE, P.MUTE_CHUNKS, P.RB_SIZE, P.FOFFSET)
        assert got == (r["UP"], r["DOWN"], r["FS_OUT"], r["IN_CHUNK_SIZE"], r["MUTE_CHUNKS"], r["RB_SIZE"], r["FOFFSET"]), r


def test_oracle_restatement_equals_the_executed_reference():
    for r in ROWS:
        fs, fso = float(r["SRATE"]), float(r["FS_OUT_REQ"])
        assert so.chunk_sizes(fs, fso) == (r["UP"], r["DOWN"], r["FS_OUT"], r["IN_CHUNK_SIZE"])
        rb = so.rb_size(r["NUM_RX"], r["SDR_TYPE"], r["FS_OUT"])
        assert rb == r["RB_SIZE"] and so.adjust_foffset(r["FOFFSET_IN"], fs, rb) == r["FOFFSET"]


def test_af_gain_slider_is_receiver_py_200():
    # receiver.py:200  af_gain = pow(10.,P.AF_GAIN)-1 ; slider AF_GAIN = 2*v/100 (gui.py:1050)
    for v in (0, 1, 25, 50, 99, 100):
        assert rates.af_gain(2 * v / 100.0) == pow(10., 2 * v / 100.0) - 1 == so.af_gain(2 * v / 100.0)
