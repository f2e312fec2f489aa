"""Generate tests/golden/rates_ref.json by EXECUTING the reference's own rate / size arithmetic
(run in the build container only: /root/reference does not travel).

The statements are taken, by line number, from the text of /root/reference/params.py
(:405-406 UP/DOWN/FS_OUT, :440 OUT_CHUNK_SIZE, :444 IN_CHUNK_SIZE, :448-449 MUTE_CHUNKS, :456-468
RB_SIZE rules, :472 adjust_foffset) and the function `adjust_foffset` of /root/reference/utils.py
(:277-289), compiled as they stand and run on a plain attribute bag that plays `self`.  The only
name the reference resolves outside its tree is `up_dn` (module sig_proc of aa2il/libs, absent):
it is served from the 39 answers the reference itself holds in srates.py:35-74, parsed from that
file's comment table -- so every number in the fixture is the reference's, none is this build's.
Nothing of the reference's text is stored: the fixture holds inputs and outputs only.

    python tests/golden/make_rates_golden.py
"""
import ast
import json
import os
import re
import textwrap
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def srates_table():
    rows = {}
    for line in open(os.path.join(REF, "srates.py")):
        m = re.match(r"#\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)", line)
        if m:
            rows[(int(float(m.group(1)) * 1e6), int(float(m.group(2)) * 1e3))] = (int(m.group(3)), int(m.group(4)))
    assert len(rows) == 39, len(rows)
    return rows


def params_block():
    src = open(os.path.join(REF, "params.py")).read().splitlines()
    want = [405, 406, 440, 444, 448, 449, 456, 459, 460, 462, 463, 465, 466, 467, 468, 472]
    text = textwrap.dedent("\n".join(src[i - 1] for i in want))
    # what was picked must be exactly the statements named in the docstring
    for frag in ("up_dn(self.SRATE", "self.FS_OUT = int(", "self.OUT_CHUNK_SIZE = 1024", "self.IN_CHUNK_SIZE  = int(",
                 "self.MUTE_CHUNKS = int(", "self.RB_SIZE        = 32*", "if self.NUM_RX>2", "== 'rtlsdr'",
                 "if self.FS_OUT>100e3", "elif self.FS_OUT>50e3", "adjust_foffset(self)"):
        assert frag in text, frag
    return compile(text, "params.py:405-472", "exec")


def utils_adjust_foffset():
    tree = ast.parse(open(os.path.join(REF, "utils.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "adjust_foffset"][0]
    ns = {}
    exec(compile(ast.Module([fn], []), "utils.py:277-289", "exec"), ns)
    return ns["adjust_foffset"]


def main():
    kat = srates_table()
    block = params_block()
    adj = utils_adjust_foffset()
    rows = []
    for (fs, fso), (up, dn) in sorted(kat.items()):
        for num_rx in (1, 3):
            for sdr_type in ("sdrplay", "rtlsdr"):
                for fo in (0.0, 100e3, -37.5e3, 455e3, 1234567.0):
                    self = types.SimpleNamespace(SRATE=float(fs), FS_OUT=float(fso), NUM_RX=num_rx, SDR_TYPE=sdr_type, FOFFSET=fo)
                    exec(block, dict(self=self, up_dn=lambda a, b: kat[(int(a), int(b))], adjust_foffset=adj))
                    rows.append(dict(SRATE=fs, FS_OUT_REQ=fso, NUM_RX=num_rx, SDR_TYPE=sdr_type, FOFFSET_IN=fo,
                                     UP=self.UP, DOWN=self.DOWN, FS_OUT=self.FS_OUT, IN_CHUNK_SIZE=self.IN_CHUNK_SIZE,
                                     MUTE_CHUNKS=self.MUTE_CHUNKS, RB_SIZE=self.RB_SIZE, FOFFSET=self.FOFFSET))
    json.dump(dict(source="executed text of /root/reference/params.py:405-472 and utils.py:277-289; up_dn from srates.py:35-74",
                   rows=rows), open(os.path.join(HERE, "rates_ref.json"), "w"))
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
