"""Generate tests/golden/host_misc_ref.json by EXECUTING small pieces of the reference's host logic as
they stand (build container only: /root/reference does not travel):

  Tables.py:34-45   the MODES / AF_BWs / VIDEO_BWs / RTLsrates / SDRplaysrates lists (assignments run)
  Tables.py:48-62   find_filter (function, extracted with `ast`) on a grid of bandwidths
  params.py:291-302 SOURCE padding and NUM_PLAYERS per AUDIO_SCHEME
  params.py:311-329 FOFFSET == 0 centring, CW BFO default, VIDEO_BW defaults
  receiver.py:633-651  SDR_EXECUTIVE.mode_freq_change, the mode part ('FM' -> 'NFM', AGC / PLL reset)
  receiver.py:826-835  create_Receivers: the LO offset each dsp.Receiver is built with (SOURCE / FOFFSET)

Line-picked statements run on attribute bags (`self`, `P`, `args`); `dsp.Receiver` and the reset methods
are recorders.  The fixture holds inputs and what the reference computed: data only.

    python tests/golden/make_host_misc_ref_golden.py
"""
import ast
import json
import os
import textwrap
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def lines(path, *nums):
    src = open(os.path.join(REF, path)).read().splitlines()
    return textwrap.dedent("\n".join(src[i - 1] for i in nums))


def func(path, name, cls=None):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = tree.body
    if cls:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    fn = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    return fn


def main():
    out = {}
    # ---- Tables.py
    ns = {}
    exec(compile(lines("Tables.py", 34, 36, 37, 41, 42, 44, 45), "Tables.py:34-45", "exec"), ns)
    out["tables"] = {k: ns[k] for k in ("MODES", "AF_BWs", "VIDEO_BWs", "RTLsrates", "SDRplaysrates")}
    ff = func("Tables.py", "find_filter")
    assert (ff.lineno, ff.end_lineno) == (48, 62)
    fns = dict(print=lambda *a, **k: None)
    exec(compile(ast.Module([ff], []), "Tables.py:48-62", "exec"), fns)
    grid = [60.0, 499.0, 500.0, 2.4e3, 3e3, 9.9e3, 10e3, 48e3, 96e3, 120e3, 250e3, 1.024e6, 2e6, 8e6]
    out["find_filter"] = [dict(max_bw=b, video=fns["find_filter"](b, ns["VIDEO_BWs"]) if b >= 5e3 else None,
                               af=fns["find_filter"](b, ns["AF_BWs"]) if b >= 50 else None) for b in grid]
    # ---- params.py: SOURCE / NUM_PLAYERS / FOFFSET / BFO / VIDEO_BW
    src_block = compile(lines("params.py", 291, 292, 293, 294, 295, 296), "params.py:291-296", "exec")
    players_block = compile(lines("params.py", 299, 300, 301, 302), "params.py:299-302", "exec")
    fo_block = compile(lines("params.py", 311, 312, 314), "params.py:311-314", "exec")
    bfo_block = compile(lines("params.py", 318, 319, 320), "params.py:318-320", "exec")
    vbw_block = compile(lines("params.py", 324, 325, 326, 327, 328, 329), "params.py:324-329", "exec")
    for frag, text in (("src = np.array(args.src)*1", lines("params.py", 291)), ("if self.AUDIO_SCHEME==1:", lines("params.py", 299)),
                       ("if self.FOFFSET==0:", lines("params.py", 311)), ("self.FOFFSET = fo-max(fc)", lines("params.py", 314)),
                       ("self.BFO             = args.bfo", lines("params.py", 318)), ("self.BFO         = 700", lines("params.py", 320)),
                       ("self.VIDEO_BW        = args.vid_bw*1e3", lines("params.py", 324)), ("self.VIDEO_BW = 10e3", lines("params.py", 329))):
        assert frag in text, (frag, text)
    rows = []
    for fc in ([7.1e6], [14.074e6, 14.08e6, 14.1e6], [3.5e6, 28.4e6], [7e6] * 6):
        for audio in (1, 2):
            for src in ([], [0], [-1, 0, 0]):
                for mode, bfo, vid, fo in (("CW", 0, 0, 0.0), ("CW", 600, 0, 100e3), ("WFM", 0, 0, 0.0), ("AM", 0, 25, -37500.0), ("USB", 0, 0, 100e3)):
                    self = types.SimpleNamespace(NUM_RX=len(fc), AUDIO_SCHEME=audio, FOFFSET=fo, MODE=mode)
                    args = types.SimpleNamespace(src=list(src)[:len(fc)], bfo=bfo, vid_bw=vid)
                    env = dict(self=self, args=args, np=np, fc=np.array(fc), print=lambda *a, **k: None)
                    exec(src_block, env)
                    exec(players_block, env)
                    exec(fo_block, env)
                    exec(bfo_block, env)
                    exec(vbw_block, env)
                    rows.append(dict(fc=fc, audio=audio, src=list(src)[:len(fc)], mode=mode, bfo=bfo, vid_bw=vid, foffset=fo,
                                     SOURCE=[int(v) for v in self.SOURCE], NUM_PLAYERS=self.NUM_PLAYERS, FOFFSET=float(self.FOFFSET),
                                     BFO=float(self.BFO), VIDEO_BW=float(self.VIDEO_BW)))
    out["params"] = rows
    # ---- receiver.py: create_Receivers offsets, mode_freq_change (mode part)
    cr = func("receiver.py", "create_Receivers", "SDR_EXECUTIVE")
    assert (cr.lineno, cr.end_lineno) == (826, 835)
    mf = func("receiver.py", "mode_freq_change", "SDR_EXECUTIVE")
    assert mf.lineno == 633
    made = []
    dsp = types.SimpleNamespace(Receiver=lambda P, frq, irx, name, vb, ab: made.append((float(frq), irx, name)) or ("rx", irx))
    rns = dict(np=np, dsp=dsp, VIDEO_BWs=ns["VIDEO_BWs"], AF_BWs=ns["AF_BWs"], print=lambda *a, **k: None)
    exec(compile(ast.Module([cr, mf], []), "receiver.py:633-835", "exec"), rns)
    offs = []
    for fc, source, fo in (([7.1e6], [-1], 100e3), ([14.074e6, 14.08e6, 14.1e6], [-1, -1, -1], -13000.0),
                           ([14.074e6, 14.08e6, 14.1e6], [-1, 0, 0], 5000.0), ([3.5e6, 3.6e6], [-1, 0], 0.0)):
        del made[:]
        P = types.SimpleNamespace(FOFFSET=fo, NUM_RX=len(fc), SOURCE=np.array(source), FC=np.array(fc), rx=[None] * len(fc))
        rns["create_Receivers"](types.SimpleNamespace(P=P))
        offs.append(dict(fc=fc, source=source, foffset=fo, frq=[m[0] for m in made], names=[m[2] for m in made]))
    out["create_Receivers"] = offs
    modes = []
    for old, new, scheme in (("AM", "FM", 1), ("USB", "USB", 1), ("CW", "AM-Synch", 2), ("AM", "NFM", 3)):
        calls = []
        rx0 = types.SimpleNamespace(agc=types.SimpleNamespace(reset=lambda: calls.append("agc")),
                                    demod=types.SimpleNamespace(am_pll=types.SimpleNamespace(reset=lambda: calls.append("pll"))))
        P = types.SimpleNamespace(MODE_CHANGE=True, MODE=old, NEW_MODE=new, MP_SCHEME=scheme, FREQ_CHANGE=False, SDR_TYPE='sdrplay',
                                  rx=[rx0], gui=types.SimpleNamespace(ModeSelect=lambda v: calls.append("gui")))
        rns["mode_freq_change"](types.SimpleNamespace(P=P))
        modes.append(dict(old=old, new=new, mp_scheme=scheme, MODE=P.MODE, NEW_MODE=P.NEW_MODE, MODE_CHANGE=bool(P.MODE_CHANGE),
                          resets=[c for c in calls if c != "gui"]))
    out["mode_change"] = modes
    json.dump(out, open(os.path.join(HERE, "host_misc_ref.json"), "w"), indent=0)
    print(len(rows), "param rows;", len(offs), "receiver sets;", len(modes), "mode changes")


if __name__ == "__main__":
    main()
