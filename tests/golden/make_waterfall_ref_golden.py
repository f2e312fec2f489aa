"""Generate tests/golden/waterfall_ref.npz by EXECUTING the numeric statements of the reference's
waterfall display (build container only: /root/reference does not travel).

From /root/reference/Plotting.py as they stand: :385-388 (history, counters, line buffer), the
statements of three_box_plot.plot at :536-540, :543, :547-548 (push), :583, :587 (background), :594,
:596 (peak pick), :610, :618-619, :625-626 (image), picked by line number, and the method
shift_waterfall (:689-695, extracted with `ast`).  `self`, `P` and `self.psd` are attribute bags;
`self.imager.imagesc` records its first argument (the image the reference hands to Qt).  Every name
the statements use resolves to NumPy / SciPy or to those bags.  The fixture holds the pushed lines and
what the reference computed from them: data, none of its text.

    python tests/golden/make_waterfall_ref_golden.py
"""
import ast
import os
import textwrap
import types

import numpy as np
from scipy import signal

REF = "/root/reference/Plotting.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read()
    L = src.splitlines()
    pick = lambda *nums: textwrap.dedent("\n".join(L[i - 1] for i in nums))
    init = pick(385, 386, 387, 388)
    push = pick(536, 537, 538, 539, 540) + "\n" + pick(543) + "\n" + pick(547, 548)
    image = pick(583) + "\n" + pick(587) + "\n" + pick(594) + "\n" + pick(596) + "\n" + pick(610) + "\n" + pick(618, 619) + "\n" + pick(625, 626)
    for frag, text in (("-1e38*np.ones( (nfft,100) )", init), ("self.line = -1e38*np.ones( (nfft,1) )", init),
                       ("npsd=len(PSD)", push), ("np.flipud(PSD)", push), ("self.line[npsd:,0]=-1e38", push),
                       ("np.concatenate( (self.wf[:,1:self.wf.shape[1]],self.line),axis=1 )", push), ("self.wf_cnt += 1", push),
                       ("PSD2=np.mean(self.wf[:,-self.wf_cnt:],1)", image), ("bkgnd = np.median(PSD2)", image),
                       ("signal.find_peaks(PSD2,distance=dist,height=bkgnd+10)", image), ("zz = self.wf[0:npsd,:] - med", image),
                       ("zmax = np.nanmax(zz)", image), ("np.maximum(zz,zmax-P.PAN_DR)", image)):
        assert frag in text, frag
    tree = ast.parse(src)
    shift = None
    for n in ast.walk(tree):
        if isinstance(n, ast.FunctionDef) and n.name == "shift_waterfall":
            shift = n
    ns = dict(np=np, print=lambda *a, **k: None)
    exec(compile(ast.Module([shift], []), "Plotting.py:689-695", "exec"), ns)
    shift_waterfall = ns["shift_waterfall"]

    nfft, df = 512, 0.125
    rng = np.random.default_rng(16)
    shown = []
    self = types.SimpleNamespace()
    self.psd = types.SimpleNamespace(frq=(np.arange(nfft) - nfft // 2) * df, df=df)
    self.imager = types.SimpleNamespace(imagesc=lambda img, **kw: shown.append(np.array(img)))
    P = types.SimpleNamespace(RIG_IF=0, PAN_DR=60.0, PEAK_DIST=2.0)
    self.P = P
    exec(compile(init, "Plotting.py:385-388", "exec"), dict(self=self, np=np, nfft=nfft))
    c_push = compile(push, "Plotting.py:536-548", "exec")
    c_img = compile(image, "Plotting.py:583-626", "exec")
    lines, lens, retune, snaps = [], [], {}, {}
    fc = 0.0
    for k in range(130):                                  # more than 100 columns: the history wraps
        line = (rng.standard_normal(nfft) * 3 - 90).astype(np.float32)
        line[100 + k] += 40                               # a drifting carrier
        n = nfft if k % 7 else nfft // 2                  # real-input PSDs are half length
        if k in (40, 90):
            fc = fc + (3 if k == 40 else -5) * df
            shift_waterfall(self, fc)
            retune[k] = fc
        P.RIG_IF = -1 if k == 60 else 0                   # one flipped line (:537-538)
        env = dict(self=self, P=P, np=np, signal=signal, PSD=line[:n].astype(np.float64), f1=0, f2=1, frq=self.psd.frq, force=False)
        exec(c_push, env)
        lines.append(line)
        lens.append(n)
        if k in (0, 5, 99, 129):
            env["PSD"] = line[:n].astype(np.float64)
            exec(c_img, env)
            snaps[k] = (shown[-1], float(env["bkgnd"]), np.array(env["PSD2"]), np.array(env["peaks"]), int(self.wf_cnt))
    out = dict(lines=np.stack(lines), lens=np.array(lens), retune_k=np.array(sorted(retune)), retune_fc=np.array([retune[k] for k in sorted(retune)]),
               flip_k=np.array([60]), df=df, pan_dr=P.PAN_DR, peak_dist=P.PEAK_DIST, snap_k=np.array(sorted(snaps)))
    for k, (img, bk, psd2, peaks, cnt) in snaps.items():
        out[f"img{k}"] = img.astype(np.float32)
        out[f"bk{k}"] = bk
        out[f"psd2_{k}"] = psd2
        out[f"peaks{k}"] = peaks
        out[f"cnt{k}"] = cnt
    np.savez_compressed(os.path.join(HERE, "waterfall_ref.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "waterfall_ref.npz")), {k: v[0].shape for k, v in snaps.items()})


if __name__ == "__main__":
    main()
