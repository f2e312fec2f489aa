"""Generate tests/golden/rtty_ref.npz by EXECUTING the reference's own RTTY filterbank text (build
container only: /root/reference does not travel).

Taken from /root/reference/rtty.py as they stand: class RTTY_Params (:376-404, extracted with `ast`)
and, by line number, the statements of RTTY_EXEC.run that make one waterfall line -- :807 the window,
:831 `x = concatenate((prev, iq))`, :837/:839/:841/:843 slice, windowed zero-padded FFT + fftshift,
10*log10(re^2+im^2), flipud.  Two names resolve outside the tree and are supplied here: `nextpow2`
(module `utilities` of aa2il/libs, absent; the usual ceil(log2(n)): N = 1056 -> NFFT = 2048) and the
module-level list `mark_bins` (:58-72, decoder placement only, not used by the lines).  The symbols
are pulled N samples at a time as `self.rb.pull(self.N)` does (:825).  The fixture holds the input
and the lines (float64): data, none of the reference's text.

    python tests/golden/make_rtty_ref_golden.py
"""
import ast
import math
import os
import sys
import textwrap
import types

import numpy as np

REF = "/root/reference/rtty.py"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    src = open(REF).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RTTY_Params"][0]
    ns = dict(np=np, nextpow2=lambda n: int(math.ceil(math.log2(n))), mark_bins=[860], print=lambda *a, **k: None)
    exec(compile(ast.Module([cls], []), "rtty.py:376-404", "exec"), ns)
    rtty = ns["RTTY_Params"](48000)
    lines_src = src.splitlines()
    pick = lambda *nums: textwrap.dedent("\n".join(lines_src[i - 1] for i in nums))
    win_stmt, cat_stmt, line_stmts = pick(807), pick(831), pick(837, 839, 841, 843)
    for frag, text in (("np.kaiser(self.N,8.6)", win_stmt), ("np.concatenate( (prev , iq) )", cat_stmt),
                       ("self.NSTART[i]", line_stmts), ("np.fft.fftshift( np.fft.fft(xx * self.window , self.NFFT) )", line_stmts),
                       ("10*np.log10( np.square(X.real) + np.square(X.imag) )", line_stmts), ("np.flipud(XX)", line_stmts)):
        assert frag in text, frag
    self = types.SimpleNamespace(N=rtty.N, NSTART=rtty.NSTART, NFFT=rtty.NFFT)       # run() :784-787
    self.line = np.zeros((1, self.NFFT))                                             # :803
    exec(compile(win_stmt, "rtty.py:807", "exec"), dict(self=self, np=np))
    from oracle import rtty_oracle as ro                                            # input generator only
    x_all, bits = ro.synth_rtty(48000, 7, 1500.0, seed=6, noise=2e-3)
    out, prev = [], None
    for s in range(len(x_all) // self.N):
        iq = x_all[s * self.N:(s + 1) * self.N]                                      # self.rb.pull(self.N) :825
        if prev is None:
            prev = iq                                                                # :826-829
            continue
        env = dict(self=self, np=np, prev=prev, iq=iq)
        exec(compile(cat_stmt, "rtty.py:831", "exec"), env)
        for i in range(4):                                                           # :834
            env["i"] = i
            exec(compile(line_stmts, "rtty.py:837-843", "exec"), env)
            out.append(self.line[0].copy())
        prev = iq                                                                    # :856
    lines = np.stack(out)
    np.savez_compressed(os.path.join(HERE, "rtty_ref.npz"), x=x_all, lines=lines,
                        params=np.array([rtty.N, rtty.NFFT, rtty.NBINS, rtty.M] + list(rtty.NSTART), np.int64))
    print(lines.shape, lines.dtype, os.path.getsize(os.path.join(HERE, "rtty_ref.npz")))


if __name__ == "__main__":
    main()
