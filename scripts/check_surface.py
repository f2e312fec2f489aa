#!/usr/bin/env python3
"""Pin the drop-in boundary to the reference TEXT (not to SURVEY.md's reading of it).

Build-container only: parses /root/reference/{receiver,gui,Plotting,watchdog,utils,pySDR,am,mp,
params,srates,rtty}.py with `ast` -- nothing of the reference is imported or executed -- and
collects every use of the absent `sig_proc` module:

  * `dsp.<name>` / `from sig_proc import <name>`                      -> module attributes
  * attribute chains rooted at a sub-receiver (`rx.`, `P.rx[..].`, `self.rx.` ...)   -> Receiver
  * `<x>.psd.<attr>` / `self.psd.<attr>`                                -> spectrum
  * methods called on ring buffers (`rb`, `rb_rf`, `rb_af`, `rb_baseband`, `.rb.`)   -> ring_buffer2/3
  * `P.lo.<attr>` / `self.lo.<attr>`                                    -> signal_generator
  * `<x>.convolve_fast` on objects built by dsp.convolver               -> convolver

and checks each against the objects `pysdr_amd.sig_proc` provides (structurally: class attributes,
properties and the attributes their __init__ assigns -- no GPU needed).  Prints the table; exit 1
on a missing name.  tests/test_surface.py runs it when /root/reference exists."""
from __future__ import annotations

import ast
import os
import sys

REF = os.environ.get("PYSDR_REFERENCE", "/root/reference")
FILES = ["receiver.py", "gui.py", "Plotting.py", "watchdog.py", "utils.py", "pySDR.py", "am.py", "mp.py",
         "params.py", "srates.py", "rtty.py", "hopper.py", "udp.py", "sigs/iq.py"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RB_NAMES = {"rb", "rb_rf", "rb_af", "rb_baseband", "rb_audio"}
RB_ATTRS_CONTAINER = {"buf"}          # rb.buf is a queue: .qsize()/.put() belong to it, not to us


def chain(node):
    """['P', 'rx', '[]', 'lo', 'change_freq'] for P.rx[i].lo.change_freq"""
    out = []
    while True:
        if isinstance(node, ast.Attribute):
            out.append(node.attr)
            node = node.value
        elif isinstance(node, ast.Subscript):
            out.append("[]")
            node = node.value
        elif isinstance(node, ast.Call):
            out.append("()")
            node = node.func
        elif isinstance(node, ast.Name):
            out.append(node.id)
            break
        else:
            out.append("?")
            break
    return out[::-1]


def collect():
    uses = {"module": {}, "Receiver": {}, "spectrum": {}, "ring_buffer": {}, "signal_generator": {}, "convolver": {}}

    def note(kind, name, where):
        uses[kind].setdefault(name, []).append(where)

    for fn in FILES:
        path = os.path.join(REF, fn)
        if not os.path.exists(path):
            continue
        tree = ast.parse(open(path, encoding="utf-8", errors="replace").read(), filename=fn)
        # one level of aliasing: `agc = P.rx[0].agc` ... `agc.gain` (watchdog.py:298-302);
        # `psd = dsp.spectrum(...)` ... `psd.frq2` (sigs/iq.py:75-78)
        alias = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                c = [p for p in chain(node.value) if p not in ("[]", "()")]
                if "rx" in c[:-1]:
                    alias[node.targets[0].id] = ("Receiver", ".".join(c[len(c) - c[::-1].index("rx"):]))
                elif len(c) == 2 and c[0] == "dsp" and c[1] == "spectrum":
                    alias[node.targets[0].id] = ("spectrum", "")
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in alias \
                    and node.value.id not in ("rx", "self", "P"):
                kind, prefix = alias[node.value.id]
                note(kind, (prefix + "." if prefix else "") + node.attr, f"{fn}:{node.lineno}")
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module == "sig_proc":
                for a in node.names:
                    note("module", a.name, f"{fn}:{node.lineno}")
            if not isinstance(node, ast.Attribute):
                continue
            c = [p for p in chain(node) if p not in ("[]", "()")]
            where = f"{fn}:{node.lineno}"
            if len(c) >= 2 and c[0] == "dsp" and node.attr == c[1] and len(c) == 2:
                note("module", c[1], where)
            # sub-receiver chains: ... rx . a . b ...
            if "rx" in c[:-1]:
                i = len(c) - 1 - c[::-1].index("rx")            # last 'rx' in the chain
                tail = c[i + 1:]
                if tail and node.attr == tail[-1]:
                    note("Receiver", ".".join(tail), where)
            if "psd" in c[:-1]:
                i = len(c) - 1 - c[::-1].index("psd")
                tail = c[i + 1:]
                if len(tail) == 1 and node.attr == tail[0]:
                    note("spectrum", tail[0], where)
            for j, name in enumerate(c[:-1]):
                if name in RB_NAMES and j == len(c) - 2 and node.attr == c[-1]:
                    note("ring_buffer", c[-1], where)
            if len(c) >= 2 and c[-2] == "lo" and "rx" not in c and node.attr == c[-1]:
                note("signal_generator", c[-1], where)
            if node.attr == "convolve_fast":
                note("convolver", "convolve_fast", where)
    return uses


def structural_attrs(cls):
    """names a class provides without instantiating it: class dict (methods, properties) of the
    MRO + every `self.<name> = ...` in any of its methods"""
    import inspect
    names = set()
    for k in cls.__mro__:
        names |= set(vars(k))
        try:
            src = inspect.getsource(k)
        except (OSError, TypeError):
            continue
        import textwrap
        for n in ast.walk(ast.parse(textwrap.dedent(src))):
            if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "self" and \
                    isinstance(n.ctx, ast.Store):
                names.add(n.attr)
    return names


def check():
    from pysdr_amd import sig_proc as sp
    uses = collect()
    missing = []
    rows = []

    def have(kind, name, ok, where):
        rows.append((kind, name, "ok" if ok else "MISSING", ", ".join(where[:3]) + (" ..." if len(where) > 3 else "")))
        if not ok:
            missing.append((kind, name, where))

    for name, where in sorted(uses["module"].items()):
        have("sig_proc", name, hasattr(sp, name), where)
    rx = structural_attrs(sp.Receiver)
    sub = {"lo": structural_attrs(sp._ReceiverLO), "dec": structural_attrs(sp._Decimator),
           "demod": structural_attrs(sp._Demod), "agc": structural_attrs(sp._AGC)}
    subsub = {("demod", "am_pll"): structural_attrs(sp._PLLHandle), ("demod", "wfm_video"): structural_attrs(sp._WfmVideo)}
    NUMPY_OK = {"real", "imag", "copy", "astype", "shape", "dtype", "size"}     # attributes of the ndarrays rx.am / rx.iq
    for name, where in sorted(uses["Receiver"].items()):
        parts = name.split(".")
        ok = parts[0] in rx
        if ok and len(parts) >= 2:
            if parts[0] in sub:
                ok = parts[1] in sub[parts[0]]
                if ok and len(parts) >= 3 and (parts[0], parts[1]) in subsub:
                    ok = parts[2] in subsub[(parts[0], parts[1])]
            elif parts[0] in ("am", "iq"):
                ok = parts[1] in NUMPY_OK
        have("Receiver", name, ok, where)
    spa = structural_attrs(sp.spectrum)
    for name, where in sorted(uses["spectrum"].items()):
        have("spectrum", name, name in spa, where)
    rba = structural_attrs(sp.ring_buffer2) | structural_attrs(sp.ring_buffer3)
    for name, where in sorted(uses["ring_buffer"].items()):
        have("ring_buffer2/3", name, name in rba, where)
    sga = structural_attrs(sp.signal_generator)
    for name, where in sorted(uses["signal_generator"].items()):
        have("signal_generator", name, name in sga, where)
    for name, where in sorted(uses["convolver"].items()):
        have("convolver", name, name in structural_attrs(sp.convolver), where)
    return rows, missing


if __name__ == "__main__":
    if not os.path.isdir(REF):
        print(f"{REF} not present: nothing to check (this script only runs in the build container)")
        sys.exit(0)
    rows, missing = check()
    w = max(len(r[1]) for r in rows)
    for kind, name, st, where in rows:
        print(f"{kind:17s} {name:{w}s} {st:8s} {where}")
    print(f"{len(rows)} names used by the reference, {len(missing)} missing")
    sys.exit(1 if missing else 0)
