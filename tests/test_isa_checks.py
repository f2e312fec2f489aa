"""Build-time checks on the generated ISA (CPU: hipcc cross-compiles for gfx950 without a GPU).

ADVICE r3: the hand-rolled asynchronous loads of the serial-PLL kernels (stage2.hip: am_pll_lanes_kernel,
wfm_pll_walk) are `asm volatile("global_load_dword[x2] %0, ...")` with the `s_waitcnt vmcnt(0)` in a separate, later asm
statement.  The hardware does not interlock on vmcnt: any instruction the compiler places between the load and the wait
that READS the destination register (a merge copy, a spill) sees the register before the data lands.  The sources tie the
register "+v" so that no merge copy is needed; this test looks at what hipcc actually emitted and fails if any instruction
between such a load and the next wait for it touches its destination (along the fall-through path: the scan is in
layout order and gives up at an unconditional branch).  It found one on its first run: `cur = nxt` copies of
am_pll_lanes_kernel hoisted above the wait (fixed by swapping the roles of two register blocks instead of copying)."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")


def _isa(src, flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
               *flags, os.path.join(ROOT, "pysdr_amd", "csrc", src), "-o", out]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return open(out).read().splitlines()


def _regs(tok):
    """registers named by an operand token: v12 -> {12}, v[4:7] -> {4,5,6,7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_nothing_touches_an_asm_load_destination_before_its_wait():
    lines = _isa("stage2.hip", ["-fno-slp-vectorize"])
    in_asm, pending, checked = False, {}, 0          # pending: register -> line number of the load that defines it
    for no, raw in enumerate(lines, 1):
        ln = raw.strip()
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ln or ln.startswith(";") or ln.startswith(".") or ln.endswith(":"):
            continue
        code = ln.split(";")[0].strip()
        op, _, rest = code.partition(" ")
        toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest) if t.strip()]
        if in_asm and op.startswith("global_load_dword"):
            for r in _regs(toks[0]):
                pending[r] = no
            continue
        if op == "s_waitcnt" and "vmcnt(0)" in rest:
            checked += len(pending)
            pending.clear()
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending.clear()                           # layout order stops being execution order (no CFG here: the check
            continue                                  # follows the fall-through path from a load to its wait)
        if pending and op.startswith("s_waitcnt") and "vmcnt" in rest:
            continue                                  # a partial wait: the loads stay pending
        if pending and (op.startswith("v_") or op.startswith("global_") or op.startswith("ds_") or op.startswith("buffer_")
                        or op.startswith("flat_") or op.startswith("scratch_")):
            used = set()
            for t in toks:
                used |= _regs(t.split(" ")[0])
            hit = used & set(pending)
            assert not hit, (f"stage2.hip ISA line {no}: `{code}` touches v{sorted(hit)} between the asm load of line "
                             f"{pending[sorted(hit)[0]]} and its s_waitcnt vmcnt(0)")
    assert checked >= 9, f"expected the PLL kernels' asm prefetches in the ISA, saw {checked} load registers waited for"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_shipped_kernel_uses_scratch():
    """A spilled register is a hidden wait for the LDS-DMA (its reload's s_waitcnt vmcnt(0) also waits for the NEXT tile's
    copies: DESIGN.md 4.1, round 3): every kernel of the shipped library must compile without scratch, with the flags
    pysdr_amd/build.py uses."""
    from concurrent.futures import ThreadPoolExecutor
    from pysdr_amd import build as pb                                        # SOURCES, EXTRA_FLAGS: importing builds nothing
    files = [f for f in pb.SOURCES if f != "api.hip"]                        # api.hip holds no device code

    def one(f):
        return f, _isa(f, ["-fPIC", *pb.EXTRA_FLAGS.get(f, [])])

    bad, nk = [], 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        for f, lines in ex.map(one, files):
            name = None
            for ln in lines:
                m = re.search(r"\.name:\s+(\S+)", ln)
                if m:
                    name = m.group(1)
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
                if m:
                    nk += 1
                    if int(m.group(1)) != 0:
                        bad.append((f, name, int(m.group(1))))
    assert nk >= 40, nk
    assert not bad, bad
