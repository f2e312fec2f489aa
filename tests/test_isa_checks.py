"""Build-time checks on the generated ISA (CPU: hipcc cross-compiles for gfx950 without a GPU).

ADVICE r3: the hand-rolled asynchronous loads of the serial-PLL kernels (stage2.hip: am_pll_lanes_kernel,
wfm_pll_walk) were `asm volatile("global_load_dword[x2] %0, ...")` with the `s_waitcnt vmcnt(0)` in a separate, later asm
statement (rounds 3-5; gone since: see test_nothing_touches_...).  The hardware does not interlock on vmcnt: any instruction the compiler places between the load and the wait
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


def _functions(lines):
    """[(symbol, [(line number, instruction text, inside inline asm)])] for every function of an assembly listing"""
    out, cur, name, in_asm = [], None, None, False
    for no, raw in enumerate(lines, 1):
        ln = raw.strip()
        m = re.fullmatch(r"(_Z\w+):", ln.split(";")[0].strip())
        if m and cur is None:
            name, cur, in_asm = m.group(1), [], False
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            out.append((name, cur))
            cur = None
            continue
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = ln.split(";")[0].strip()
        if not code or (code.startswith(".") and not code.endswith(":")):
            continue
        cur.append((no, code, in_asm))
    return out


def _pending_load_violations(insns):
    """Forward data flow over the control-flow graph of ONE function (round 5: the walks keep several blocks of loads in
    flight across a loop's back edge, which a scan in layout order does not follow).  State = the asm loads that may
    still be in flight, in issue order, each with its destination registers.  `s_waitcnt vmcnt(N)` -- hand-placed or
    the compiler's -- leaves at most N operations outstanding; loads return in order, so all but the N newest load
    instructions have landed (stores in between, which gfx9 does not order against loads, can only mean that fewer
    loads than N are outstanding).  Any vector / memory instruction that names a register of a load still in flight is a
    violation: the hardware does not interlock on vmcnt, and a register the compiler believes free while a load is
    about to land in it is how a wild address is made.  Returns (violations, loads seen waited for)."""
    # basic blocks
    blocks, label_at, cur = [], {}, []
    for no, code, in_asm in insns:
        if code.endswith(":"):
            if cur:
                blocks.append(cur)
                cur = []
            label_at[code[:-1]] = len(blocks)
            continue
        cur.append((no, code, in_asm))
        op = code.split(" ")[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64") or op.startswith("s_cbranch"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, blk in enumerate(blocks):
        no, code, _ = blk[-1] if blk else (0, "", False)
        op, _, rest = code.partition(" ")
        if op == "s_branch":
            succ.append([label_at[rest.strip()]])
        elif op.startswith("s_cbranch"):
            succ.append([label_at[rest.strip()]] + ([i + 1] if i + 1 < len(blocks) else []))
        elif op in ("s_endpgm", "s_setpc_b64"):
            succ.append([])
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])
    violations, waited = [], set()
    seen = [set() for _ in blocks]
    work = [(0, ())]
    while work:
        bi, state = work.pop()
        if bi >= len(blocks) or state in seen[bi]:
            continue
        seen[bi].add(state)
        assert len(seen[bi]) <= 256, "state explosion: a loop issues asm loads without ever waiting for them?"
        st = list(state)
        for no, code, in_asm in blocks[bi]:
            op, _, rest = code.partition(" ")
            toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest) if t.strip()]
            m = re.search(r"vmcnt\((\d+)\)", rest) if op == "s_waitcnt" else None
            if m:
                n = int(m.group(1))
                for ld in st[:max(0, len(st) - n)]:
                    waited.add(ld[0])
                st = st[max(0, len(st) - n):]
                continue
            if op.startswith(("v_", "global_", "ds_", "buffer_", "flat_", "scratch_")):
                used = set()
                for t in toks:
                    used |= _regs(t.split(" ")[0])
                pend = {r: ld[0] for ld in st for r in ld[1]}
                hit = used & set(pend)
                if hit:
                    violations.append((no, code, sorted(hit), pend[sorted(hit)[0]]))
                if in_asm and op.startswith("global_load_dword"):
                    st = [ld for ld in st if ld[0] != no]     # (an older instance of the same load is older than everything kept)
                    st.append((no, tuple(sorted(_regs(toks[0])))))
        for sx in succ[bi]:
            work.append((sx, tuple(st)))
    return violations, len(waited)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_nothing_touches_an_asm_load_destination_before_its_wait():
    """(Since the end of round 5 the PLL walks prefetch with plain loads a group of blocks ahead and hipcc places the waits;
    the check stays for whatever inline-asm load the stage-2 sources grow next, and its own test below keeps it honest.)"""
    bad, nfun = [], 0
    for src in ("stage2.hip", "pllseed.hip", "resamp_small.hip"):
        for name, insns in _functions(_isa(src, ["-fno-slp-vectorize"])):
            nfun += 1
            v, _ = _pending_load_violations(insns)
            bad += [(src, name[:60]) + x for x in v]
    assert nfun >= 12, nfun
    assert not bad, "an instruction touches the destination of an asm load still in flight:\n" + \
        "\n".join(f"  {f} {n}: line {no} `{code}` touches v{hit} (load of line {at})" for f, n, no, code, hit, at in bad[:12])


def test_the_pending_load_check_finds_a_copy_on_a_back_edge():
    """the checker itself: a loop that copies an in-flight register at its bottom (what hipcc does to a loop-carried value
    whose two definitions got different registers) is found although no fall-through path leads from the load to the copy"""
    fn = """
_Z4testv:
	v_mov_b32_e32 v1, 0
.LBB0_1:
	;;#ASMSTART
	global_load_dword v2, v[4:5], off
	;;#ASMEND
	s_cbranch_scc1 .LBB0_3
	s_branch .LBB0_2
.LBB0_3:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
.LBB0_2:
	v_mov_b32_e32 v3, v2
	s_branch .LBB0_1
.Lfunc_end0:
"""
    (name, insns), = _functions(fn.splitlines())
    v, n = _pending_load_violations(insns)
    assert 16 in [x[0] for x in v] and n == 1, (v, n)       # (+ the load itself, re-issued into a register still in flight)
    ok = fn.replace("\tv_mov_b32_e32 v3, v2\n", "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v3, v2\n")
    (name, insns), = _functions(ok.splitlines())
    assert _pending_load_violations(insns)[0] == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_shipped_kernel_uses_scratch():
    """A spilled register is a hidden wait for the LDS-DMA (its reload's s_waitcnt vmcnt(0) also waits for the NEXT tile's
    copies: DESIGN.md 4.1, round 3): no kernel of the shipped library may touch private memory, with the flags
    pysdr_amd/build.py uses.  What is checked is the hazard itself -- no `scratch_*` / private `buffer_*` instruction in any
    kernel, no spilled vector register -- and, for every kernel but the ones named below, an empty stack frame as well.  The
    matrix-core shapes of mixdec.hip (round 6) are AT the scalar-register limit: hipcc gives them a frame of 36 bytes for
    scalar-spill slots (36 - 68 bytes) that it then serves from VGPR lanes (v_writelane / v_readlane; `-Rpass-analysis=stack-frame-layout`:
    "Spill, Size: 32" + one variable) without emitting a single memory instruction for it."""
    from concurrent.futures import ThreadPoolExecutor
    from pysdr_amd import build as pb                                        # SOURCES, EXTRA_FLAGS: importing builds nothing
    files = [f for f in pb.SOURCES if f != "api.hip"]                        # api.hip holds no device code

    def one(f):
        return f, _isa(f, ["-fPIC", *pb.EXTRA_FLAGS.get(f, [])])

    def lane_served(fname, kname):
        return fname == "mixdec.hip" and re.search(r"mixdec_kernelILi\d+ELi(21|11)ELi(768|512)ELi\d+ELi1E", kname or "") is not None

    bad, nk, framed = [], 0, []
    with ThreadPoolExecutor(max_workers=4) as ex:
        for f, lines in ex.map(one, files):
            for sym, insns in _functions(lines):
                for no, text, in_asm in insns:
                    op = text.split()[0] if text.split() else ""
                    if op.startswith("scratch_") or (op.startswith("buffer_") and " off" not in text and "s[0:3]" in text):
                        bad.append((f, sym, no, text))
            name = None
            for ln in lines:
                m = re.search(r"\.name:\s+(\S+)", ln)
                if m:
                    name = m.group(1)
                m = re.search(r"\.vgpr_spill_count:\s+(\d+)", ln)
                if m and int(m.group(1)) != 0:
                    bad.append((f, name, "vgpr_spill_count", int(m.group(1))))
                m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
                if m:
                    nk += 1
                    if int(m.group(1)) != 0:
                        if lane_served(f, name) and int(m.group(1)) <= 128:
                            framed.append((name, int(m.group(1))))
                        else:
                            bad.append((f, name, int(m.group(1))))
    assert nk >= 40, nk
    assert not bad, bad
    assert len(framed) <= 10, framed
