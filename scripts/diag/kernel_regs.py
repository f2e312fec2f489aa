#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one source file, from the ISA hipcc emits for gfx950:
    python scripts/diag/kernel_regs.py mixdec.hip [name filter] [-- extra hipcc flags]
(the shipped build's flags, pysdr_amd/build.py).  Runs without a GPU."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pysdr_amd import build


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [os.path.join(build.ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-Wno-unused-function", *build.flags_for(src, {}), *extra, "--cuda-device-only", "-S",
               os.path.join(build.CSRC, src), "-o", out]
        subprocess.check_call(cmd)
        s = open(out).read()
        if os.environ.get("KEEP_ASM"):
            open(os.environ["KEEP_ASM"], "w").write(s)
    for b in re.findall(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', s, re.S):
        name = re.search(r'\.name:\s+(\S+)', b).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r'pysdr::\(anonymous namespace\)::', '', dem)
        dem = re.sub(r'\(.*', '', dem).replace('void ', '')
        if filt and filt not in dem:
            continue
        g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
        print(f"{dem:60s} vgpr {g('vgpr_count'):>3s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} "
              f"spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} max_wg {g('max_flat_workgroup_size')}")


if __name__ == "__main__":
    main()
