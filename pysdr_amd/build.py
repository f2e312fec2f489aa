"""Build libpysdr_hip.so in-tree with hipcc for gfx950 (MI355X).

    python -m pysdr_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpysdr_hip.so")
LIB_DIAG = os.path.join(HERE, "libpysdr_hip_diag.so")   # loaded only when PYSDR_USE_DIAG_LIB=1
SOURCES = ["api.hip", "mixdec.hip", "mixdec_mfma.hip", "resamp_small.hip", "stage2.hip", "pllseed.hip", "misc.hip", "psdfft.hip", "waterfall.hip"]
# Flags every build uses (measured choices, part of the shipped configuration):
#   stage2.hip   -fno-slp-vectorize: packed-f32 pairs built by the SLP vectoriser run at half rate on gfx950 and are fed
#                by v_mov shuffles; the AF FIR is written for plain FMAs with its own v_pk_fma_f32
#   psdfft.hip   FFT butterflies are adds: v_pk_add_f32 issues at 5.6 cycles against 2 x 3.1 for two plain adds and hipcc
#                pays for the pairing with v_mov shuffles and 17 more registers; -fno-signed-zeros lets the zero-padded
#                half of the first DFT16 fold away (C3, PSD ms per 10666 frames: default 2.752 / 2.774, -fno-slp-vectorize
#                2.731, + -fno-signed-zeros 2.721: the pair is fabric-bound, 17 % fewer issue cycles buy 1.5 %)
BASE_FLAGS = {"stage2.hip": ["-fno-slp-vectorize"],
              "psdfft.hip": ["-fno-slp-vectorize", "-fno-signed-zeros"]}
# Extra flags for A/B runs (-DMM_C1_NBUF=3, -DFIRX_THREADS=128, ...), one variable per source file.  They are read ONLY
# under PYSDR_TUNING=1 (the same master switch the library's run-time tuning variables obey) or for the diagnostic build:
# an ambient variable cannot change the shipped library.  PYSDR_PSD_FLAGS REPLACES psdfft.hip's base flags (its A/B is about them).
# api.hip gets -DPYSDR_EXTRA_FLAGS_HASH=<31-bit hash of every extra flag> (pysdr_build_flags_hash, echoed by bench.py), and
# the work-skipping ablation switches compile only with -DPYSDR_ABLATE, which only --diag defines (common.h).
FLAG_VARS = {"mixdec.hip": "PYSDR_MIXDEC_FLAGS", "mixdec_mfma.hip": "PYSDR_MFMA_FLAGS", "resamp_small.hip": "PYSDR_RESAMP_FLAGS",
             "api.hip": "PYSDR_API_FLAGS",        # a shape's S / NB enter the host's plan: pass the same -D to both
             "stage2.hip": "PYSDR_STAGE2_FLAGS", "pllseed.hip": "PYSDR_SEED_FLAGS", "psdfft.hip": "PYSDR_PSD_FLAGS"}


def extra_flags(diag=False):
    """-> ({source: [flags]}, hash): what the environment adds to this build (nothing unless PYSDR_TUNING=1 or --diag)."""
    import hashlib
    allowed = diag or os.environ.get("PYSDR_TUNING", "0") not in ("", "0")
    out, seen, ignored = {}, [], []
    for src, var in FLAG_VARS.items():
        v = os.environ.get(var, "").split()
        if v and not allowed:
            ignored.append(var)
            v = []
        out[src] = v
        seen += [f"{src}:{f}" for f in v]
    if ignored:
        print(f"pysdr_amd.build: ignoring {', '.join(ignored)} (extra compiler flags are read only under PYSDR_TUNING=1 or --diag)",
              file=sys.stderr, flush=True)
    if diag:
        seen.append("*:-DPYSDR_DIAG -DPYSDR_ABLATE")
    h = (int(hashlib.sha256(" ".join(sorted(seen)).encode()).hexdigest()[:8], 16) & 0x7fffffff) if seen else 0
    return out, (h or 1) if seen else 0


def flags_for(src, extra):
    if src == "psdfft.hip" and extra.get(src):
        return list(extra[src])
    return BASE_FLAGS.get(src, []) + extra.get(src, [])


# (tests/test_isa_checks.py compiles with these: the flags of the shipped build)
EXTRA_FLAGS = {src: flags_for(src, {}) for src in SOURCES}
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newest_source_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "pysdr_hip.h"))
    return max(os.path.getmtime(f) for f in files)


def needs_build():
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < _newest_source_mtime()


def build_variant(name, verbose=True):
    """A/B builds without a hipcc on the critical path of a GPU call: ``PYSDR_TUNING=1 PYSDR_MIXDEC_FLAGS=-DMD_LONG_TPB=512
    python -m pysdr_amd.build --variant tpb512`` compiles ONLY the sources that got extra flags (into ``*.<name>.o``), links
    them with the shipped objects of the others into ``libpysdr_hip_<name>.so``; ``PYSDR_TUNING=1 PYSDR_LIB_VARIANT=<name>``
    loads it (``_lib.py``).  Variant libraries are never loaded otherwise and are git-ignored like every ``.so``."""
    build(force=False, verbose=verbose)
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    extra, fhash = extra_flags(False)
    objs, procs = [], []
    for src in SOURCES:
        if extra.get(src) or (src == "api.hip" and fhash):
            obj = os.path.join(CSRC, src.replace(".hip", f".{name}.o"))
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-ffp-contract=off",
                   *([f"-DPYSDR_EXTRA_FLAGS_HASH={fhash}"] if (src == "api.hip" and fhash) else []),
                   *flags_for(src, extra), "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    out = os.path.join(HERE, f"libpysdr_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs +
                          ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-ldl", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    return out


def build(force=False, verbose=True, diag=False):
    """``diag=True`` (``--diag``) compiles the work-skipping ablation switches in (-DPYSDR_DIAG: the run-time ones of the
    mix+decimate kernel, read from PYSDR_DEBUG_FLAGS; -DPYSDR_ABLATE: the compile-time ones, common.h) into
    libpysdr_hip_diag.so; the default build has none and refuses them."""
    if not force and not diag and not needs_build():
        return LIB
    lib_out = LIB_DIAG if diag else LIB
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    extra, fhash = extra_flags(diag)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".diag.o" if diag else ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
               "-Wno-unused-function", "-ffp-contract=off", *(["-DPYSDR_DIAG", "-DPYSDR_ABLATE"] if diag else []),
               *([f"-DPYSDR_EXTRA_FLAGS_HASH={fhash}"] if (src == "api.hip" and fhash) else []),
               *flags_for(src, extra),
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_out] + objs + \
          ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-ldl",
           "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        print("built", build_variant(sys.argv[sys.argv.index("--variant") + 1]))
    else:
        print("built", build(force="--force" in sys.argv, diag="--diag" in sys.argv))
