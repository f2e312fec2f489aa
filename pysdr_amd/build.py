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
SOURCES = ["api.hip", "mixdec.hip", "mixdec_mfma.hip", "resamp_small.hip", "stage2.hip", "misc.hip", "psdfft.hip", "waterfall.hip"]
# per-file extra flags (none needed today; -fno-slp-vectorize on mixdec.hip folds the DPP
# reduction into v_add_f32_dpp but measured the same 96-99 us, so the default stays)
EXTRA_FLAGS = {"mixdec.hip": os.environ.get("PYSDR_MIXDEC_FLAGS", "").split(),
               "mixdec_mfma.hip": os.environ.get("PYSDR_MFMA_FLAGS", "").split(),
               "resamp_small.hip": os.environ.get("PYSDR_RESAMP_FLAGS", "").split(),
               "api.hip": os.environ.get("PYSDR_API_FLAGS", "").split(),   # a shape's S / NB enter the host's plan: pass the same -D to both   # experiments: -DMM_NO_PK, -DMM_PROD_PRIO=n
               # packed-f32 pairs built by the SLP vectoriser run at half rate on gfx950 and are fed by
               # v_mov shuffles: the AF FIR is written for plain FMAs
               "stage2.hip": ["-fno-slp-vectorize"] + os.environ.get("PYSDR_STAGE2_FLAGS", "").split(),
               # FFT butterflies are adds: v_pk_add_f32 issues at 5.6 cycles against 2 x 3.1 for two plain adds and
               # hipcc pays for the pairing with v_mov shuffles and 17 more registers (A/B: PYSDR_PSD_FLAGS)
               # measured on C3, PSD ms per 10666 frames: default 2.752 / 2.774, -fno-slp-vectorize 2.731 / 2.731,
               # + -fno-signed-zeros (lets the zero-padded half of the first DFT16 fold away) 2.721: the pair is
               # fabric-bound, 17 % fewer issue cycles buy 1.5 %
               "psdfft.hip": os.environ.get("PYSDR_PSD_FLAGS", "-fno-slp-vectorize -fno-signed-zeros").split()}
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newest_source_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files.append(os.path.join(HERE, "..", "include", "pysdr_hip.h"))
    return max(os.path.getmtime(f) for f in files)


def needs_build():
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < _newest_source_mtime()


def build(force=False, verbose=True, diag=False):
    """``diag=True`` (``--diag``) compiles the work-skipping ablation switches of the mix+decimate
    kernel in (-DPYSDR_DIAG, read from PYSDR_DEBUG_FLAGS); the default build has none."""
    if not force and not diag and not needs_build():
        return LIB
    lib_out = LIB_DIAG if diag else LIB
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".diag.o" if diag else ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
               "-Wno-unused-function", "-ffp-contract=off", *(["-DPYSDR_DIAG"] if diag else []),
               *EXTRA_FLAGS.get(src, []),
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
    print("built", build(force="--force" in sys.argv, diag="--diag" in sys.argv))
