"""SURVEY.md 5 "sanitizers on the CPU build": the HOST half of libpysdr_hip.so -- pysdr_amd/csrc/api.hip
(context / receiver bookkeeping, tile geometry handed to the mix + decimate kernel, PLL plans, the setter
snapshot, the ingest ring's slot state machine, the spectrum object) compiled as plain C++ over a fake
HIP runtime whose "device" memory is the host heap, with a launch layer that touches exactly what each
kernel may touch and re-walks every mixdec tile with the kernel's own geometry code
(pysdr_amd/csrc/mixdec_geom.h) -- built and run under AddressSanitizer + UBSan, and the RX-thread-vs-Qt-
thread scenario under ThreadSanitizer.  CPU only; no GPU sanitizer exists on this pool.

Seeded-bug check done when the harness was written: halving the d_am allocation is reported by ASan in
launch_apply (IQ mode writes 2 floats per output), removing the lock of apply_pending by TSan."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tests", "host_san", "run.sh")


def _run(which, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    env = dict(os.environ, HOST_SAN_OUT=str(tmp_path))
    p = subprocess.run(["bash", RUN, which], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if p.returncode != 0 and ("cannot find -lasan" in p.stderr or "cannot find -ltsan" in p.stderr or "unexpected memory mapping" in p.stderr):
        pytest.skip("sanitizer runtime not usable here: " + p.stderr[-200:])
    assert p.returncode == 0 and "HOST_SAN_ALL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    return p.stdout


def test_host_half_under_address_and_ub_sanitizers(tmp_path):
    out = _run("asan", tmp_path)
    assert "HOST_SAN_OK" in out and "HOST_SAN_RACE_OK" in out


def test_setters_against_process_under_thread_sanitizer(tmp_path):
    assert "HOST_SAN_RACE_OK" in _run("tsan", tmp_path)
