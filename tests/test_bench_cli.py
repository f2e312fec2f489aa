"""bench.py's launch contract on CPU: `--gpus N` without a launcher starts N rank processes (before any
GPU call), a WORLD_SIZE that disagrees with --gpus is refused, the workload table covers every BASELINE
configuration that fits one GPU, and the PMC traffic figure is dropped as soon as the kernel sources no
longer hash to what was profiled."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr and p.stdout.strip() == ""


def test_gpus_n_spawns_n_ranks_before_touching_the_gpu(monkeypatch):
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, cwd=None):
            started.append((cmd, env))

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    assert "pysdr_amd._lib" not in sys.modules or True      # (the parent never needs the library)
    rc = bench.spawn_ranks(bench.parse(["--gpus", "3", "--steps", "2"]))
    assert rc == 0 and len(started) == 3
    ports = set()
    for r, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "3", "--steps", "2"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "3", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1


def test_workloads_cover_the_single_gpu_baseline_configs():
    cfgs = {w: bench.workload_cfg(bench.parse(["--workload", w])) for w in bench.DEFAULT_CHUNKS}
    assert cfgs["c1"]["fs"] == 2.048e6 and cfgs["c1"]["ntaps_dec"] == 1001 and [r["mode"] for r in cfgs["c1"]["rx"]] == ["AM"]
    assert cfgs["c2"]["fs"] == 8e6 and [r["mode"] for r in cfgs["c2"]["rx"]] == ["NFM"] and cfgs["c2"]["ntaps_dec"] == 255
    assert [r["mode"] for r in cfgs["c3"]["rx"]] == ["USB", "CW", "NFM", "AM"]
    assert cfgs["c4"]["fs"] == 10e6 and cfgs["c4"]["rx"][0]["mode"] == "WFM2" and cfgs["c4mono"]["rx"][0]["mode"] == "WFM"
    six = bench.workload_cfg(bench.parse(["--nrx", "6"]))
    assert len(six["rx"]) == 6
    # the reference's own launch scripts (FT8tri:47-74, TEST:13-32) with its default filter length (params.py:134)
    ft8, t2 = cfgs["ft8tri"], cfgs["test2rx"]
    assert ft8["fs"] == 8e6 and ft8["ntaps_dec"] == 1001 and [r["mode"] for r in ft8["rx"]] == ["USB"] * 3
    assert [r["frq"] for r in ft8["rx"]] == [0.0, 2974e3, 6815e3] and all(r["video_bw"] == 45e3 and r["af_bw"] == 5e3 for r in ft8["rx"])
    assert t2["fs"] == 4e6 and t2["ntaps_dec"] == 1001 and [(r["mode"], r["frq"]) for r in t2["rx"]] == [("NFM", 0.0), ("NFM", -600e3)]
    assert bench.workload_cfg(bench.parse(["--workload", "c3", "--ntaps", "1001"]))["ntaps_dec"] == 1001


def test_stale_profile_means_no_traffic_figure(tmp_path, monkeypatch):
    args = bench.parse([])
    real = {s: bench.source_sha(s) for s in ("mixdec.hip", "psdfft.hip")}
    assert all(real.values())
    prof = tmp_path / "profiles"
    prof.mkdir()
    doc = dict(git_head="abc", source_sha256=dict(real), kernels=[dict(kernel="mixdec_kernel", hbm_bytes_per_launch=123.0)])
    (prof / "r02_pmc_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_sha", lambda s: real[s])
    assert bench.measured_traffic(args, 4, 2048, "mixdec", ["mixdec.hip"])[0] == 123.0
    doc["source_sha256"]["mixdec.hip"] = "0000000000000000"
    (prof / "r02_pmc_traffic.json").write_text(json.dumps(doc))
    t, why = bench.measured_traffic(args, 4, 2048, "mixdec", ["mixdec.hip"])
    assert t is None and "stale" in why
    # another configuration than the profiled one never gets the figure
    assert bench.measured_traffic(bench.parse(["--workload", "c2"]), 1, 2048, "mixdec", ["mixdec.hip"]) == (None, None)


def test_stream_slice_is_the_continuous_stream_the_steps_read():
    """--verify rebuilds the input of any stretch of the absolute stream from the loop, the seam and
    the step size: a slice across a step boundary equals the two steps' reads put end to end."""
    nloop, nsamp = 1000, 2348                     # seam 348: step k reads the loop from (k * 348) % 1000
    xu = (np.arange(nloop) + 1j * np.arange(nloop)).astype(np.complex64)
    seam = nsamp % nloop

    def step_input(k):
        off = bench.step_offset(k, seam, nloop)
        return np.resize(np.roll(xu, -off), nsamp)

    whole = np.concatenate([step_input(k) for k in range(4)])
    for s, n in ((0, 10), (2300, 100), (2 * nsamp - 7, 2 * nsamp), (4 * nsamp - 5, 5)):
        assert np.array_equal(bench.stream_slice(xu, nloop, seam, nsamp, s, n), whole[s:s + n])
    # no seam: every step reads the loop from its start
    assert np.array_equal(bench.stream_slice(xu, nloop, 0, 3 * nloop, 3 * nloop - 2, 4), np.r_[xu[-2:], xu[:2]])


@pytest.mark.parametrize("which", ["c2", "c3"])
def test_primed_oracle_joins_the_stream_where_a_full_run_is(which):
    """The checker of --verify starts its oracle a few chunks in front of what it compares, with the
    absolute counters (LO phase, resampler index, output index / BFO phase) set as if it had run
    from sample 0: after the filters have filled it must give what the full run gives."""
    from oracle import sdr_oracle as so
    args = bench.parse(["--workload", which])
    cfg = bench.workload_cfg(args)
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    n, k0 = 6, 3
    x = bench.synth_batch(cfg, n * L, 21)
    full = bench.oracle_receivers(cfg)
    late = bench.primed_oracle(cfg, None, k0 * L)
    for k in range(n):
        xc = x[k * L:(k + 1) * L]
        for f, o in zip(full, late):
            af = f.demod_data(xc)
            if k < k0:
                continue
            ao = o.demod_data(xc)
            assert len(af) == len(ao) and f.dec.n_abs == o.dec.n_abs and f.demod.m_abs == o.demod.m_abs
            if k >= k0 + 1:                       # one chunk for the FIR histories to fill
                assert bench._relerr(o.iq, f.iq) <= 1e-6, (which, f.mode, k)
                # the AGC's memory is longer than this test (--verify primes 192 chunks): compare the audio per unit gain
                gf, go = float(f.agc.gain), float(o.agc.gain)
                assert bench._relerr(np.asarray(ao) / go, np.asarray(af) / gf) <= 2e-6, (which, f.mode, k)


def test_the_printed_line_fits_the_drivers_record():
    """The driver keeps a 2000-character tail of stdout: the printed line must fit with every other configuration in it
    (round 5's 14 KB line left the record with fragments of two of six).  Input: round 5's verbose object
    (profiles/r05_bench_default.json) with two more children, as the default command now has eight."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    oc = full["other_configs"]
    assert len(oc) == 6
    oc["ft8tri"] = dict(oc["rx6"])
    oc["test2rx"] = dict(oc["c2"])
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= 1900, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "traffic_source"}
    assert abs(line["roofline"]["traffic_ratio"] - full["roofline"]["traffic"] / full["roofline"]["algorithmic_bytes_per_launch"]) < 1e-3
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert set(line["other_configs"]) == set(bench.OTHER_CONFIGS) and all(len(v) == 6 for v in line["other_configs"].values())
    assert line["full"] == "gpurun_out/bench_full.json" and abs(line["value"] - full["value"]) <= 1e-6 * full["value"]


def test_live_traffic_falls_back_quietly_without_the_profiler(monkeypatch, tmp_path):
    """The default command measures its own HBM traffic in two rocprofv3 --pmc child passes; where the profiler is missing or a
    pass fails the line keeps the stored, hash-guarded figure: live_traffic() answers None, it never raises."""
    import shutil
    import subprocess
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, "exists", lambda p: False if str(p).endswith("rocprofv3") else real_exists(p))
    assert bench.live_traffic(None) is None
    # a profiler that is there but fails
    monkeypatch.setattr(shutil, "which", lambda name: "/bin/false")
    monkeypatch.setattr(os.path, "exists", real_exists)
    assert bench.live_traffic(None) is None
    # ... and one whose passes succeed: 2 * FETCH_SIZE + WRITE_SIZE in KiB, full-batch launches only
    def fake_run(cmd, **kw):
        ctr = cmd[cmd.index("--pmc") + 1]
        d = os.path.join(cmd[cmd.index("-d") + 1], "host", "1")
        os.makedirs(d)
        with open(os.path.join(d, "1_counter_collection.csv"), "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for v in (100.0, 100.0, 3.0):                       # the 3.0: a small launch of the same kernel (host-fed slots)
                f.write(f'"void pysdr::(anonymous namespace)::mixdec_kernel<4, 6, 1024, 0, 0>(pysdr::MixDecArgs)",{ctr},{v if ctr == "FETCH_SIZE" else v / 10}\n')
            f.write(f'"__amd_rocclr_fillBufferAligned",{ctr},5\n')
        return subprocess.CompletedProcess(cmd, 0, b"", b"")
    monkeypatch.setattr(shutil, "which", lambda name: "/bin/true")
    monkeypatch.setattr(subprocess, "run", fake_run)
    got = bench.live_traffic(None)
    assert got == {"mixdec_kernel": (2 * 100.0 + 10.0) * 1024.0}
