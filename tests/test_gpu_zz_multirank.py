"""The N > 1 path of bench.py on ONE device: two rank processes (gloo control plane, both on device 0,
stream-sharded = BASELINE config #5's layout with two of its eight streams), each checking the last
timed step's device buffers against the float32 oracle on its OWN stream seed (`--verify`, default
on when --gpus > 1).  The fan-out this stands for: receiver.py:726-739, am.py:85-114.  RCCL refuses
two ranks on one GPU, so the split-RX broadcast is covered with a world of one in
test_gpu_zz_rccl.py; here the line a future SCALE run prints is shown to certify itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--full-line"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_on_one_device_verify_themselves_against_the_oracle():
    p, out = _bench("--gpus", "2", "--chunks", "64", "--steps", "2", "--verify", "--no-cpu-baseline", "--no-host-fed")
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["rccl_ranks"] == 0
    assert out["verified_ranks"] == 2 and out["verify_worst_rel"] <= 1e-5
    ranks = out["verify"]["ranks"]
    assert sorted(r["rank"] for r in ranks) == [0, 1]
    for r in ranks:
        rx = [c for c in r["checks"] if "rx" in c]
        assert [c["mode"] for c in rx] == ["USB", "CW", "NFM", "AM"] and all(c["counts_ok"] for c in rx)
        assert all(c["primed_chunks"] == 192 for c in rx)          # (5 + 2 - 1) steps x 64 chunks lie in front of the last step
        assert any("psd_frame" in c for c in r["checks"])
    # two streams (seeds 10 and 11): the two ranks did different work and were timed separately
    assert len(out["per_rank_ms"]) == 2 and all(t > 0 for t in out["per_rank_ms"])
    assert sorted(r["stream_seed"] for r in ranks) == [10, 11]


def test_a_world_of_eight_on_one_device():
    """BASELINE config #5 IS a world of eight (8 streams x 4 RX, seeds 10-17, one stream per rank: SURVEY 8(d) C5,
    am.py:85-114); no 8-GPU node has been available to any round, so the one thing that can be proven is that the
    line such a run prints holds together: eight rank processes (all on device 0: device = local_rank % ndev), each
    with its own stream, each verified against the oracle on its own seed, a clock per rank, the exit code gating."""
    p, out = _bench("--gpus", "8", "--chunks", "16", "--steps", "2", "--verify", "--no-cpu-baseline", "--no-host-fed",
                    "--no-other-configs", timeout=900)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["rccl_ranks"] == 0
    assert out["verified_ranks"] == 8 and out["verify_worst_rel"] <= 1e-5
    ranks = out["verify"]["ranks"]
    assert sorted(r["rank"] for r in ranks) == list(range(8))
    assert sorted(r["stream_seed"] for r in ranks) == list(range(10, 18))
    assert len(out["per_rank_ms"]) == 8 and all(t > 0 for t in out["per_rank_ms"])
    # whole-job rate = 8 streams' samples over the slowest rank's clock
    assert out["config"]["parallelism"].startswith("stream-sharded x8")
    assert abs(out["value"] - 8 * out["config"]["samples_per_step"] / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-6 * out["value"]


def test_a_wrong_answer_fails_the_line():
    """--verify is a gate, not a decoration: with the checker's tolerance forced to zero the same run exits non-zero."""
    code = ("import sys; sys.argv = ['bench.py', '--chunks', '16', '--steps', '1', '--warmup', '1', '--verify', "
            "'--no-cpu-baseline', '--no-host-fed', '--full-line']; import bench; bench.VERIFY_TOL = 0.0; sys.exit(bench.main())")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 3 and "--verify FAILED" in p.stderr, (p.returncode, p.stderr[-2000:])
    out = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["verified_ranks"] == 0
