"""The N>1 layout on CPU: world_size-2 gloo processes (127.0.0.1).  Shard-by-stream (C5) and
the split-RX variant with a per-chunk broadcast both reproduce the single-process result;
the benchmark's max-over-ranks clock returns the slowest rank's time on every rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, os.environ["PYSDR_ROOT"])
import numpy as np
import torch.distributed as dist
from oracle import sdr_oracle as so
from pysdr_amd import multi
from tests.test_golden import small_cfg

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
cfg = small_cfg()
L = so.chunk_sizes(cfg["fs"], cfg["fs_out"])[3]
nchunks, nstreams = 3, 4
streams = [so.synth_iq(cfg, nchunks * L, 200 + s) for s in range(nstreams)]

def make_rx(si, idx):
    rxs = so.make_receivers(cfg, np.float32)
    return rxs if idx is None else [rxs[i] for i in idx]

by_stream = multi.run_sharded(streams, make_rx, L, nchunks, dist, mode="stream")
by_rx = multi.run_sharded(streams, make_rx, L, nchunks, dist, mode="rx", nrx=len(cfg["rx"]))
slow = multi.max_over_ranks(1.0 + rank, dist)
dist.barrier()
if rank == 0:
    with open(os.environ["PYSDR_OUT"], "wb") as f:
        pickle.dump(dict(by_stream=by_stream, by_rx=by_rx, slow=slow), f)
else:
    assert by_stream is None and by_rx is None and slow == float(world)
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partitions():
    from pysdr_amd import multi
    assert multi.partition_streams(8, 8) == [[i] for i in range(8)]
    assert multi.partition_streams(8, 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert multi.partition_rx(4, 2) == [[0, 2], [1, 3]]
    assert multi.partition_rx(4, 8)[5] == []
    assert multi.partition_streams(4, 8) == [[0], [1], [2], [3], [], [], [], []]
    assert multi.max_over_ranks(2.5) == 2.5
    assert multi.gather_audio({(0, 0): np.ones(3)})[(0, 0)].sum() == 3


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_world_matches_single_process(tmp_path, world):
    """world = 2: two ranks, two streams each.  world = 8 (BASELINE config #5's world, am.py:85-114): 4 streams over 8
    ranks -- ranks 4-7 hold NO stream (partition_streams' empty shares) and still take part in every collective; the
    split-RX variant deals 4 sub-receivers over 8 ranks the same way."""
    import pickle
    from oracle import sdr_oracle as so
    from tests.test_golden import small_cfg
    port = free_port()
    out = str(tmp_path / "res.pkl")
    wfile = tmp_path / "worker.py"
    wfile.write_text(WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYSDR_ROOT=ROOT, PYSDR_OUT=out, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(wfile)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = pickle.load(open(out, "rb"))
    assert res["slow"] == float(world)
    cfg = small_cfg()
    L = so.chunk_sizes(cfg["fs"], cfg["fs_out"])[3]
    for s in range(4):
        x = so.synth_iq(cfg, 3 * L, 200 + s)
        for i, rx in enumerate(so.make_receivers(cfg, np.float32)):
            want = np.concatenate([rx.demod_data(x[k * L:(k + 1) * L]) for k in range(3)])
            assert np.array_equal(res["by_stream"][(s, i)], want)
            if s == 0:
                assert np.array_equal(res["by_rx"][(0, i)], want)
    assert sorted(res["by_stream"]) == [(s, i) for s in range(4) for i in range(4)]
    assert sorted(res["by_rx"]) == [(0, i) for i in range(4)]


SPLIT_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, os.environ["PYSDR_ROOT"])
import numpy as np
import torch.distributed as dist
import bench
from pysdr_amd import multi

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
corrupt = int(os.environ.get("PYSDR_CORRUPT_RANK", "-1"))
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
nrx = 4
rx_idx = multi.partition_rx(nrx, world)[rank]              # ranks 4 .. 7 of a world of 8 own NO sub-receiver
n = 4096
x = (np.arange(n) * (1 + 2j)).astype(np.complex64) if rank == 0 else np.zeros(n, np.complex64)
x = np.array(multi.broadcast_chunk_host(x, dist, src=0))   # the host stand-in of ncclBroadcast (bench.py --split rx: pysdr_comm_bcast)
if rank == corrupt:
    x[17] += 1                                             # a batch that is not the root's
mine = dict(ok=True, worst_rel=1e-7 * (1 + len(rx_idx)), checks=[dict(rx=i) for i in rx_idx], rank=rank, stream_seed=10,
            bcast_checksum=bench.checksum_words(x.view(np.uint64)))
allv = [None] * world
dist.all_gather_object(allv, mine)
v = bench.aggregate_verify(allv, True)
dist.barrier()
if rank == 0:
    with open(os.environ["PYSDR_OUT"], "wb") as f:
        pickle.dump(dict(verified=v["verified_ranks"], ranks=[(r["rank"], r["ok"], r["bcast_equals_root"], len(r["checks"])) for r in v["ranks"]]), f)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("corrupt", [-1, 6])
def test_split_rx_world_of_eight_with_four_receivers_gates_on_the_broadcast(tmp_path, corrupt):
    """`bench.py --gpus 8 --split rx` cannot run on one GPU (RCCL refuses two ranks of a communicator on one device) and no
    8-GPU node has been available to any round, so its parts are proven separately: HERE the stamp logic of the line with
    8 ranks and 4 sub-receivers -- ranks 4-7 own no receiver and still take part in the broadcast (gloo standing in for
    ncclBroadcast), every rank's copy is checksummed against the root's, and a rank whose copy differs (rank 6, which has
    nothing else to show) takes the whole line's exit code down.  The RCCL call itself: tests/test_gpu_zz_rccl.py (a world of one)."""
    import pickle
    port = free_port()
    out = str(tmp_path / "res.pkl")
    wfile = tmp_path / "worker.py"
    wfile.write_text(SPLIT_WORKER)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYSDR_ROOT=ROOT, PYSDR_OUT=out, OMP_NUM_THREADS="1", PYSDR_CORRUPT_RANK=str(corrupt))
        procs.append(subprocess.Popen([sys.executable, str(wfile)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = pickle.load(open(out, "rb"))
    assert [r[0] for r in res["ranks"]] == list(range(8))
    assert [r[3] for r in res["ranks"]] == [1, 1, 1, 1, 0, 0, 0, 0]          # RX r on rank r, nothing on ranks 4-7
    if corrupt < 0:
        assert res["verified"] == 8 and all(r[1] and r[2] for r in res["ranks"])
    else:
        assert res["verified"] == 7 and [r[0] for r in res["ranks"] if not r[1]] == [corrupt]


def test_rank_r_binds_device_r_on_an_eight_gpu_node(monkeypatch):
    """The launch contract of an 8-GPU node, dry: `bench.py --gpus 8` starts ranks 0-7 with LOCAL_RANK = rank before any GPU
    call (spawn_ranks), a rank binds device LOCAL_RANK when the node reports 8 devices (and wraps round on the one-GPU test
    boxes), and --split rx is refused when there are fewer devices than ranks."""
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, cwd=None):
            started.append(env)

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--split", "rx"])
    assert bench.spawn_ranks(bench.parse(["--gpus", "8", "--split", "rx"])) == 0
    assert [(e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"]) for e in started] == [(str(r), str(r), "8") for r in range(8)]
    assert [bench.pick_device(int(e["LOCAL_RANK"]), 8) for e in started] == list(range(8))
    assert [bench.pick_device(r, 1) for r in range(8)] == [0] * 8
    assert bench.split_rx_refusal(True, 8, 8) is None and bench.split_rx_refusal(False, 8, 1) is None
    assert "one GPU per rank" in bench.split_rx_refusal(True, 8, 1)
