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
