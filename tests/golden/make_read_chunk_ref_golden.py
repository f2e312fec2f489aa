"""Generate tests/golden/read_chunk_ref.npz by EXECUTING the reference's own SDR_EXECUTIVE.read_chunk
(/root/reference/receiver.py:538-631, extracted with `ast` and compiled as it stands; build
container only: /root/reference does not travel).

`self` and `P` are attribute bags initialised as receiver.py:436-445 does (xold = [], xx / x =
complex64[IN_CHUNK_SIZE]); the module-level names the method reads are given their reference values
(SAMPLE_FORMAT = SOAPY_SDR_CF32, receiver.py:34) or a stand-in that only reports (`error_trap`).
`P.sdr` is a scripted SoapySDR-shaped device: readStream(stream, [buf], n) copies the next
`schedule[k]` samples of a recorded stream into buf (capped at n and at the end of the stream)
and returns an object whose `.ret` is that count (soapy.py:33-48).  Two legs:
  live    short reads, zero-length reads, reads that overshoot the chunk -> the `xold` carry (:579-622)
  replay  REPLAY_MODE slicing of a recording with the strict `<` of :543 (the tail that does not
          fill a chunk is dropped, RX_DONE is set) and no tuning offset (P.lo.fo == 0)
The fixture holds the stream, the schedule and the chunks the reference assembled: data only.

    python tests/golden/make_read_chunk_ref_golden.py
"""
import ast
import os
import types

import numpy as np

REF = "/root/reference/receiver.py"
HERE = os.path.dirname(os.path.abspath(__file__))


class Result:
    def __init__(self, ret):
        self.ret = ret


class ScriptedSDR:
    def __init__(self, samples, schedule):
        self.samples, self.schedule, self.pos, self.k = samples, list(schedule), 0, 0

    def readStream(self, stream, buffs, n, timeoutUs=100000):
        want = self.schedule[self.k % len(self.schedule)]
        self.k += 1
        k = min(want, n, len(self.samples) - self.pos)
        if k <= 0:
            return Result(0)
        buffs[0][:k] = self.samples[self.pos:self.pos + k]
        self.pos += k
        return Result(k)


def main():
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SDR_EXECUTIVE"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "read_chunk"][0]
    assert (fn.lineno, fn.end_lineno) == (538, 631), (fn.lineno, fn.end_lineno)
    ns = dict(np=np, SOAPY_SDR_CF32="CF32", SAMPLE_FORMAT="CF32",
              error_trap=lambda *a, **k: (_ for _ in ()).throw(RuntimeError(a)))
    exec(compile(ast.Module([fn], []), "receiver.py:538-631", "exec"), ns)
    read_chunk = ns["read_chunk"]

    L = 1365
    rng = np.random.default_rng(8)
    stream = (rng.standard_normal(11 * L + 77) + 1j * rng.standard_normal(11 * L + 77)).astype(np.complex64)
    schedule = [400, 0, L, 37, 5000, 1, 0, 0, 964, L - 1, 2, 700, 700, 3 * L]
    P = types.SimpleNamespace(IN_CHUNK_SIZE=L, REPLAY_MODE=False, Stopper=None, USE_FAKE_RTL=False, rxStream=1,
                              sdr=ScriptedSDR(stream, schedule), RX_DONE=False)
    self = types.SimpleNamespace(P=P, DEBUG=False, xold=[])                    # receiver.py:436
    self.xx = np.array([0] * L, np.complex64)                                  # :438
    self.x = np.array([0] * L, np.complex64)                                   # :445
    live, carry = [], []
    for _ in range(10):
        read_chunk(self)
        live.append(self.x.copy())
        carry.append(len(self.xold))
    # replay leg
    P2 = types.SimpleNamespace(IN_CHUNK_SIZE=L, REPLAY_MODE=True, RX_DONE=False, lo=types.SimpleNamespace(fo=0),
                               players=[types.SimpleNamespace(active=False)])
    s2 = types.SimpleNamespace(P=P2, DEBUG=False, raw=stream[:4 * L], praw=0, x=np.array([0] * L, np.complex64))
    replay, done = [], []
    for _ in range(5):
        read_chunk(s2)
        replay.append(np.array(s2.x).copy())
        done.append(bool(P2.RX_DONE))
    np.savez_compressed(os.path.join(HERE, "read_chunk_ref.npz"), stream=stream, L=L, schedule=np.array(schedule),
                        live=np.stack(live), carry=np.array(carry), replay=np.stack(replay), replay_done=np.array(done))
    print("live", np.stack(live).shape, "carry", carry, "replay done", done)


if __name__ == "__main__":
    main()
