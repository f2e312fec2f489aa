"""Generate tests/golden/host_loop_ref.npz by EXECUTING the reference's own `demodulate_data`
(/root/reference/receiver.py:231-297) and `audio_out` (:153-225), extracted with `ast` and compiled as
they stand (build container only: /root/reference does not travel).

What they call is scripted: `rx.demod_data(x)` sets and returns a prepared array (the DSP itself lives
in the absent sig_proc; here only what the reference does AROUND it is pinned), `rx.auto_mute(x)` follows
a schedule, the ring buffers / files / GUI button record what they are handed.  Two scenarios:
  A  AUDIO_SCHEME 1, 2 RX, P.MODE = 'AM' (DC removal :250-252), AF PSD tap on RX 0, SAVE_DEMOD,
     auto-mute toggling (:238-245), RX 1 muted by hand, AF_GAIN 0.37
  B  AUDIO_SCHEME 2, 3 RX (two on the first stereo player, the odd one alone :158-189), P.MODE = 'CW',
     complex audio (IQ-like) so that `.real` matters, baseband PSD tap + SAVE_BASEBAND
The fixture holds the scripted arrays and everything the reference pushed / saved: data only.

    python tests/golden/make_host_loop_ref_golden.py
"""
import ast
import os
import types

import numpy as np

REF = "/root/reference/receiver.py"
HERE = os.path.dirname(os.path.abspath(__file__))


class Rec:
    def __init__(self):
        self.items = []

    def push(self, a):
        self.items.append(np.array(a))

    def put(self, a):
        self.items.append(np.array(a))

    def save_data(self, a, **kw):
        self.items.append(np.array(a))


class FakeRx:
    def __init__(self, ams, iqs, mutes):
        self.ams, self.iqs, self.mutes, self.k = ams, iqs, mutes, 0
        self.am = self.iq = None

    def demod_data(self, x):
        self.am, self.iq = self.ams[self.k], self.iqs[self.k]
        return self.am

    def auto_mute(self, x):
        return bool(self.mutes[self.k])


class Player:
    def __init__(self):
        self.rb, self.active, self.started = Rec(), False, []

    def start_playback(self, delay, flag):
        self.started.append(delay)
        self.active = True


def scenario(P, rxs, nchunks, demodulate_data, audio_out):
    colors = []
    P.gui = types.SimpleNamespace(btn9=types.SimpleNamespace(setColor=colors.append))
    muted_trace = []
    for k in range(nchunks):
        for rx in rxs:
            rx.k = k
        for irx in range(P.NUM_RX):
            demodulate_data(P, np.zeros(8, np.complex64), irx)
        muted_trace.append(bool(P.AUTO_MUTED))
        audio_out(P)
    return colors, muted_trace


def main():
    tree = ast.parse(open(REF).read())
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("demodulate_data", "audio_out")}
    assert (fns["audio_out"].lineno, fns["demodulate_data"].lineno) == (153, 231)
    ns = dict(np=np)
    exec(compile(ast.Module([fns["audio_out"], fns["demodulate_data"]], []), "receiver.py:153-297", "exec"), ns)
    demodulate_data, audio_out = ns["demodulate_data"], ns["audio_out"]
    rng = np.random.default_rng(12)
    out = {}

    # ---- A
    n, nch = 64, 4
    ams = [[(rng.standard_normal(n) + 0.3).astype(np.float32) for _ in range(nch)] for _ in range(2)]
    iqs = [[(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(nch)] for _ in range(2)]
    mutes = [[0, 1, 1, 0], [0, 0, 1, 0]]
    rxs = [FakeRx(ams[i], iqs[i], mutes[i]) for i in range(2)]
    P = types.SimpleNamespace(rx=rxs, NUM_RX=2, MODE='AM', ENABLE_AUTO_MUTE=True, AUTO_MUTED=False, SHOW_AF_PSD=True, PLOT_RX=0,
                              PANADAPTOR=False, MP_SCHEME=1, rb_af=Rec(), SHOW_BASEBAND_PSD=False, ENABLE_RTTY=False,
                              SAVE_BASEBAND=False, SAVE_DEMOD=True, demod_io=Rec(), AUDIO_SCHEME=1, players=[Player(), Player()],
                              MUTED=[False, True, False, False, False, False], AF_GAIN=0.37, audio_playback=True, LOOPBACK=False,
                              AUX_AUDIO=False, DELAY=1024)
    colors, trace = scenario(P, rxs, nch, demodulate_data, audio_out)
    out.update(A_am=np.array(ams), A_iq=np.array(iqs), A_mutes=np.array(mutes), A_af_gain=0.37,
               A_player0=np.stack(P.players[0].rb.items), A_player1=np.stack(P.players[1].rb.items),
               A_rb_af=np.stack(P.rb_af.items), A_saved=np.stack(P.demod_io.items), A_colors=np.array(colors),
               A_auto_muted=np.array(trace), A_started=np.array([len(p.started) for p in P.players]))

    # ---- B
    ams = [[(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(nch)] for _ in range(3)]
    iqs = [[(rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) for _ in range(nch)] for _ in range(3)]
    rxs = [FakeRx(ams[i], iqs[i], [0] * nch) for i in range(3)]
    P = types.SimpleNamespace(rx=rxs, NUM_RX=3, MODE='CW', ENABLE_AUTO_MUTE=False, AUTO_MUTED=False, SHOW_AF_PSD=False, PLOT_RX=0,
                              PANADAPTOR=False, MP_SCHEME=1, SHOW_BASEBAND_PSD=True, rb_baseband=Rec(), ENABLE_RTTY=False,
                              SAVE_BASEBAND=True, baseband_iq_io=Rec(), SAVE_DEMOD=False, AUDIO_SCHEME=2,
                              players=[Player(), Player()], MUTED=[False, False, True, False, False, False], AF_GAIN=1.0,
                              audio_playback=True, LOOPBACK=False, AUX_AUDIO=False, DELAY=2048)
    scenario(P, rxs, nch, demodulate_data, audio_out)
    out.update(B_am=np.array(ams), B_iq=np.array(iqs), B_af_gain=1.0,
               B_player0=np.stack(P.players[0].rb.items), B_player1=np.stack(P.players[1].rb.items),
               B_rb_baseband=np.stack(P.rb_baseband.items), B_saved=np.stack(P.baseband_iq_io.items))
    np.savez_compressed(os.path.join(HERE, "host_loop_ref.npz"), **out)
    print({k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
