"""Generate the committed golden vectors under tests/golden/.

    python tests/golden/make_golden.py

The reference's own arithmetic (module sig_proc of aa2il/libs) is not in /root/reference and
cannot be imported, so NO captured reference I/O exists.  These vectors are outputs of this
repository's float32 CPU oracle (oracle/sdr_oracle.py) on seeded inputs; they freeze the
oracle (a change of the DSP spec shows up as a diff here) and give the GPU tests a fixed
target that does not depend on recomputing the oracle.

small_*.npz   reduced-rate cases with the INPUT stored (256 kS/s -> 48 kHz, 3/16)
c{1,2,3}.npz  the SURVEY 8(d) configurations: input regenerated from the seed (a checksum
              of it is stored), expected .am / .iq of every sub-receiver for the first chunks
psd_*.npz     spectrum.periodogram lines
c4.npz        config C4 (10 MS/s, WFM2 stereo): .am (L + jR) and .iq (S + jD) of chunks 1..2
rtty.npz      RTTY filterbank lines (rtty.py:822-846) of a stored 48 kHz FSK signal
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sdr_oracle as so  # noqa: E402
from oracle import rtty_oracle as ro  # noqa: E402
from oracle import wfm_oracle as wo  # noqa: E402

SMALL = dict(fs=256e3, fs_out=48e3, ntaps_dec=255, noise=2e-3,
             carriers=[dict(f=20e3, kind='am', amp=0.25, tone=800.0, depth=0.5),
                       dict(f=-30e3, kind='fm', amp=0.25, tone=1000.0, dev=3000.0),
                       dict(f=60e3, kind='usb', amp=0.2, tone=900.0),
                       dict(f=-80e3, kind='cw', amp=0.2)],
             rx=[dict(frq=20e3, mode='AM', video_bw=10e3, af_bw=5e3),
                 dict(frq=-30e3, mode='NFM', video_bw=20e3, af_bw=4e3),
                 dict(frq=60e3, mode='USB', video_bw=10e3, af_bw=3e3),
                 dict(frq=-80e3, mode='CW', video_bw=10e3, af_bw=500.0, bfo=700.0)])


def run(cfg, x, L, nchunks):
    rxs = so.make_receivers(cfg, np.float32)
    out = {}
    for i, rx in enumerate(rxs):
        am, iq = [], []
        for k in range(nchunks):
            am.append(rx.demod_data(x[k * L:(k + 1) * L]))
            iq.append(rx.iq)
        out[f'am{i}'] = np.concatenate(am)
        out[f'iq{i}'] = np.concatenate(iq)
        out[f'n{i}'] = np.array([len(a) for a in am], np.int32)
        out[f'gain{i}'] = np.float32(rx.agc.gain)
    return out


def main():
    L = so.chunk_sizes(SMALL['fs'], SMALL['fs_out'])[3]
    x = so.synth_iq(SMALL, 6 * L, 101)
    np.savez_compressed(os.path.join(HERE, 'small_4rx.npz'), x=x, L=L, nchunks=6, **run(SMALL, x, L, 6))
    for name, seed, nchunks in (('C1', 1, 3), ('C2', 2, 2), ('C3', 3, 2)):
        cfg = so.CONFIGS[name]
        L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
        x = so.synth_iq(cfg, nchunks * L, seed)
        chk = np.array([np.sum(x.real.astype(np.float64)), np.sum(x.imag.astype(np.float64)),
                        float(x[12345].real), float(x[-1].imag)])
        np.savez_compressed(os.path.join(HERE, name.lower() + '.npz'), seed=seed, L=L, nchunks=nchunks,
                            input_checksum=chk, **run(cfg, x, L, nchunks))
    cfg = so.CONFIGS['C3']
    x = so.synth_iq(cfg, 32768, 33)
    sp = so.Spectrum(8000.0, 32768, 65536, 0.0, np.float32)
    np.savez_compressed(os.path.join(HERE, 'psd_rf64k.npz'), seed=33, psd=sp.periodogram(x, True))
    x = so.synth_iq(SMALL, 3 * 2048, 34)
    sp = so.Spectrum(48.0, 4096, 8192, 0.5, np.float32)
    lines = np.stack([sp.periodogram(x[i:i + 2048], True) for i in range(0, 3 * 2048, 2048)])
    np.savez_compressed(os.path.join(HERE, 'psd_af8k.npz'), x=x, psd=lines)
    # C4: 3 chunks of 213333 samples at 10 MS/s, stereo; chunk 0 (start-up on an empty FIR) is not stored
    fs, L4 = 10e6, 213333
    x = wo.synth_wfm(fs, 3 * L4, 4)
    rx = wo.WfmReceiver(fs, 48e3, 300e3, stereo=True, ntaps_dec=255, dtype=np.float32)
    am, iq = [], []
    for k in range(3):
        am.append(rx.demod_data(x[k * L4:(k + 1) * L4]))
        iq.append(rx.iq)
    chk = np.array([np.sum(x.real.astype(np.float64)), np.sum(x.imag.astype(np.float64)),
                    float(x[12345].real), float(x[-1].imag)])
    np.savez_compressed(os.path.join(HERE, 'c4.npz'), seed=4, L=L4, nchunks=3, input_checksum=chk,
                        am=np.concatenate(am[1:]), iq=np.concatenate(iq[1:]),
                        n=np.array([len(a) for a in am], np.int32))
    # RTTY filterbank: 12 symbols of FSK at 48 kHz, input stored, lines as float32
    xr, bits = ro.synth_rtty(48000, 12, 1500.0, seed=6, noise=2e-3)
    lines = ro.RttyFilterbank(48000).push(xr).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'rtty.npz'), x=xr, bits=bits, lines=lines)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
