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
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sdr_oracle as so  # noqa: E402

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
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
