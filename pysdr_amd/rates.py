"""Rate / size arithmetic of the reference's run-time parameter object
(``params.py:399-472``, ``utils.py:277-289``, ``receiver.py:200,818-820``)."""
from math import gcd

OUT_CHUNK_SIZE = 1024            # params.py:440 "Pulse audio wants chunks of 1024"
MAX_RX = 6                       # params.py:33


def up_dn(fs_in, fs_out):
    """Interpolation/decimation factors: fs_out/fs_in in lowest terms
    (``params.py:405``; known answers ``srates.py:35-74``)."""
    a, b = int(round(fs_out)), int(round(fs_in))
    g = gcd(a, b)
    return a // g, b // g


def derive(srate, fs_out_req):
    """-> dict(UP, DOWN, FS_OUT, IN_CHUNK_SIZE) exactly as ``params.py:405-406,444``."""
    up, down = up_dn(srate, fs_out_req)
    fs_out = int(srate * up / down)
    in_chunk = int(OUT_CHUNK_SIZE * down / float(up) + 0 * 0.5)
    return dict(UP=up, DOWN=down, FS_OUT=fs_out, IN_CHUNK_SIZE=in_chunk)


def ring_buffer_size(num_rx, sdr_type, fs_out):
    """``params.py:456-468``."""
    n = 32 * OUT_CHUNK_SIZE
    if num_rx > 2:
        n *= 4
    if sdr_type == 'rtlsdr':
        n *= 2
    if fs_out > 100e3:
        n *= 4
    elif fs_out > 50e3:
        n *= 2
    return n


def adjust_foffset(foffset, srate, rb_size):
    """Snap the tuning offset to a multiple of SRATE/RB_SIZE (``utils.py:277-289``)."""
    m = round(rb_size * foffset / srate)
    return m * srate / rb_size


def af_gain(slider):
    """AF slider -> linear gain (``receiver.py:171,200``)."""
    return pow(10., slider) - 1


def out_range(s0, s1, up, down):
    """Output indices produced by input samples [s0, s1) of the rational resampler."""
    return -(-s0 * up // down), -(-s1 * up // down)
