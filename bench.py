#!/usr/bin/env python3
"""Throughput of the pySDR receiver hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c1|c1synch|c2|c3|rx6|c4|c4mono|ft8tri|test2rx] [--ntaps N] [--split stream|rx]

One "step" = one pass of the hot path over one device-resident batch of `--chunks` chunks of one
synthetic wideband stream.  Default workload = SURVEY.md 8(d) config C3: the fused mix+decimate
kernel for all 4 sub-receivers (USB/CW/NBFM/AM), the 48 kHz detector/AF/AGC kernels, and the RF
PSD (chunk 32768 -> 64k FFT, every sample PSD'd).  The other BASELINE configurations run through
the same tool: c1 (am.py path: 2.048 MS/s, 1 RX AM, 1001 taps), c1synch (the same with the AM-Synch carrier PLL,
Tables.py:34, receiver.py:649), c2 (8 MS/s, 1 RX NBFM), rx6 (C3's stream through MAX_RX = 6 sub-receivers, params.py:33,
no PSD), c4 / c4mono (10 MS/s broadcast FM, pilot-PLL stereo / mono), and the reference's own multi-receiver launch scripts with
its default 1001-tap prototype (params.py:134): ft8tri (FT8tri:47-74: 8 MS/s, 3 RX USB, -vid_bw 45 -af_bw 5) and test2rx
(TEST:13-32: 4 MS/s, 2 RX NFM).  --ntaps re-runs any workload with another prototype length (c2 / c3 at 1001).

N > 1: one process per GPU.  Started by `torch.distributed.run` (RANK/LOCAL_RANK/WORLD_SIZE in the
environment) or, when those are absent, by this script itself: the parent starts N children
BEFORE it makes any GPU call, waits for them and exits with their worst code; rank 0 prints the
ONE JSON line.
  --split stream (default, config C5): every rank runs its own independent stream, no data-path
      collective, weak scaling;
  --split rx: ONE stream, sub-receiver r on rank r mod N; rank 0's batch is broadcast to the other
      GPUs with RCCL (pysdr_comm_bcast = ncclBroadcast over xGMI) every step -- the analogue of
      MP_SCHEME 3's queue fan-out (receiver.py:728-739); strong scaling over the RX count.
torch.distributed (gloo) is control plane only: rendezvous, barrier, max-over-ranks clock.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PSD_CHUNK, PSD_NFFT = 32768, 65536
DEFAULT_CHUNKS = {"c1": 4096, "c1synch": 4096, "c2": 2048, "c3": 2048, "rx6": 2048, "c4": 2048, "c4mono": 2048,
                  "ft8tri": 2048, "test2rx": 4096}
TUNING_ENV = ("PYSDR_TUNING", "PYSDR_MIXDEC_WGS", "PYSDR_MIXDEC_YFLUSH", "PYSDR_DEBUG_FLAGS", "PYSDR_PSD_GROUP",
              "PYSDR_PSD_ROCFFT", "PYSDR_PSD_PATH", "PYSDR_PSD_STREAMS", "PYSDR_PSD_PACKED", "PYSDR_WFM_PLL", "PYSDR_MIXDEC_MFMA", "PYSDR_MIXDEC_GRID", "PYSDR_RESAMP_PLAIN",
              "PYSDR_AM_PLL", "PYSDR_OVERLAP", "PYSDR_USE_DIAG_LIB", "PYSDR_LIB_VARIANT", "PYSDR_MIXDEC_FLAGS", "PYSDR_MFMA_FLAGS")
# the single-GPU configurations the default line carries next to C3: BASELINE.json configs[0], [1], [3], and the three the
# reference also runs that the driver's record did not hold until round 5 (mono broadcast FM, MAX_RX = 6, AM-Synch)
OTHER_CONFIGS = ("c1", "c2", "c4", "c4mono", "rx6", "c1synch", "ft8tri", "test2rx")
# A timed bracket carries ~1 ms that no step owns (the first steps after the idle barrier run slower: 10 / 30 / 100 / 300 steps of
# C1 = 0.422 / 0.398 / 0.373 / 0.369 ms per step): the workloads whose step is a fraction of a millisecond time at least this
# many steps (~60 ms) -- as other_configs children whatever K the driver passed for the C3 loop, and by default on their own.
MIN_STEPS = {"c1": 150, "c1synch": 150, "c2": 120, "rx6": 80, "c4": 60, "c4mono": 80, "ft8tri": 80, "test2rx": 80}
# ... and warm up for ~20 ms (five C1 steps are 2 ms of GPU work: the clocks have not come up yet -- 30 timed steps after 5 / 50 / 300
# warm-up steps = 0.390 / 0.364 / 0.367 ms per step; C3's five steps are 15 ms, and 20 or 60 change nothing there)
MIN_WARMUP = {"c1": 60, "c1synch": 60, "c2": 40, "rx6": 25, "c4": 20, "c4mono": 25, "ft8tri": 25, "test2rx": 25}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 30 for c3; for the short-step workloads as many as make ~60 ms, see MIN_STEPS)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 5 for c3, MIN_WARMUP for the short-step workloads)")
    ap.add_argument("--workload", default="c3", choices=sorted(DEFAULT_CHUNKS))
    ap.add_argument("--split", default="stream", choices=["stream", "rx"])
    ap.add_argument("--chunks", type=int, default=0, help="chunks per step (batch resident in HBM); 0 = workload default")
    ap.add_argument("--nrx", type=int, default=0, help="c3 only: first NRX of a 6-RX list (reference MAX_RX), 0 = the 4 of C3")
    ap.add_argument("--ntaps", type=int, default=0,
                    help="length of the decimator's prototype (the reference's -nfilt, default 1001: params.py:134); 0 = the workload's own "
                         "(BASELINE's 255 for c2 / c3 / rx6 / c4, 1001 for c1 / ft8tri / test2rx)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the verbose object itself (what bench_full.json holds) instead of the compact line; the children of "
                         "other_configs and the profile collection use it")
    ap.add_argument("--no-psd", action="store_true")
    ap.add_argument("--no-cpu-mp", action="store_true", help="skip the one-process-per-RX CPU figure")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=10, help=argparse.SUPPRESS)
    ap.add_argument("--overlap-psd", action="store_true", help="PSD on its own stream, unordered w.r.t. the demod")
    ap.add_argument("--serial-psd", action="store_true", help="PSD strictly behind the whole demod (incl. stage 2)")
    ap.add_argument("--no-demod", action="store_true", help="diagnostic: PSD only")
    ap.add_argument("--no-overlap", action="store_true",
                    help="A/B: every call on ONE stream (pysdr_set_overlap(ctx, 0)); default 1: the audio-rate half of a call with a serial "
                         "loop in it (AM-Synch, WFM2) runs beside the front end of the next one")
    ap.add_argument("--overlap-all", action="store_true", help="A/B: pysdr_set_overlap(ctx, 2), every call in two overlapped halves")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: no HIP events inside the calls of the timed loop (what the live kernel timing costs)")
    ap.add_argument("--tile-bytes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive ingest-ring figure")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks timed on the CPU oracle (0 = about --cpu-seconds worth)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="size of the CPU-oracle sample when --cpu-chunks is 0")
    ap.add_argument("--cpu-only", action="store_true",
                    help="time the CPU oracle on this workload and print {'cpu_baseline': ...}; never touches the GPU "
                         "(other_configs runs one such child per configuration, side by side)")
    ap.add_argument("--psd-hz", type=float, default=20.0,
                    help="c3, 1 GPU: also time the job with ONE 64k PSD frame per SRATE/psd_hz samples -- the reference's real "
                         "duty (20 Hz timer, pySDR.py:252-256, gui.py:1264-1267) -- reported as `at_reference_psd_duty`, never as "
                         "`value`; 0 = skip")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                    help="after the timed loop every rank checks the LAST step's device buffers against the float32 "
                         "oracle (first 2 chunks per sub-receiver, PSD frame 0); default ON")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="default command only: do not measure the HBM traffic of this run in two rocprofv3 --pmc child passes "
                         "(the line then carries the stored figure of profiles/, guarded by source hashes)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default command (workload c3, 1 GPU) also runs c1, c2 and c4 at their default batch for --steps "
                         "steps each, in child processes AFTER its own timed loop, and reports them as `other_configs` "
                         "(never as `value`); this switch skips that")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = MIN_STEPS.get(args.workload, 30)
    if args.warmup is None:
        args.warmup = MIN_WARMUP.get(args.workload, 5)
    return args


# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes.  Nothing in this
    process has touched the GPU (no library load, no device count), and no process that has is
    ever exec'ed: the children are fresh interpreters."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT))
    rc = 0
    deadline = time.time() + 3600
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=max(1.0, deadline - time.time()))))
        except subprocess.TimeoutExpired:
            rc = max(rc, 124)
    for p in procs:
        if p.poll() is None:
            p.kill()
    return rc


# ---------------------------------------------------------------------------------------------
RX6 = [dict(frq=200e3, mode='USB', video_bw=10e3, af_bw=3e3),
       dict(frq=-310e3, mode='CW', video_bw=10e3, af_bw=500.0, bfo=700.0),
       dict(frq=455e3, mode='NFM', video_bw=20e3, af_bw=4e3),
       dict(frq=-1.2e6, mode='AM', video_bw=10e3, af_bw=5e3),
       # sub-receivers 5 and 6 listen to carriers the C3 stream HAS (the FM one as LSB, the keyed one as AM): tuned to
       # empty spectrum (as in rounds 2-3: +900 kHz, -2.1 MHz) their output is stop-band leakage 80 dB down, which the AGC
       # blows up to full scale -- the verification (on by default now) then measured rounding noise against itself
       # (2.8e-5); the work per sample is the same
       dict(frq=455e3, mode='LSB', video_bw=10e3, af_bw=3e3),
       dict(frq=-310e3, mode='AM', video_bw=10e3, af_bw=5e3)]


def workload_cfg(args):
    """cfg dict in the shape of pysdr_amd.synth.CONFIGS (+ 'wfm': mono/stereo for C4)."""
    from pysdr_amd.synth import CONFIGS
    cfg = _workload_cfg(args)
    if args.ntaps:
        cfg['ntaps_dec'] = args.ntaps
    return cfg


def _workload_cfg(args):
    from pysdr_amd.synth import CONFIGS
    w = args.workload
    if w in ("c1", "c2", "c3", "ft8tri", "test2rx"):
        # ft8tri / test2rx: the reference's own multi-receiver launch scripts (FT8tri:47-74, TEST:13-32) with its default
        # 1001-tap prototype -- what pySDR really runs, where c2 / c3 are BASELINE's 255-tap configurations
        cfg = dict(CONFIGS[w.upper()])
        if w == "c3" and args.nrx:
            cfg['rx'] = RX6[:args.nrx]
        return cfg
    if w == "rx6":                       # the reference's MAX_RX (params.py:33) on C3's stream, demod only
        return dict(CONFIGS['C3'], rx=RX6[:6])
    if w == "c1synch":
        # C1 with the carrier PLL in the chain (MODES, Tables.py:34).  The synthetic stream is a loop of 8 chunks: the
        # carrier sits on the nearest frequency that closes the loop without a phase jump (17066 cycles in 8 x 43690
        # samples: 99997.6 Hz, the receiver stays tuned to 100 kHz = 2.4 Hz off) -- no station jumps by 146 degrees every
        # 170 ms, and a loop that re-acquires forever would be timed on its pull-in, not on tracking
        cfg = dict(CONFIGS['C1'])
        nloop = 8 * int(1024 * 128 / 3)
        f = round(100e3 * nloop / cfg['fs']) * cfg['fs'] / nloop
        cfg['carriers'] = [dict(cfg['carriers'][0], f=f)]
        cfg['rx'] = [dict(cfg['rx'][0], mode='AM-Synch')]
        return cfg
    return dict(fs=10e6, fs_out=48e3, ntaps_dec=255, wfm=('WFM2' if w == "c4" else 'WFM'),
                rx=[dict(frq=300e3, mode=('WFM2' if w == "c4" else 'WFM'), video_bw=200e3)])


def synth_batch(cfg, n, seed):
    from pysdr_amd.synth import synth_iq, synth_wfm
    if 'wfm' in cfg:
        return synth_wfm(cfg['fs'], n, seed)
    return synth_iq(cfg, n, seed)


def build_receivers(cfg, device, max_chunks, rx_idx=None):
    """sig_proc.Receiver objects of the workload on one context; rx_idx = this rank's share."""
    from pysdr_amd import sig_proc
    from pysdr_amd.params import RunTimeParams
    rxl = cfg['rx'] if rx_idx is None else [cfg['rx'][i] for i in rx_idx]
    if not rxl:
        return None, []
    r0 = rxl[0]
    kw = dict(foffset=300e3, vid_bw=200e3) if 'wfm' in cfg else {}
    P = RunTimeParams(fs=cfg['fs'], fsout=cfg['fs_out'], fc=[14.2e6] * len(rxl),
                      mode=r0['mode'], nfilt=cfg['ntaps_dec'], device=device,
                      max_batch_chunks=max_chunks, **kw)
    rxs = []
    for i, r in enumerate(rxl):
        P.VIDEO_BW = r.get('video_bw', 10e3)
        rx = sig_proc.Receiver(P, r['frq'], i, str(i + 1))
        rx.mode, rx.af_bw, rx.bfo = r['mode'], r.get('af_bw', 0.0), r.get('bfo', 0.0)
        rxs.append(rx)
    P.rx = rxs
    for rx in rxs:
        rx._sync_controls()
    return P, rxs


def oracle_receivers(cfg):
    from oracle import sdr_oracle as so
    if 'wfm' in cfg:
        from oracle import wfm_oracle as wo
        return [wo.WfmReceiver(cfg['fs'], cfg['fs_out'], r['frq'], stereo=(cfg['wfm'] == 'WFM2'),
                               ntaps_dec=cfg['ntaps_dec'], video_bw=r['video_bw']) for r in cfg['rx']]
    return so.make_receivers(cfg, np.float32)


def cpu_baseline(cfg, nchunks, with_psd, seed, seconds=10.0):
    """The NumPy/SciPy oracle (kind "port": the reference's own sig_proc is absent) on a
    bounded sample of the same workload, one host core."""
    from oracle import sdr_oracle as so
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    uniq = 8                              # same 8-chunk synthetic loop the GPU batch is built from
    x = synth_batch(cfg, uniq * L, seed)
    rxs = oracle_receivers(cfg)
    sp = so.Spectrum(cfg['fs'] / 1e3, PSD_CHUNK, PSD_NFFT, 0.0, np.float32)
    for rx in rxs:                       # warm-up chunk (BLAS init, page faults)
        rx.demod_data(x[:L])
    if nchunks <= 0:                     # size the sample to about `seconds` from one timed chunk
        t0 = time.perf_counter()
        for rx in rxs:
            rx.demod_data(x[L:2 * L])
        if with_psd:
            for i in range(0, L - PSD_CHUNK + 1, PSD_CHUNK):
                sp.periodogram(x[i:i + PSD_CHUNK], True)
        nchunks = int(max(4, min(4096, seconds / max(time.perf_counter() - t0, 1e-4))))
    t0 = time.perf_counter()
    for k in range(nchunks):
        xc = x[(k % uniq) * L:(k % uniq + 1) * L]
        for rx in rxs:
            rx.demod_data(xc)
        if with_psd:                     # every sample PSD'd: L/32768 frames per chunk
            for i in range(0, L - PSD_CHUNK + 1, PSD_CHUNK):
                sp.periodogram(xc[i:i + PSD_CHUNK], True)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return dict(value=nchunks * L / dt / 1e6, unit="MS/s", cores=1, kind="port",
                sample=f"{nchunks} chunks x {L} samples ({nchunks * L / cfg['fs']:.2f} s of signal), "
                       f"{len(rxs)} RX serial{' + 64k PSD' if with_psd else ''}, float32 NumPy/SciPy oracle, "
                       f"{dt:.1f} s wall; host has {os.cpu_count()} cores"), nchunks


def _cpu_worker(job):
    """One process of the NUM_RX-cores baseline: sub-receiver `irx` (or the PSD when irx < 0)
    over the same chunks; returns its own wall time."""
    cfg, nchunks, seed, irx = job
    from oracle import sdr_oracle as so
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    uniq = 8
    x = synth_batch(cfg, uniq * L, seed)
    if irx >= 0:
        rx = oracle_receivers(cfg)[irx]
        rx.demod_data(x[:L])
        t0 = time.perf_counter()
        for k in range(nchunks):
            rx.demod_data(x[(k % uniq) * L:(k % uniq + 1) * L])
        return time.perf_counter() - t0
    sp = so.Spectrum(cfg['fs'] / 1e3, PSD_CHUNK, PSD_NFFT, 0.0, np.float32)
    t0 = time.perf_counter()
    for k in range(nchunks):
        xc = x[(k % uniq) * L:(k % uniq + 1) * L]
        for i in range(0, L - PSD_CHUNK + 1, PSD_CHUNK):
            sp.periodogram(xc[i:i + PSD_CHUNK], True)
    return time.perf_counter() - t0


def cpu_baseline_per_rx(args, cfg, nchunks, with_psd, seed):
    """The analogue of the reference's MP_SCHEME 3 (one process per sub-receiver,
    receiver.py:726-739): NUM_RX (+1 for the PSD) single-threaded child processes
    (`bench.py --cpu-worker`, plain subprocesses with a deadline: the parent holds a HIP context
    and must neither fork it nor ever wait forever) over the same sample; the job's rate is set
    by the slowest of them.  Returns None if a child fails."""
    from oracle import sdr_oracle as so
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    ids = list(range(len(cfg['rx']))) + ([-1] if with_psd else [])
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i),
                               "--workload", args.workload, "--nrx", str(args.nrx), "--ntaps", str(args.ntaps),
                               "--cpu-chunks", str(nchunks), "--cpu-seed", str(seed)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
             for i in ids]
    times = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            times.append(float(out.strip().splitlines()[-1]))
    except Exception:
        for p in procs:
            if p.poll() is None:
                p.kill()
        return None
    wall = time.perf_counter() - t0
    dt = max(times)
    return dict(value=nchunks * L / dt / 1e6, unit="MS/s", cores=len(ids), kind="port",
                sample=f"{nchunks} chunks x {L} samples, one process per sub-receiver"
                       f"{' + one for the 64k PSD' if with_psd else ''}, slowest {dt:.1f} s "
                       f"({', '.join('%.1f' % t for t in times)}), {wall:.1f} s wall incl. start-up")


def host_fed_rate(ctx, cfg, L, cps, nslots_run=24):
    """The PCIe-inclusive figure (row N4; receiver.py:596 `readStream(rxStream, [self.xx], n)`, soapy.py:33-48): host
    arrays in, audio + baseband IQ out, through the ingest ring.  Two legs and what bounds them:
      value            the host COPIES every chunk into the pinned slot first (numpy, one thread) -- the stand-in for a
                       driver whose readStream() fills a buffer of its own that the caller then copies;
      slots_prefilled  the source has written into the slot it was GIVEN (readStream's contract: it fills the caller's
                       buffer, and the ring hands out its pinned slots as that buffer): submit + collect only;
      memcpy_GBps / h2d_GBps   the two one-thread / one-DMA rates behind them, measured here on this box (the first moved
                       between driver boxes 4.0 -> 5.2 GS/s worth; the second is the link's ceiling for ANY ingest)."""
    from pysdr_amd import _lib
    from pysdr_amd.ingest import IngestRing
    lib = _lib.lib()
    x = synth_batch(cfg, 8 * L, 10)
    ring = IngestRing(ctx, 3, cps)
    nbytes = cps * L * 8

    # the stand-in for a driver that fills a buffer of its own which the caller then copies: FILL_THREADS host threads (NumPy's
    # slice copy releases the GIL) -- with ONE thread this memcpy, not the link, bounded the figure (round 5: 77 % of the link)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=FILL_THREADS)

    def fill(slot):
        buf = ring.buffer(slot)

        def part(t):
            for k in range(t, cps, FILL_THREADS):
                buf[k * L:(k + 1) * L] = x[(k % 8) * L:(k % 8 + 1) * L]
        list(pool.map(part, range(FILL_THREADS)))

    def run(copy_in):
        slot, pending = 0, None
        for w in range(3):
            fill(w)
            ring.submit(w, cps * L)
            ring.collect(w)
        t_fill = t_sub = t_col = 0.0
        t0 = time.perf_counter()
        for j in range(nslots_run):
            ta = time.perf_counter()
            if copy_in:
                fill(slot)
            tb = time.perf_counter()
            ring.submit(slot, cps * L)
            tc = time.perf_counter()
            if pending is not None:
                ring.collect(pending)
            td = time.perf_counter()
            t_fill += tb - ta; t_sub += tc - tb; t_col += td - tc
            pending, slot = slot, (slot + 1) % 3
        ring.collect(pending)
        dt = (time.perf_counter() - t0) / (nslots_run * cps)
        return dt, {"fill_ms": t_fill / nslots_run * 1e3, "submit_ms": t_sub / nslots_run * 1e3,
                    "collect_wait_and_copy_out_ms": t_col / nslots_run * 1e3}

    dt, per_slot = run(True)
    dt2, per_slot2 = run(False)
    # the two rates behind the legs: one thread copying into a pinned slot, one DMA of a pinned slot to the device
    t0 = time.perf_counter()
    for _ in range(4):
        fill(0)
    memcpy_gbps = 4 * nbytes / (time.perf_counter() - t0) / 1e9
    d_tmp = C.c_void_p()
    h2d_gbps = None
    dev = int(ctx.cfg.device)
    if lib.pysdr_dev_alloc(dev, nbytes, C.byref(d_tmp)) == 0:
        src = C.c_void_p(ring.buffer(0).ctypes.data)
        lib.pysdr_dev_upload(dev, d_tmp, src, nbytes)
        t0 = time.perf_counter()
        for _ in range(8):
            lib.pysdr_dev_upload(dev, d_tmp, src, nbytes)
        h2d_gbps = 8 * nbytes / (time.perf_counter() - t0) / 1e9
        lib.pysdr_dev_free(dev, d_tmp)
    ring.close()
    pool.shutdown()
    return {"fill_threads": FILL_THREADS, "ms_per_chunk": dt * 1e3, "value": L / dt / 1e6, "unit": "MS/s", "chunks_per_slot": cps, "per_slot_ms": per_slot,
            "slots_prefilled": {"value": L / dt2 / 1e6, "unit": "MS/s", "ms_per_chunk": dt2 * 1e3, "per_slot_ms": per_slot2},
            "memcpy_GBps": memcpy_gbps, "h2d_GBps": h2d_gbps, "slot_MB": nbytes / 1e6,
            "pcie_ceiling_MSps": (h2d_gbps * 1e3 / 8.0) if h2d_gbps else None,
            "note": "host arrays in, audio + baseband out over PCIe: pinned ring slots, one H2D copy + one launch "
                    "sequence + 2*NUM_RX+1 D2H copies per slot, results collected one slot late; `value` includes the host "
                    "memcpy into the slot that stands for a readStream() with a buffer of its own, `slots_prefilled` is the "
                    "ring alone (the source wrote into the slot it was given); the memcpy runs on `fill_threads` host threads"}



# ---------------------------------------------------------------------------------------------
# --verify: the checker of the multi-rank line.  The oracle is used here as a CHECKER only, after
# the timed region; nothing it computes reaches `value`.
FILL_THREADS = 4             # host threads of the host-fed leg's stand-in memcpy
VERIFY_TOL = 1e-5            # north_star: 1e-5 relative float32 (same bar as tests/test_gpu_parity.py)
VERIFY_CHUNKS = 2            # chunks of the last step compared per sub-receiver
VERIFY_PRIME = {"nb": 192, "wfm": 16}   # chunks the oracle runs in front of them (AGC: 0.9^192 = 2e-9; pilot PLL: 17.5 tau = 6 chunks)


def step_offset(k, seam, nloop):
    """Offset into the synthetic loop at which step k reads (the stream is continuous across steps)."""
    return (k * seam) % nloop if seam else 0


def stream_slice(xu, nloop, seam, nsamp, s_abs, n):
    """n samples of this rank's stream from ABSOLUTE sample index s_abs: exactly what steps
    0, 1, ... handed to the context (step k = xu tiled, read from step_offset(k))."""
    out = np.empty(n, np.complex64)
    done = 0
    while done < n:
        k, i = divmod(s_abs + done, nsamp)
        j = (step_offset(k, seam, nloop) + i) % nloop
        m = min(n - done, nsamp - i, nloop - j)
        out[done:done + m] = xu[j:j + m]
        done += m
    return out


def primed_oracle(cfg, rx_idx, s_start):
    """Oracle receivers whose absolute counters (LO phase, resampler sample index, output index)
    say that `s_start` samples have gone by: the contexts on the GPU have processed every step
    since 0, the checker only the last VERIFY_PRIME chunks in front of what it compares."""
    rxs = oracle_receivers(cfg)
    if rx_idx is not None:
        rxs = [rxs[i] for i in rx_idx]
    for rx in rxs:
        rx.lo.phase = (rx.lo.fword * s_start) % (1 << 32)
        if 'wfm' in cfg:
            rx.front.n_abs = s_start
            n1 = -(-s_start // rx.d1)                       # IF samples produced before s_start
            rx.audio.n_abs = n1
            rx.demod.m_abs = -(-n1 * rx.up2 // rx.down2)
        else:
            rx.dec.n_abs = s_start
            rx.demod.m_abs = -(-s_start * rx.up // rx.down)
    return rxs


def _relerr(got, want):
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape:
        return float('inf')
    if want.size == 0:
        return 0.0
    return float(np.max(np.abs(got - want)) / max(float(np.max(np.abs(want))), 1e-30))


def verify_rank(cfg, ctx, rxs, rx_idx, xu, nloop, seam, nsamp, L, B, steps_done, psd=None):
    """Compare the device buffers the LAST step left behind with the oracle.  Returns
    dict(ok, worst_rel, checks=[...]).  psd = (lib, device, d_psd) on the rank that ran the PSD."""
    from oracle import sdr_oracle as so
    checks, worst = [], 0.0
    s0 = (steps_done - 1) * nsamp                           # absolute index of the last step's first sample
    nchk = min(VERIFY_CHUNKS, B)
    if rxs:
        prime = min(VERIFY_PRIME["wfm" if 'wfm' in cfg else "nb"], s0 // L)
        s_start = s0 - prime * L
        orx = primed_oracle(cfg, rx_idx, s_start)
        x = stream_slice(xu, nloop, seam, nsamp, s_start, (prime + nchk) * L)
        want_am = [[] for _ in orx]
        want_iq = [[] for _ in orx]

        def one_rx(i):                  # the sub-receivers are independent: one thread each (NumPy / SciPy release the GIL in the
            o = orx[i]                  # resampler and the FIRs, where the time goes); 6 RX x 194 chunks took 19 s in a row
            for k in range(prime + nchk):
                a = o.demod_data(x[k * L:(k + 1) * L])
                if k >= prime:
                    want_am[i].append(np.array(a))
                    want_iq[i].append(np.array(o.iq))

        if len(orx) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(orx)) as ex:
                list(ex.map(one_rx, range(len(orx))))
        else:
            one_rx(0)
        for i, o in enumerate(orx):
            am, iq, cn, _pk = ctx.fetch(i, B)
            n = int(cn[:nchk].sum())
            wa, wi = np.concatenate(want_am[i]), np.concatenate(want_iq[i])
            e_am = _relerr(am[:n], wa)
            e_iq = _relerr(iq[:n], wi)
            counts_ok = [int(v) for v in cn[:nchk]] == [len(a) for a in want_am[i]]
            worst = max(worst, e_am, e_iq)
            checks.append(dict(rx=(rx_idx[i] if rx_idx is not None else i), mode=o.mode, chunks=nchk, primed_chunks=prime,
                               am=e_am, iq=e_iq, counts_ok=counts_ok))
    if psd is not None:
        lib, device, d_psd = psd
        from pysdr_amd import _lib
        got = np.empty(PSD_NFFT, np.float32)
        _lib.check(lib.pysdr_dev_download(device, C.c_void_p(got.ctypes.data), d_psd, got.nbytes), "download psd")
        sp = so.Spectrum(cfg['fs'] / 1e3, PSD_CHUNK, PSD_NFFT, 0.0, np.float64)
        # pysdr_spectrum_batch reads the batch buffer from its first sample: frame 0 = the loop's first 32768
        ref = np.asarray(sp.periodogram(np.resize(xu, PSD_CHUNK).astype(np.complex128), True), np.float64)
        lin, rl = 10 ** (got.astype(np.float64) / 10.0), 10 ** (ref / 10.0)
        e = float(np.max(np.abs(lin - rl)) / rl.max())      # linear power, every bin, of the peak
        worst = max(worst, e)
        checks.append(dict(psd_frame=0, linear_power_err_of_peak=e))
    ok = all((c.get("am", 0) <= VERIFY_TOL and c.get("iq", 0) <= VERIFY_TOL and c.get("counts_ok", True)
              and c.get("linear_power_err_of_peak", 0) <= VERIFY_TOL) for c in checks)
    return dict(ok=bool(ok), worst_rel=worst, checks=checks)


def pick_device(local_rank, ndev):
    """Rank r of a node binds device r (one process per GPU); with fewer devices than ranks -- the single-GPU boxes of the
    tests, where every rank of a world of 8 shares device 0 -- the ranks wrap round."""
    return int(local_rank) % max(1, int(ndev))


def split_rx_refusal(split_rx, world, ndev):
    """--split rx broadcasts over RCCL, which refuses two ranks of one communicator on one device: the reason to refuse
    the run, or None."""
    if split_rx and world > ndev:
        return f"--split rx needs one GPU per rank (RCCL): {world} ranks, {ndev} device(s)"
    return None


def checksum_words(words, s=0, x=0):
    """(wrapping sum, xor) of uint64 words, continued from (s, x)."""
    with np.errstate(over='ignore'):
        s = np.uint64(s) + np.add.reduce(words, dtype=np.uint64)
        x = np.uint64(x) ^ (np.bitwise_xor.reduce(words) if len(words) else np.uint64(0))
    return int(s), int(x)


def aggregate_verify(allv, split_rx):
    """Every rank's verification record -> the line's stamp.  --split rx: a rank is only good if its copy of the broadcast
    batch has the root's checksum -- ranks WITHOUT a sub-receiver (4 RX over 8 GPUs: ranks 4-7) have nothing else to show,
    and a rank whose receivers pass on a batch that is not the root's would be passing on the wrong data."""
    if split_rx:
        for v in allv:
            v["bcast_equals_root"] = (v["bcast_checksum"] == allv[0]["bcast_checksum"])
            v["ok"] = bool(v["ok"] and v["bcast_equals_root"])
    return dict(verified_ranks=sum(1 for v in allv if v["ok"]),
                worst_rel=max(v["worst_rel"] for v in allv), tol=VERIFY_TOL, ranks=allv)


def device_checksum(lib, device, d_ptr, nbytes, piece=256 << 20):
    """64-bit checksum (wrapping sum and xor of the 8-byte words) of a DEVICE buffer, downloaded in
    pieces: what --split rx compares between the root's batch and every other rank's copy."""
    from pysdr_amd import _lib
    host = np.empty(min(piece, nbytes) // 8, np.uint64)
    s, x = 0, 0
    for off in range(0, nbytes, piece):
        n = min(piece, nbytes - off) // 8
        _lib.check(lib.pysdr_dev_download(device, C.c_void_p(host.ctypes.data), C.c_void_p(d_ptr + off), n * 8), "download")
        s, x = checksum_words(host[:n], s, x)
    return s, x


def source_sha(name):
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "pysdr_amd", "csrc", name), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(args, nrx, B, kernel_prefix, sources):
    """HBM bytes per launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
    correction of MI355X_MICROARCH.md), or None: valid only for the profiled configuration AND only
    while the kernel sources still hash to what was profiled (the profile file carries the hashes)."""
    if B != DEFAULT_CHUNKS[args.workload] or args.nrx or args.ntaps:
        return None, None
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        # C3 (the demod kernels are the same with and without the PSD) or the workload's own passes
        name = f"{tag}_pmc_traffic.json" if args.workload == "c3" else f"{tag}_{args.workload}_pmc_traffic.json"
        p = os.path.join(ROOT, "profiles", name)
        try:
            doc = json.load(open(p))
        except Exception:
            continue
        rows = doc["kernels"] if isinstance(doc, dict) else doc
        stamp = doc.get("source_sha256", {}) if isinstance(doc, dict) else {}
        if any(stamp.get(s) != source_sha(s) for s in sources):
            return None, f"profiles/{name} is stale: {', '.join(sources)} changed since it was collected"
        tot = sum(r["hbm_bytes_per_launch"] for r in rows if r["kernel"].startswith(kernel_prefix))
        if tot > 0:
            return tot, (f"profiles/{name} (git {doc.get('git_head', '?') if isinstance(doc, dict) else '?'}): "
                         "2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of this same command")
    return None, None


def live_traffic(args):
    """HBM traffic of THIS run's kernels, measured now: two child passes of this same command (3 steps each) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- counters in passes of their own, never beside a trace, as
    MI355X_MICROARCH.md prescribes -- -> {kernel: bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB}.  Launches of the
    full batch only (within a factor two of the largest of a kernel).  None where rocprofv3 is missing or a pass fails: the line
    then carries the stored, hash-guarded figure of profiles/ (measured_traffic) as before."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="pysdr_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__),
               "--no-cpu-baseline", "--no-host-fed", "--no-other-configs", "--no-verify", "--full-line", "--steps", "3", "--warmup", "1"]
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, TMPDIR="/tmp"))
            if p.returncode != 0:
                return None
            vals = {}
            for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == ctr and "pysdr" in r.get("Kernel_Name", ""):
                        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                        vals.setdefault(m.group(1) if m else r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not vals:
                return None
            for k, v in vals.items():
                v = [x for x in v if x >= 0.5 * max(v)]
                per.setdefault(k, {})[ctr] = sum(v) / len(v)
        except Exception:
            return None
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return {k: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for k, v in per.items()}


def other_configs(args):
    """The other single-GPU configurations (BASELINE.json configs[0], [1], [3]; mono FM, MAX_RX = 6, AM-Synch) through this
    same tool, one child process each, AFTER the C3 loop has been timed and its buffers freed: throughput, step time, the
    front-end kernel's and the job's roofline fraction and the verification stamp of each -- so that the driver's one
    record certifies every configuration, not only the headline one.  Their CPU baselines (the oracle on ~2 s of the
    same chunks, one core each: BASELINE.json's config #1 IS "am.py path ... CPU NumPy reference", am.py:54-75) run as
    CPU-only children side by side once the GPU children are done.  Never `value`."""
    res = {}
    for w in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", str(max(args.steps, MIN_STEPS.get(w, 0))), "--warmup",
               str(max(args.warmup, MIN_WARMUP.get(w, 0))), "--no-cpu-baseline", "--no-host-fed", "--no-other-configs", "--full-line"]
        if args.verify is not None:
            cmd.append("--verify" if args.verify else "--no-verify")
        t0 = time.time()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            res[w] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                      "warmup": d["warmup"], "workload": d["config"]["workload"],
                      "roofline": {"kernel": d["roofline_mixdec"]["kernel"], "frac": d["roofline_mixdec"]["frac"],
                                   "achieved": d["roofline_mixdec"]["achieved"], "unit": "GB/s",
                                   "avg_launch_ms": d["roofline_mixdec"]["avg_launch_ms"],
                                   "traffic": d["roofline_mixdec"]["traffic"]} if d.get("roofline_mixdec") else None,
                      "roofline_job": {"frac": d["roofline_job"]["frac"],
                                       "algorithmic_bytes_per_sample": d["roofline_job"]["algorithmic_bytes_per_sample"]}
                      if d.get("roofline_job") else None,
                      "kernel_ms": d.get("kernel_ms"), "step_ms_stats": d.get("step_ms_stats"), "pilot_pll": d.get("pilot_pll"),
                      "carrier_pll": d.get("carrier_pll"),
                      "verified_ranks": d.get("verified_ranks"), "verify_worst_rel": d.get("verify_worst_rel"),
                      "exit_code": p.returncode, "wall_s": round(time.time() - t0, 1)}
        except Exception as e:            # a failed child must not take the headline line with it
            res[w] = {"error": repr(e)[:300], "wall_s": round(time.time() - t0, 1)}
    if not args.no_cpu_baseline:
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = {w: subprocess.Popen([sys.executable, os.path.abspath(__file__), "--workload", w, "--cpu-only", "--cpu-seconds", "2"],
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
                 for w in OTHER_CONFIGS}
        for w, p in procs.items():
            try:
                out, _ = p.communicate(timeout=120)
                res[w]["cpu_baseline"] = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])["cpu_baseline"]
            except Exception as e:
                if p.poll() is None:
                    p.kill()
                res[w]["cpu_baseline"] = {"error": repr(e)[:200]}
    return res


def _sig(v, n=5):
    """Round to n significant digits (the printed line; bench_full.json keeps every digit)."""
    if v is None or isinstance(v, (bool, int, str)):
        return v
    try:
        return float(f"{float(v):.{n}g}")
    except (TypeError, ValueError):
        return v


def write_full(out, args):
    """The verbose object (every kernel time, verification detail, tuning echo, host-fed legs, other_configs in full) goes to a
    file beside the line: gpurun_out/ when that exists (it travels back from a GPU box), else the repo root."""
    name = "bench_full.json" if args.workload == "c3" and not args.ntaps else f"bench_full_{args.workload}{'_%d' % args.ntaps if args.ntaps else ''}.json"
    d = os.path.join(ROOT, "gpurun_out")
    path = os.path.join(d if os.path.isdir(d) else ROOT, name)
    try:
        with open(path, "w") as f:
            json.dump(out, f)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def compact_line(out, full_path):
    """The ONE printed JSON line, kept under ~1.9 KB: the driver's record holds a 2000-character tail of stdout, and a 14 KB line
    (round 5) left it with fragments of two of six other_configs.  Contract keys as they were; `roofline` adds the kernel's name,
    `traffic_ratio` (measured HBM bytes / algorithmic bytes: the waste, visible without arithmetic) and where the traffic figure
    came from; every other configuration is one row [GS/s, ms per step, kernel fraction of HBM, job fraction, worst verification
    error, CPU oracle MS/s on one core]; everything else is in `full`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: (_sig(out[k], 7) if k in ("value", "ms_per_step") else out[k]) for k in keep}
    cfgd = out["config"]
    line["config"] = {"workload": cfgd["workload"].split("; ")[0] + f"; {cfgd['chunks_per_step']} chunks x {cfgd['in_chunk']} in HBM",
                      "parallelism": cfgd["parallelism"].split(" (")[0]}
    r = out.get("roofline")
    if r:
        ratio = (r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else None
        line["roofline"] = {"bound": r["bound"], "achieved": _sig(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _sig(r["frac"], 4),
                            "traffic": _sig(r.get("traffic"), 5), "traffic_ratio": _sig(ratio, 4),
                            "traffic_source": ("none" if not r.get("traffic_source") else ("stale profile" if r.get("traffic") is None else
                                               ("live: 2 rocprofv3 --pmc child passes of this command" if r["traffic_source"].startswith("live") else
                                                r["traffic_source"].split(" (")[0] + " (stored PMC passes, source-hash guarded)"))),
                            "kernel": r["kernel"].split(" (")[0], "avg_launch_ms": _sig(r["avg_launch_ms"])}
    else:
        line["roofline"] = None
    rm = out.get("roofline_mixdec")
    if rm and (not r or rm["kernel"] != r["kernel"]):
        line["roofline_mixdec"] = {"kernel": rm["kernel"].split(" (")[0], "frac": _sig(rm["frac"], 4), "avg_launch_ms": _sig(rm["avg_launch_ms"])}
    if out.get("roofline_job"):
        line["roofline_job"] = {"frac": _sig(out["roofline_job"]["frac"], 4), "bytes_per_sample": _sig(out["roofline_job"]["algorithmic_bytes_per_sample"], 5)}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = ({"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample"].split(", float32")[0] + "; " + cb["sample"].rsplit("; ", 1)[-1]} if cb else None)
    mp = out.get("cpu_baseline_per_rx_process")
    if mp:
        line["cpu_baseline_per_rx_process"] = {"value": _sig(mp["value"]), "cores": mp["cores"]}
    for k in ("verified_ranks", "verify_worst_rel"):
        if k in out:
            line[k] = _sig(out[k], 3)
    if out.get("n_gpus", 1) > 1:
        line["per_rank_ms"] = [_sig(v, 4) for v in out["per_rank_ms"]]
        line["rccl_ranks"] = out.get("rccl_ranks")
    if out.get("host_fed"):
        line["host_fed_MSps"] = _sig(out["host_fed"]["value"], 4)
    if out.get("at_reference_psd_duty"):
        line["at_20Hz_psd_MSps"] = _sig(out["at_reference_psd_duty"]["value"], 5)
    oc = out.get("other_configs")
    if oc:
        rows = {}
        for w, d in oc.items():
            if "error" in d:
                rows[w] = "error"
                continue
            rows[w] = [_sig(d["value"] / 1e3, 4), _sig(d["ms_per_step"], 4), _sig((d.get("roofline") or {}).get("frac"), 3),
                       _sig((d.get("roofline_job") or {}).get("frac"), 3), _sig(d.get("verify_worst_rel"), 2) if d.get("verified_ranks") == 1 else "FAILED",
                       _sig((d.get("cpu_baseline") or {}).get("value"), 3)]
        line["other_configs"] = rows
        line["other_configs_cols"] = "GS/s,ms_per_step,kernel_frac,job_frac,verify_worst_rel,cpu_MSps_1core"
    if out.get("invalid"):
        line["invalid"] = out["invalid"]
    line["full"] = full_path
    return line


def main():
    args = parse()
    cfg = workload_cfg(args)
    if args.cpu_worker is not None:      # child of cpu_baseline_per_rx: CPU only, never touches the GPU
        print(_cpu_worker((cfg, args.cpu_chunks, args.cpu_seed, args.cpu_worker)))
        return 0
    if args.cpu_only:                    # child of other_configs: the oracle on a bounded sample, no GPU, no library load
        cb, _ = cpu_baseline(cfg, args.cpu_chunks, False, 10, args.cpu_seconds)
        print(json.dumps({"cpu_baseline": cb}), flush=True)
        return 0
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return spawn_ranks(args)     # before ANY GPU call in this process
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled number",
                  file=sys.stderr)
            return 2
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # Load the HIP library BEFORE torch so that both share one libamdhip64.
    from pysdr_amd import _lib
    lib = _lib.lib()
    _lib.require_gpu()
    ndev = _lib.device_count()
    device = pick_device(local_rank, ndev)

    dist = None
    if world > 1:
        import torch.distributed as dist      # gloo: barrier + MAX only, no GPU tensors
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    split_rx = args.split == "rx"
    do_verify = True if args.verify is None else bool(args.verify)
    refuse = split_rx_refusal(split_rx, world, ndev)
    if refuse:
        raise SystemExit(refuse)
    with_psd = (args.workload == "c3") and not args.no_psd and (not split_rx or rank == 0)
    B = args.chunks or DEFAULT_CHUNKS[args.workload]
    nrx_total = len(cfg['rx'])
    from pysdr_amd import multi
    rx_idx = multi.partition_rx(nrx_total, world)[rank] if split_rx else None
    P, rxs = build_receivers(cfg, device, B, rx_idx)
    from pysdr_amd import rates
    L = P.IN_CHUNK_SIZE if P is not None else rates.derive(cfg['fs'], cfg['fs_out'])['IN_CHUNK_SIZE']
    ctx = P._pysdr_stream if P is not None else None
    if ctx is None:                       # a rank without sub-receivers still takes part in the broadcast
        from pysdr_amd import sig_proc
        from pysdr_amd.params import RunTimeParams
        P = RunTimeParams(fs=cfg['fs'], fsout=cfg['fs_out'], fc=[14.2e6], mode='AM', nfilt=cfg['ntaps_dec'],
                          device=device, max_batch_chunks=B)
        ctx = sig_proc._context_for(P)
    nsamp = B * L
    if args.no_overlap or args.overlap_all:
        _lib.check(lib.pysdr_set_overlap(ctx.h, 2 if args.overlap_all else 0), "set_overlap")
    if args.tile_bytes or args.threads:
        _lib.check(lib.pysdr_set_tile(ctx.h, args.tile_bytes, args.threads or 1024), "set_tile")

    # synthetic stream: a loop of 8 unique chunks (seed per rank = its own stream) repeated to fill
    # the batch.  Broadcast FM: the loop is 1.7 M samples = a whole number of cycles of the carrier,
    # the pilot and both tones, so the repeated stream is seamless -- a pilot that jumps every
    # 8 chunks would keep the pilot PLL re-acquiring, which no real broadcast does.
    # A batch of B chunks is not a whole number of those 1.7 M samples, so step k reads the (one loop
    # longer) buffer from offset k * nsamp mod 1.7 M: what the receiver sees across steps is ONE
    # continuous stream, as from an antenna, not a station that jumps in phase every step.
    seed = 10 + (0 if split_rx else rank)
    nloop = 1700000 if 'wfm' in cfg else 8 * L
    seam = nsamp % nloop                                  # 0 for the narrow-band configurations
    nbuf = nsamp + (nloop if seam else 0)
    # The fast load path wants 16-byte aligned batches, i.e. an EVEN sample offset.  When the seam is
    # odd, odd steps read a second copy of the buffer that is shifted by one sample, from the even
    # offset below theirs: the stream stays exactly continuous (no sample is repeated or dropped).
    need_shifted = bool(seam & 1)
    d_x, d_x1 = C.c_void_p(), C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(device, nbuf * 8, C.byref(d_x)), "alloc x")
    if need_shifted:
        _lib.check(lib.pysdr_dev_alloc(device, nbuf * 8, C.byref(d_x1)), "alloc x (shifted copy)")
    xu = None
    if not split_rx or rank == 0 or do_verify:
        xu = synth_batch(cfg, nloop, seed)
    if not split_rx or rank == 0:
        for dst, src in ((d_x, xu), (d_x1, np.roll(xu, -1) if need_shifted else None)):
            if src is None:
                continue
            src = np.ascontiguousarray(src)
            for off in range(0, nbuf, nloop):
                n = min(nloop, nbuf - off)
                _lib.check(lib.pysdr_dev_upload(device, C.c_void_p(dst.value + off * 8),
                                                C.c_void_p(src.ctypes.data), n * 8), "upload")
    step_no = [0]
    bc = multi.RcclBroadcaster(ctx, dist) if split_rx else None

    sp = None
    nframes = nsamp // PSD_CHUNK
    d_psd = C.c_void_p()
    if with_psd:
        from pysdr_amd import design
        win = np.ascontiguousarray(design.psd_window(PSD_CHUNK), np.float32)
        sp = C.c_void_p()
        _lib.check(lib.pysdr_spectrum_create(device, PSD_CHUNK, PSD_NFFT, nframes, _lib.as_pf(win),
                                             C.byref(sp)), "spectrum_create")
        _lib.check(lib.pysdr_dev_alloc(device, nframes * PSD_NFFT * 4, C.byref(d_psd)), "alloc psd")

    def step():
        if bc is not None:
            # the wideband batch travels root -> every GPU on the context's stream, in front of the
            # kernels that read it (ncclBroadcast over xGMI)
            if sp is not None:
                _lib.check(lib.pysdr_spectrum_order(sp, ctx.h, 1), "spectrum_order")   # PSD of the last step has read d_x
            bc.bcast(d_x.value, nsamp * 8, 0)
        if not args.no_demod and rxs:
            off = step_offset(step_no[0], seam, nloop)
            base = d_x.value
            if off & 1:                                  # odd start: the copy shifted by one sample, one sample lower
                base, off = d_x1.value, off - 1
            step_no[0] += 1
            ctx.process_batch(base + off * 8, B, L, on_device=True)
        if sp is not None:
            # Same order as pySDR's RX thread (demod of the chunk, then its PSD): the two are
            # both HBM-bound, so overlapping them on two streams buys nothing and only smears
            # the per-kernel timings; --overlap-psd restores the two-stream form.
            if not args.overlap_psd and not args.no_demod:
                # behind the mix+decimate kernel only: the (VALU-bound) audio-rate stages run
                # beside the (memory-bound) PSD kernels
                _lib.check(lib.pysdr_spectrum_order(sp, ctx.h, 0 if args.serial_psd else 2), "spectrum_order")
            _lib.check(lib.pysdr_spectrum_batch(sp, d_x, nframes, PSD_CHUNK, d_psd), "spectrum_batch")
            if not args.overlap_psd and not args.no_demod:
                _lib.check(lib.pysdr_spectrum_order(sp, ctx.h, 1), "spectrum_order")

    def sync():
        _lib.check(lib.pysdr_sync(ctx.h), "sync")
        if sp is not None:
            _lib.check(lib.pysdr_spectrum_sync(sp), "spectrum_sync")

    for _ in range(args.warmup):
        step()
    sync()
    # Live kernel timing (HIP events inside the call) on the LAST quarter of the timed steps, at least 4: an event record
    # costs ~5.5 us of the stream's timeline on this runtime (three per call = 2-4 % of a C1 / C2 step,
    # scripts: bench.py --no-kernel-events), so the timing is sampled inside the timed region instead of riding on every step
    ev_steps = 0 if args.no_kernel_events else min(args.steps, 64, max(4, args.steps // 4))
    _lib.check(lib.pysdr_set_profile(ctx.h, 0), "profile")

    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev_steps and i == args.steps - ev_steps:
            _lib.check(lib.pysdr_set_profile(ctx.h, 1), "profile")
        step()
    sync()
    dt_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0

    # kernel timings: HIP events on the stream the kernels run on, averaged over the timed steps
    nev = 0 if (args.no_demod or not rxs) else ev_steps
    ms = C.c_float(0)
    k1, k2 = [], []
    for back in range(nev):
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, back, C.byref(ms)), "elapsed")
        k1.append(ms.value)
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 1, back, C.byref(ms)), "elapsed")
        k2.append(ms.value)
    # per-step spread of the timed loop (the step period on the stream: start of call k-1 -> start of call k,
    # which includes whatever the PSD ordered behind / in front of it took)
    periods = []
    for back in range(max(0, min(nev - 1, 62)) if nev else 0):
        if lib.pysdr_get_elapsed_ms(ctx.h, 3, back, C.byref(ms)) == 0:
            periods.append(ms.value)
    k1_ms = float(np.mean(k1)) if k1 else float('nan')
    k2_ms = float(np.mean(k2)) if k2 else None
    psd_ms = None
    if sp is not None:
        _lib.check(lib.pysdr_spectrum_elapsed_ms(sp, C.byref(ms)), "psd elapsed")
        psd_ms = ms.value

    per_rank_ms = [dt_local / args.steps * 1e3]
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt_local / args.steps * 1e3], dtype=torch.float64))
        per_rank_ms = [float(v.item()) for v in allt]

    # ---- --verify: the LAST timed step's device buffers against the oracle, on every rank
    verify = None
    if do_verify and not args.no_demod:
        mine = verify_rank(cfg, ctx, rxs, rx_idx, xu, nloop, seam, nsamp, L, B, step_no[0],
                           psd=(lib, device, d_psd) if sp is not None else None)
        mine["rank"] = rank
        mine["stream_seed"] = seed
        if split_rx:
            # the only proof that RCCL moved the bytes: every rank's copy of the batch == the root's
            mine["bcast_checksum"] = device_checksum(lib, device, d_x.value, nsamp * 8)
        allv = [mine]
        if dist is not None:
            allv = [None] * world
            dist.all_gather_object(allv, mine)
        verify = aggregate_verify(allv, split_rx)

    # ---- the reference's real PSD duty next to the headline (never `value`): the GUI's 20 Hz timer
    # takes one 32768-sample chunk per tick and flushes the backlog (pySDR.py:252-256, gui.py:1264-1267),
    # i.e. one 64k frame per SRATE/20 samples; everything else of the step is unchanged
    duty = None
    if sp is not None and world == 1 and args.psd_hz > 0 and not args.no_demod and not args.overlap_psd:
        hop20 = int(round(cfg['fs'] / args.psd_hz))
        nf20 = (nsamp - PSD_CHUNK) // hop20 + 1

        def step20():
            ctx.process_batch(d_x.value, B, L, on_device=True)
            _lib.check(lib.pysdr_spectrum_order(sp, ctx.h, 0 if args.serial_psd else 2), "spectrum_order")
            _lib.check(lib.pysdr_spectrum_batch(sp, d_x, nf20, hop20, d_psd), "spectrum_batch")
            _lib.check(lib.pysdr_spectrum_order(sp, ctx.h, 1), "spectrum_order")

        for _ in range(min(args.warmup, 3)):
            step20()
        sync()
        t20 = time.perf_counter()
        for _ in range(args.steps):
            step20()
        sync()
        dt20 = (time.perf_counter() - t20) / args.steps
        _lib.check(lib.pysdr_spectrum_elapsed_ms(sp, C.byref(ms)), "psd elapsed")
        bps20 = 8.0 + nrx_total * (P.UP / P.DOWN) * 12.0 + 4.0 * PSD_NFFT / hop20
        duty = {"value": nsamp / dt20 / 1e6, "unit": "MS/s", "ms_per_step": dt20 * 1e3, "psd_hz": args.psd_hz,
                "frames_per_step": int(nf20), "hop_samples": hop20, "psd_call_ms": ms.value,
                "roofline_job": {"bound": "hbm", "achieved": nsamp / dt20 * bps20 / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": nsamp / dt20 * bps20 / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_sample": bps20},
                "note": "same step as `value` but ONE 64k PSD frame per SRATE/psd_hz samples, the duty pySDR's 20 Hz GUI timer "
                        "really runs at (BASELINE.md: 8.8 B/sample); `value` PSDs every sample"}

    pll = cpll = None
    synch = bool(rxs) and cfg['rx'][0]['mode'] == 'AM-Synch'
    if ('wfm' in cfg or synch) and rxs:
        sg, pt = C.c_int(0), C.c_int(0)
        _lib.check(lib.pysdr_pll_stats(ctx.h, 0, C.byref(sg), C.byref(pt)), "pll_stats")
        jw, jd = C.c_int(0), C.c_float(0)
        _lib.check(lib.pysdr_pll_join_margin(ctx.h, 0, C.byref(jw), C.byref(jd)), "pll_join_margin")
        st = {"segments": sg.value, "patched_serially": pt.value,
              "widest_join": {"phase_words_of_2^32": jw.value, "tolerance": 1024 if synch else 512,
                              "integrator_rad_per_sample": jd.value, "tolerance_w": 2e-8 if synch else 1e-9}}
        if synch:
            nl = C.c_int(0)
            _lib.check(lib.pysdr_pll_linear_starts(ctx.h, 0, C.byref(nl)), "pll_linear_starts")
            st["linear_starts"] = nl.value        # segments whose warm-up was the linear solve (the rest walked it)
        pll, cpll = (None, st) if synch else (st, None)
    tune = (C.c_int32 * 8)()
    _lib.check(lib.pysdr_get_tuning(ctx.h, tune), "get_tuning")
    sp_tune = (C.c_int32 * 4)()
    if sp is not None:
        _lib.check(lib.pysdr_spectrum_get_tuning(sp, sp_tune), "spectrum_get_tuning")
    ablated = bool(tune[1])               # work-skipping switches active (diagnostic build only)

    nrx = len(rxs)
    is_wfm = 'wfm' in cfg
    n_out = nsamp * P.UP // P.DOWN
    # algorithmic bytes of ONE front-end launch: every input sample once (8 B, shared by all RX) +
    # the baseband / IF IQ it writes (8 B per RX per output)   [SURVEY.md 8(d), DESIGN.md 5]
    n_front_out = (nsamp // rxs[0].demod.wfm_d1) if (is_wfm and rxs) else n_out
    k1_bytes = nsamp * 8 + nrx * n_front_out * 8
    psd_bytes = nframes * (PSD_CHUNK * 8 + PSD_NFFT * 4)
    # whole job, SURVEY 8(d): input once + baseband IQ (8 B) and audio (4 B) per RX per output (+ PSD out)
    bytes_per_sample_job = 8.0 + nrx_total * (P.UP / P.DOWN) * 12.0 + (4.0 * PSD_NFFT / PSD_CHUNK if (args.workload == "c3" and not args.no_psd) else 0.0)
    streams = 1 if split_rx else world
    job_rate = streams * nsamp * args.steps / dt          # input samples per second, whole job

    def roof(kernel, nbytes, t_ms, traffic=(None, None), **extra):
        if t_ms is None or not (t_ms > 0) or ablated:
            return None
        a = nbytes / (t_ms * 1e-3) / 1e9
        d = dict(kernel=kernel, bound="hbm", achieved=a, peak=HBM_PEAK_GBPS, unit="GB/s", frac=a / HBM_PEAK_GBPS,
                 traffic=traffic[0], traffic_source=traffic[1], algorithmic_bytes_per_launch=nbytes, avg_launch_ms=t_ms)
        d.update(extra)
        return d

    mfma_on = bool(tune[7]) and nrx == 1 and ((not is_wfm and (P.UP, P.DOWN, cfg['ntaps_dec']) == (3, 128, 1001)) or
                                              (is_wfm and rxs[0].demod.wfm_d1 == 40 and cfg['ntaps_dec'] == 255))
    overlapped = bool(lib.pysdr_last_call_overlapped(ctx.h))
    mm4_on = (not is_wfm) and 2 <= nrx <= 6 and P.UP == 3 and -(-cfg['ntaps_dec'] // 3) in range(321, 337) and int(tune[5]) == 1024
    front_name = ((("mixdec_mfma_kernel (f32 MFMA, shifted-tap columns; " if mfma_on else
                    (f"mixdec_kernel<{nrx},21,MM> (f32 MFMA 4x4x1 blocks = tap residues, taps held in registers; " if mm4_on else f"mixdec_kernel<{nrx}> (")) +
                   "fused NCO mix + polyphase decimate, all RX)" + (" + am_phase_kernel" if synch else "")) if not is_wfm else
                  (("mixdec_mfma_kernel<1/40>" if mfma_on else "mixdec_kernel<1,16>") +
                   (" + wfm_disc_kernel (IF decimate, discriminator; the pilot loop runs on the second stream)" if overlapped else
                    " + wfm_disc / wfm_pll kernels (FM front end: IF decimate, discriminator, pilot PLL)")))
    r_front = roof(front_name, k1_bytes, k1_ms if k1 else None,
                   measured_traffic(args, nrx, B, "mixdec", ["mixdec_mfma.hip", "mixdec_mfma_geom.h"] if (mfma_on and not is_wfm)
                                    else (["mixdec_mfma.hip", "mixdec_mfma_geom.h", "resamp_small.hip"] if mfma_on else ["mixdec.hip"])))
    psd_tr = measured_traffic(args, nrx, B, "psd", ["psdfft.hip"])
    psd_pairs_per_call = None
    if psd_tr[0] is not None and sp is not None and sp_tune[0] > 0:
        # the profile's figure is per launch pair of one group of frames; one call = nframes / group of them
        per_launch = -(-int(sp_tune[0]) // max(1, int(sp_tune[2])))          # frames of one cols + rows launch pair
        psd_pairs_per_call = nframes / float(per_launch)
        psd_tr = (psd_tr[0] * psd_pairs_per_call, psd_tr[1] + f"; per launch pair of {per_launch} frames, scaled to the call")
    r_psd = roof("psd_cols_pk + psd_rows_pk (window, zero-pad, 64k four-step FFT with a 24-bit intermediate, |.|^2, dB, fftshift)"
                 if (sp is not None and sp_tune[3]) else "psd kernels (window, zero-pad, 64k FFT, |.|^2, dB, fftshift)",
                 psd_bytes, psd_ms, psd_tr,
                 note="one call = all frames of the batch; per launch figures are per call")
    # the kernel (group) that dominates the timed region carries the headline roofline object
    dominant = r_psd if (r_psd is not None and (not k1 or psd_ms >= k1_ms)) else r_front

    out = {
        # BASELINE.json's metric, named for what this line really ran (RX count and workload)
        "metric": f"complex IQ MS/s through {nrx_total}-RX demod chain" + ("" if args.workload == "c3" else f" ({args.workload})"),
        "value": job_rate / 1e6,
        "unit": "MS/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if split_rx else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.workload.upper()}: {cfg['fs'] / 1e6:g} MS/s synthetic IQ, {nrx_total} RX "
                         f"({'/'.join(r['mode'] for r in cfg['rx'])}), {cfg['ntaps_dec']}-tap prototype {P.UP}/{P.DOWN}"
                         f"{', + 64k-FFT RF PSD on every sample' if (args.workload == 'c3' and not args.no_psd) else ''}; "
                         f"{B} chunks x {L} samples per step resident in HBM; "
                         + ("ONE stream, sub-receivers split over the GPUs, batch broadcast by RCCL each step" if split_rx
                            else "one stream per GPU")),
            "chunks_per_step": B, "in_chunk": L, "samples_per_step": nsamp,
            "parallelism": (f"rx-split x{world} (ncclBroadcast of the batch per step)" if split_rx
                            else f"stream-sharded x{world} (no data-path collective)"),
        },
        "per_rank_ms": per_rank_ms,
        "rccl_ranks": world if split_rx else 0,
        "roofline": dominant,
        "roofline_mixdec": r_front,
        "roofline_psd": r_psd,
        "roofline_job": None if ablated else {
            "bound": "hbm", "achieved": job_rate / world * bytes_per_sample_job / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": job_rate / world * bytes_per_sample_job / 1e9 / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_sample": bytes_per_sample_job,
            "note": "per GPU: whole-step wall clock against SURVEY 8(d)'s compulsory bytes per input sample"},
        "kernel_ms": {"front": k1_ms if k1 else None, "stage2": k2_ms, "psd_call": psd_ms},
        "kernel_events": {"steps": nev, "of": args.steps,
                          "what": "the last steps of the timed loop carry HIP events inside the call (kernel_ms, roofline, step_ms_stats); "
                                  "an event record costs ~5.5 us of stream time, so the others run without"},
        "step_ms_stats": ({"min": float(np.min(periods)), "median": float(np.median(periods)), "max": float(np.max(periods)),
                           "n": len(periods), "front_min": float(np.min(k1)), "front_max": float(np.max(k1)),
                           "what": "HIP-event period between consecutive steps on the context's stream over the timed loop"}
                          if periods else None),
        "pilot_pll": pll,
        "carrier_pll": cpll,
        "tuning": {"diag_build": int(tune[0]), "build_flags_hash": int(lib.pysdr_build_flags_hash()), "debug_flags": int(tune[1]), "mixdec_wgs_per_cu": int(tune[2]),
                   "mixdec_yflush_cap": int(tune[3]), "tile_bytes": int(tune[4]), "threads": int(tune[5]),
                   "mixdec_mfma": int(tune[7]), "overlap_calls": int(lib.pysdr_get_overlap(ctx.h)), "last_call_overlapped": int(lib.pysdr_last_call_overlapped(ctx.h)),
                   "psd_group": int(sp_tune[0]) if sp is not None else None,
                   "psd_rocfft": int(sp_tune[1]) if sp is not None else None,
                   "psd_streams": int(sp_tune[2]) if sp is not None else None,
                   "psd_packed_intermediate": int(sp_tune[3]) if sp is not None else None,
                   "env": {k: os.environ[k] for k in TUNING_ENV if k in os.environ},
                   "argv": " ".join(sys.argv[1:])},
        "source_sha256": {s: source_sha(s) for s in ("mixdec.hip", "mixdec_mfma.hip", "psdfft.hip", "stage2.hip", "api.hip")},
    }
    if duty is not None:
        out["at_reference_psd_duty"] = duty
    if verify is not None:
        out["verified_ranks"] = verify["verified_ranks"]
        out["verify_worst_rel"] = verify["worst_rel"]
        out["verify"] = dict(tol=VERIFY_TOL, what=f"last timed step's device buffers, first {min(VERIFY_CHUNKS, B)} chunks of every "
                             "sub-receiver (audio + baseband IQ + per-chunk counts) and PSD frame 0 against the float32 "
                             "oracle primed over the chunks in front of them"
                             + ("; every rank's broadcast batch against the root's by 64-bit checksum" if split_rx else ""),
                             ranks=verify["ranks"])
    if ablated:
        out["invalid"] = "PYSDR_DEBUG_FLAGS != 0 in a diagnostic build: work was skipped, no roofline is reported"

    # PCIe-inclusive figure (never `value`): the same chunks handed over as HOST arrays through the
    # ingest ring (N4), slots of 16 chunks, including the host memcpy that stands for readStream()
    if rank == 0 and world == 1 and not args.no_host_fed and not split_rx and rxs and not is_wfm:
        out["host_fed"] = host_fed_rate(ctx, cfg, L, 16 if B >= 16 else B)
        out["host_fed_ms_per_chunk"] = out["host_fed"]["ms_per_chunk"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], used = cpu_baseline(cfg, args.cpu_chunks, with_psd, 10, args.cpu_seconds)
        if not args.no_cpu_mp and not is_wfm:
            out["cpu_baseline_per_rx_process"] = cpu_baseline_per_rx(args, cfg, used, with_psd, 10)
    elif rank == 0:
        out["cpu_baseline"] = None
        out["multi_rank_note"] = ("cpu_baseline is measured at N=1 only; kernel_ms and the roofline objects of a multi-rank line "
                                  "are rank 0's (per_rank_ms has every rank's step time, value the whole job's)")

    if bc is not None:
        bc.close()
    if sp is not None:
        lib.pysdr_spectrum_destroy(sp)
        lib.pysdr_dev_free(device, d_psd)
    lib.pysdr_dev_free(device, d_x)
    if need_shifted:
        lib.pysdr_dev_free(device, d_x1)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if (rank == 0 and world == 1 and args.workload == "c3" and not args.no_other_configs and not args.no_demod
            and not args.nrx and not args.no_psd and not args.chunks and not args.ntaps):
        out["other_configs"] = other_configs(args)
        if not args.no_live_traffic:
            lt = live_traffic(args)
            if lt:
                src = "live: two rocprofv3 --pmc child passes of this command, 3 steps each (2*FETCH_SIZE + WRITE_SIZE)"
                psd_live = lt.get("psd_cols_pk_kernel", 0.0) + lt.get("psd_rows_pk_kernel", 0.0)
                for key in ("roofline", "roofline_psd", "roofline_mixdec"):
                    r = out.get(key)
                    if not r:
                        continue
                    if r["kernel"].startswith("psd") and psd_live > 0 and psd_pairs_per_call:
                        r["traffic"], r["traffic_source"] = psd_live * psd_pairs_per_call, src + "; per launch pair, scaled to the call"
                    elif r["kernel"].startswith("mixdec_kernel") and lt.get("mixdec_kernel", 0.0) > 0:
                        r["traffic"], r["traffic_source"] = lt["mixdec_kernel"], src
                out["live_traffic_bytes_per_launch"] = {k: round(v) for k, v in sorted(lt.items())}
    if rank == 0:
        try:                     # RCCL prints a version banner through C stdio: the JSON stays the LAST line
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            full_path = write_full(out, args)
            print(json.dumps(compact_line(out, full_path), separators=(",", ":")), flush=True)
    if verify is not None and verify["verified_ranks"] != world:
        print(f"bench.py: --verify FAILED on rank {rank}: {verify['verified_ranks']} of {world} ranks match the oracle "
              f"(worst {verify['worst_rel']:.3g}, tol {VERIFY_TOL:g})", file=sys.stderr)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
