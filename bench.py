#!/usr/bin/env python3
"""Throughput of the pySDR receiver hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one device-resident batch of `--chunks` chunks of
one synthetic 8 MS/s wideband stream: the fused mix+decimate kernel for all 4 sub-receivers
(USB/CW/NBFM/AM), the 48 kHz detector/AF/AGC kernels, and the RF
PSD (chunk 32768 -> 64k FFT, every sample PSD'd) on its own HIP stream.  With N GPUs every rank
runs its own independent stream (config C5: shard by stream, no data-path collective; weak
scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PSD_CHUNK, PSD_NFFT = 32768, 65536


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3"])
    ap.add_argument("--chunks", type=int, default=2048, help="chunks per step (batch resident in HBM)")
    ap.add_argument("--no-psd", action="store_true")
    ap.add_argument("--no-cpu-mp", action="store_true", help="skip the one-process-per-RX CPU figure")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=10, help=argparse.SUPPRESS)
    ap.add_argument("--overlap-psd", action="store_true", help="PSD on its own stream, unordered w.r.t. the demod")
    ap.add_argument("--serial-psd", action="store_true", help="PSD strictly behind the whole demod (incl. stage 2)")
    ap.add_argument("--no-demod", action="store_true", help="diagnostic: PSD only")
    ap.add_argument("--tile-bytes", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunks", type=int, default=640, help="chunks timed on the CPU oracle (about 10 s)")
    return ap.parse_args()


def build_receivers(cfg, device, max_chunks):
    from pysdr_amd import sig_proc
    from pysdr_amd.params import RunTimeParams
    r0 = cfg['rx'][0]
    P = RunTimeParams(fs=cfg['fs'], fsout=cfg['fs_out'], fc=[14.2e6] * len(cfg['rx']),
                      mode=r0['mode'], nfilt=cfg['ntaps_dec'], device=device,
                      max_batch_chunks=max_chunks)
    rxs = []
    for i, r in enumerate(cfg['rx']):
        P.VIDEO_BW = r.get('video_bw', 10e3)
        rx = sig_proc.Receiver(P, r['frq'], i, str(i + 1))
        rx.mode, rx.af_bw, rx.bfo = r['mode'], r.get('af_bw', 0.0), r.get('bfo', 0.0)
        rxs.append(rx)
    P.rx = rxs
    for rx in rxs:
        rx._sync_controls()
    return P, rxs


def cpu_baseline(cfg, nchunks, with_psd, seed):
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
    x = so.synth_iq(cfg, uniq * L, seed)
    rxs = so.make_receivers(cfg, np.float32)
    sp = so.Spectrum(cfg['fs'] / 1e3, PSD_CHUNK, PSD_NFFT, 0.0, np.float32)
    for rx in rxs:                       # warm-up chunk (BLAS init, page faults)
        rx.demod_data(x[:L])
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
                       f"{dt:.1f} s wall; host has {os.cpu_count()} cores")


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
    x = so.synth_iq(cfg, uniq * L, seed)
    if irx >= 0:
        rx = so.make_receivers(cfg, np.float32)[irx]
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


def cpu_baseline_per_rx(workload, cfg, nchunks, with_psd, seed):
    """The analogue of the reference's MP_SCHEME 3 (one process per sub-receiver,
    receiver.py:726-739): NUM_RX (+1 for the PSD) single-threaded child processes
    (`bench.py --cpu-worker`, plain subprocesses with a deadline: the parent holds a HIP context
    and must neither fork it nor ever wait forever) over the same sample; the job's rate is set
    by the slowest of them.  Returns None if a child fails."""
    import subprocess
    from oracle import sdr_oracle as so
    L = so.chunk_sizes(cfg['fs'], cfg['fs_out'])[3]
    ids = list(range(len(cfg['rx']))) + ([-1] if with_psd else [])
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--workload", workload,
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


def measured_traffic(args, nrx, B):
    """HBM bytes per mix+decimate launch from the committed PMC passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE), valid only for the configuration that was profiled."""
    if args.workload != "c3" or B != 2048 or nrx != 4:
        return None
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        return [r["hbm_bytes_per_launch"] for r in rows if r["kernel"] == "mixdec_kernel"][0]
    except Exception:
        return None


def main():
    args = parse()
    if args.cpu_worker is not None:      # child of cpu_baseline_per_rx: CPU only, never touches the GPU
        from pysdr_amd.synth import CONFIGS
        print(_cpu_worker((CONFIGS[args.workload.upper()], args.cpu_chunks, args.cpu_seed, args.cpu_worker)))
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # Load the HIP library BEFORE torch so that both share one libamdhip64.
    from pysdr_amd import _lib
    from pysdr_amd.synth import CONFIGS, synth_iq
    lib = _lib.lib()
    _lib.require_gpu()
    ndev = _lib.device_count()
    device = local_rank % ndev

    dist = None
    if world > 1:
        import torch.distributed as dist      # gloo: barrier + MAX only, no GPU tensors
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    cfg = CONFIGS[args.workload.upper()]
    with_psd = (args.workload == "c3") and not args.no_psd
    B = args.chunks
    P, rxs = build_receivers(cfg, device, B)
    ctx = P._pysdr_stream
    L = P.IN_CHUNK_SIZE
    nsamp = B * L
    if args.tile_bytes or args.threads:
        _lib.check(lib.pysdr_set_tile(ctx.h, args.tile_bytes, args.threads or 1024), "set_tile")

    # synthetic stream: 8 unique chunks (seed per rank = its own stream), repeated to fill the batch
    uniq = 8
    xu = synth_iq(cfg, uniq * L, 10 + rank)
    d_x = C.c_void_p()
    _lib.check(lib.pysdr_dev_alloc(device, nsamp * 8, C.byref(d_x)), "alloc x")
    for k in range(0, B, uniq):
        n = min(uniq, B - k) * L
        _lib.check(lib.pysdr_dev_upload(device, C.c_void_p(d_x.value + k * L * 8),
                                        C.c_void_p(xu.ctypes.data), n * 8), "upload")

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
        if not args.no_demod:
            ctx.process_batch(d_x.value, B, L, on_device=True)
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
    _lib.check(lib.pysdr_set_profile(ctx.h, 1), "profile")

    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0

    # dominant kernel (fused mix+decimate): HIP events on its stream, averaged over the timed steps
    nev = 0 if args.no_demod else min(args.steps, 64)
    ms = C.c_float(0)
    k1 = []
    k2 = []
    for back in range(nev):
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 0, back, C.byref(ms)), "elapsed")
        k1.append(ms.value)
        _lib.check(lib.pysdr_get_elapsed_ms(ctx.h, 1, back, C.byref(ms)), "elapsed")
        k2.append(ms.value)
    k1_ms = float(np.mean(k1)) if k1 else float('nan')
    psd_ms = None
    if sp is not None:
        _lib.check(lib.pysdr_spectrum_elapsed_ms(sp, C.byref(ms)), "psd elapsed")
        psd_ms = ms.value

    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    nrx = len(rxs)
    n_out = nsamp * P.UP // P.DOWN
    # algorithmic bytes of ONE mix+decimate launch: every input sample once (8 B, shared by all
    # RX) + the baseband IQ it writes (8 B per RX per output)   [SURVEY.md 8(d), DESIGN.md 5]
    k1_bytes = nsamp * 8 + nrx * n_out * 8
    achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
    bytes_per_sample_job = 8.0 + nrx * (P.UP / P.DOWN) * 12.0 + (4.0 * PSD_NFFT / PSD_CHUNK if with_psd else 0.0)

    out = {
        "metric": "complex IQ MS/s through 4-RX demod chain",
        "value": world * nsamp * args.steps / dt / 1e6,
        "unit": "MS/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.workload.upper()}: 8 MS/s synthetic IQ, {nrx} RX "
                         f"({'/'.join(r['mode'] for r in cfg['rx'])}), {cfg['ntaps_dec']}-tap polyphase 3/500"
                         f"{', + 64k-FFT RF PSD on every sample' if with_psd else ''}; "
                         f"{B} chunks x {L} samples per step resident in HBM; one stream per GPU"),
            "chunks_per_step": B, "in_chunk": L, "samples_per_step": nsamp,
            "parallelism": f"stream-sharded x{world} (no data-path collective)",
        },
        "roofline": {
            "kernel": "mixdec_kernel<%d> (fused NCO mix + polyphase decimate, all RX)" % nrx,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": measured_traffic(args, nrx, B),
            "traffic_source": "profiles/r01_pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE of mixdec_kernel, separate rocprofv3 --pmc passes of this same command (null when the config differs from the profiled one)",
            "algorithmic_bytes_per_launch": k1_bytes,
            "avg_launch_ms": k1_ms,
        },
        "kernel_ms": {"mixdec": k1_ms, "stage2": float(np.mean(k2)) if k2 else None, "psd_last": psd_ms},
        # the spectral path, same accounting: one call = nframes frames, each reads chunk complex
        # samples and writes nfft dB values (the 512 KB/frame four-step intermediate is traffic,
        # not algorithmic bytes)
        "roofline_psd": None if psd_ms is None else {
            "kernel": "psd_cols_kernel + psd_rows_kernel (window, zero-pad, 64k FFT, |.|^2, dB, fftshift)",
            "bound": "hbm",
            "achieved": nframes * (PSD_CHUNK * 8 + PSD_NFFT * 4) / (psd_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": nframes * (PSD_CHUNK * 8 + PSD_NFFT * 4) / (psd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_call": nframes * (PSD_CHUNK * 8 + PSD_NFFT * 4),
            "avg_call_ms": psd_ms,
        },
        "job_bytes_per_sample": bytes_per_sample_job,
        "job_hbm_frac_per_gpu": (nsamp * args.steps / dt) * bytes_per_sample_job / 1e9 / HBM_PEAK_GBPS,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_chunks, with_psd, 10)
        if not args.no_cpu_mp:
            out["cpu_baseline_per_rx_process"] = cpu_baseline_per_rx(args.workload, cfg, args.cpu_chunks, with_psd, 10)
    elif rank == 0:
        out["cpu_baseline"] = None

    if sp is not None:
        lib.pysdr_spectrum_destroy(sp)
        lib.pysdr_dev_free(device, d_psd)
    lib.pysdr_dev_free(device, d_x)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
