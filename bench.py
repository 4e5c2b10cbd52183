#!/usr/bin/env python3
"""Headline benchmark: real-time factor of the demix hot path on MI355X.

Metric (BASELINE.json): audio-seconds demixed / wall-seconds, 44.1 kHz stereo, offline
model, timed around the `separator(audio)` call with the model resident and warm, file
I/O excluded (the reference's own convention, xumx_slicq_v2/inference.py:28-31).

Workload = BASELINE.json configs[1]: offline conv stack (Bark-262 sliCQT), ONE 240 s track
(10,584,000 samples = 4 full 59.4 s chunks + a 98,240-sample tail), Wiener off (mix-phase),
seeded synthetic audio and seeded synthetic weights (no dataset / checkpoint offline).
A "step" = one full pass sliCQT -> CDAE -> phasemix -> isliCQT over that track, input already
resident in HBM.  With --gpus N (one process per GPU under torch.distributed.run, RCCL)
every rank demixes its own track per step -> weak scaling; tracks are independent objects, so
there is no data-path collective (only the timing barrier / max-reduce); --gather adds the
RCCL all-gather of all stems to all ranks for reference.  value = total audio-s / max-rank time.

One JSON line on stdout from rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRACK_SAMPLES = 10_584_000        # 240 s at 44.1 kHz
CHUNK = 2_621_440
FS = 44100.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (split-bf16: three MFMAs per product)


def algorithmic_work(plan, B, chunk_lengths, wiener):
    """Per-kernel ALGORITHMIC work of one pass over the given chunks: name -> (bound, amount)
    in bytes (hbm) or flops (mfma).  Per-unit figures are SURVEY.md 8(d): per channel-slice
    sliCQT = read 9030*4 + write 18640*8 B; CDAE flops formula; Wiener 160 B per TF point."""
    L, nbins, sumFT = plan.L, plan.L // 2 + 1, plan.coefs_per_slice
    Lg = plan.Lg.astype("int64")
    w = {}

    def add(k, bound, v):
        w[k] = (bound, w.get(k, (bound, 0))[1] + v)

    from xumx_slicq_amd.weights import freq_filter
    for n in chunk_lengths:
        n = max(n, L // 2 + 1)
        S = plan.num_slices(n)
        T1, T2 = 2 * S - 1, 2 * S - 4
        r2, r8 = 2 * B * S, 8 * B * S
        # hand-written LDS slice FFTs (window / band-spectrum gather fused in)
        add("slice_rfft", "hbm", 2 * B * n * 4 + r2 * nbins * 8)
        add("slice_irfft", "hbm", r8 * sumFT * 8 + r8 * L * 4)
        # rocFFT fallback path (other plans)
        add("slice_window", "hbm", 2 * B * n * 4 + r2 * L * 4)
        add("rfft_L", "hbm", r2 * L * 4 + r2 * nbins * 8)
        # per-band DFTs: bands with Lg >= 64 on the radix-4 kernel (2*M*Lg^2 flops: four m-point DFTs),
        # the short ones on the dense GEMM (8*M*Lg^2)
        long_, short_ = Lg[Lg >= 64], Lg[Lg < 64]
        add("band_analysis_dft4", "mfma", r2 * 2 * int((long_ * long_).sum()))
        add("band_analysis_gemm", "mfma", r2 * 8 * int((short_ * short_).sum()))
        add("magnitude_whiten", "hbm", r2 * sumFT * 12)
        f1 = f2 = f3 = f4 = 0
        for (_, F, T) in plan.blocks:
            kf = freq_filter(F)
            F1, F2 = F - kf + 1, F - 2 * kf + 2
            f1 += 2 * B * F1 * T1 * (2 * kf * T) * 50 * 4
            f2 += 2 * B * F2 * T2 * (50 * kf * 4) * 51 * 4
            f3 += 2 * B * F1 * T1 * (51 * kf * 4) * 50 * 4
            f4 += 2 * B * F1 * T1 * 50 * (2 * kf * T) * 4
        add("cdae_l1_gemm", "mfma", f1)
        # layers 2/3 of long inputs run on the slab kernels (csrc/cdae_slab.h: T >= 86), short ones on the generic engine
        add("cdae_l2_slab" if T2 >= 86 else "cdae_l2_gemm", "mfma", f2)
        add("cdae_l3_slab" if T1 >= 86 else "cdae_l3_gemm", "mfma", f3)
        add("cdae_l4_gemm", "mfma", f4)
        add("band_synthesis_dft4", "mfma", r8 * 2 * int((long_ * long_).sum()))
        add("band_synthesis_gemm", "mfma", r8 * 8 * int((short_ * short_).sum()))
        add("spectrum_gather", "hbm", r8 * sumFT * 8 + r8 * nbins * 8)
        add("irfft_L", "hbm", r8 * nbins * 8 + r8 * L * 4)
        add("overlap_add", "hbm", r8 * L * 4 + 8 * B * n * 4)
        if wiener:
            add("wiener_stats", "hbm", B * S * sumFT * 80)
            add("wiener_apply", "hbm", B * S * sumFT * 144)
    return w


# event name -> the kernel launched under it, as named by tools/summarize_profiles.py in profiles/*_kernel_stats.csv
_PMC_NAMES = {"cdae_l1_gemm": ["gemm<CdaeL1Op>"], "cdae_l2_gemm": ["gemm<CdaeL2Op>"], "cdae_l3_gemm": ["gemm<CdaeL3Op>"],
              "cdae_l2_slab": ["slab<CdaeL2>"], "cdae_l3_slab": ["slab<CdaeL3>"],
              "cdae_l4_gemm": ["gemm<CdaeL4Op>"], "band_synthesis_gemm": ["gemm<BandInvOp>"],
              "band_analysis_gemm": ["gemm<BandFwdOp>"], "band_synthesis_dft4": ["band_dft4<inverse>"],
              "band_analysis_dft4": ["band_dft4<forward>"], "slice_irfft": ["k_slice_irfft"], "slice_rfft": ["k_slice_rfft"],
              "overlap_add": ["k_overlap_add"], "magnitude_whiten": ["k_magnitude_whiten"]}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/collect_profiles.sh -> tools/summarize_profiles.py; FETCH_SIZE doubled for gfx950 as
    MI355X_MICROARCH.md prescribes, WRITE_SIZE as is; both are KB per dispatch, averaged over the
    launches of a step).  None when no profile of the current kernels is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv")))
    if not files or kernel not in _PMC_NAMES:
        return None
    total, launches = 0.0, 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if row["Kernel"] in _PMC_NAMES[kernel]:
                try:
                    n = int(float(row["launches"]))
                    total += n * (2.0 * float(row["fetch_KB_mean_raw"]) + float(row["write_KB_mean_raw"])) * 1024
                    launches += n
                except (KeyError, ValueError):
                    return None
    return int(total / launches) if launches else None


def cpu_baseline(threads):
    """The CPU oracle (a port of the reference, pinned to it by tests/golden) timed on this
    box's host cores on a bounded sample: one 30 s clip through the same configuration."""
    from oracle import separator as osep
    from oracle import slicqt as oslicqt
    from xumx_slicq_amd.synth import synth_audio
    from xumx_slicq_amd.weights import seeded_state_dict
    torch.set_num_threads(threads)
    plan = oslicqt.make_plan()
    sd = seeded_state_dict([(F, T) for (_, F, T) in plan.blocks])
    n = 30 * 44100
    x = synth_audio(n)
    osep.separate(plan, sd, x[..., :44100], causal=False, wiener=False)   # warm
    t0 = time.perf_counter()
    osep.separate(plan, sd, x, causal=False, wiener=False)
    dt = time.perf_counter() - t0
    return {"value": round(n / FS / dt, 3), "unit": "x real-time", "cores": threads, "kind": "port",
            "sample": "30 s stereo clip (1,323,000 samples), offline conv stack + mix-phase, oracle/ torch-CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--wiener", action="store_true", help="BASELINE configs[2]: Wiener-EM on (default off = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x6", "bf16x3"],
                    help="arithmetic of the convolution contractions for the headline value (default: exact fp32)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra split-bf16 measurement")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph (Separator.forward_graphed)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: all-gather every track's stems to every rank inside the timed region")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with: python -m torch.distributed.run --nnodes=1 "
                     "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP library is the product path and there is no CPU fallback")
    local_dev = local_rank % torch.cuda.device_count()     # (several ranks per GPU only in smoke tests)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("XSQ_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.separator import seeded_separator
    from xumx_slicq_amd.sharding import chunk_items, demix_tracks
    from xumx_slicq_amd.synth import synth_audio

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # keep stdout to the one JSON line
        sep = seeded_separator(realtime=False, wiener=args.wiener, device=dev, chunk_size=CHUNK)
    sep.xumx_model.set_precision(args.precision)
    # inputs are resident in HBM before timing starts; without --gather a rank only ever touches
    # its own track, so only that one is materialised
    if args.gather:
        tracks = [synth_audio(TRACK_SAMPLES, seed=20260101 + t).to(dev) for t in range(world)]
    else:
        mine = synth_audio(TRACK_SAMPLES, seed=20260101 + rank).to(dev)
        tracks = [mine if t == rank else torch.empty(1, 2, TRACK_SAMPLES, device="meta")   # shape only
                  for t in range(world)]

    def step():
        run = sep.forward_graphed if args.graph else sep
        if world == 1:
            return run(tracks[0])
        return demix_tracks(run, tracks, gather=args.gather)

    # Warm-up steps: every kernel is timed with HIP events on its launch stream (the per-kernel table and the
    # choice of the dominant kernel).  Timed region: only the dominant kernel keeps its two events per launch --
    # event records around all 24 launches of a step cost ~0.1 ms of it (tools/prof_overhead.py).
    _lib.profile_filter(None)
    _lib.profile_enable(True)
    for i in range(args.warmup):
        if i == args.warmup - 1:
            torch.cuda.synchronize()
            _lib.profile_reset()        # the table comes from the LAST warm-up step (the first one builds tile tables etc.)
        step()
    torch.cuda.synchronize()
    prof_all = _lib.profile_read() if args.warmup >= 2 else None
    nwarm = 1
    dom = max(prof_all, key=lambda k: prof_all[k][0]) if prof_all else None
    _lib.profile_filter(dom)            # None (fewer than two warm-up steps): every kernel stays instrumented in the timed region
    _lib.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = _lib.profile_read()          # the dominant kernel only, over the timed region
    _lib.profile_enable(False)
    _lib.profile_filter(None)
    if prof_all is None:                # no warm-up step to take the table from
        prof_all, nwarm = prof, args.steps
        dom = max(prof, key=lambda k: prof[k][0]) if prof else None

    # Extra, outside the timed region, N = 1 only: the same step with the convolution contractions on the
    # split-bf16 matrix path (xsq_model_set_precision 1), and its stems against the fp32 stems just produced.
    variants = None
    if world == 1 and not args.no_variants and args.precision == "fp32":
        ref_out = out.clone()
        variants = {}
        what = {"bf16x6": "conv contractions as 6 x bf16 MFMA on fp32 operands cut exactly into three bf16 pieces (dropped terms <= 2^-23 |ab|: fp32-grade), fp32 accumulate; everything else unchanged",
                "bf16x3": "conv contractions as 3 x bf16 MFMA on hi/lo-split fp32 operands (~2^-17 per product), fp32 accumulate; everything else unchanged"}
        for prec in ("bf16x6", "bf16x3"):
            sep.xumx_model.set_precision(prec)
            for _ in range(max(1, args.warmup)):
                step()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(args.steps):
                vout = step()
            torch.cuda.synchronize()
            tv = time.perf_counter() - tv
            d = (vout - ref_out).double()
            variants[prec] = {
                "what": what[prec],
                "value": round(args.steps * TRACK_SAMPLES / FS / tv, 2), "ms_per_step": round(tv / args.steps * 1e3, 3),
                "stems_vs_fp32": {"rms": float(d.pow(2).mean().sqrt()), "max_abs": float(d.abs().max()),
                                  "bar": "1e-4 rms / 1e-3 max-abs (BASELINE.json north_star)"}}
            del vout, d
        sep.xumx_model.set_precision("fp32")
        del ref_out
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    result = None
    if rank == 0:
        audio_s = world * args.steps * TRACK_SAMPLES / FS
        # roofline of the dominant kernel (largest share of the timed region on this rank)
        plan = sep.nsgt.nsgt.plan
        my_items = [it.length for it in chunk_items([TRACK_SAMPLES], CHUNK)]
        work = algorithmic_work(plan, 1, my_items, args.wiener)
        roofline = None
        if dom is not None:
            ms, launches = prof[dom]
            bound, amount = work[dom]
            per_launch = amount * args.steps / launches          # algorithmic work per launch
            avg_s = ms / launches * 1e-3
            if bound == "hbm":
                ach, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = per_launch / avg_s / 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s"
                if args.precision != "fp32" and dom.startswith("cdae_"):      # useful flops at 3 / 6 bf16 MFMAs per product
                    peak = round(BF16_MFMA_PEAK_TFLOPS / (3.0 if args.precision == "bf16x3" else 6.0), 1)
            roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom),
                        "avg_launch_ms": round(ms / launches, 4), "launches": launches,
                        "share_of_step": round(ms / (dt * 1e3), 4)}
        kernels = {k: {"ms_per_step": round(v[0] / nwarm, 4), "launches_per_step": v[1] / nwarm}
                   for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}       # from the instrumented warm-up steps
        result = {
            "metric": "real-time factor (audio-s demixed / wall-s), 44.1 kHz stereo, offline model",
            "value": round(audio_s / dt, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x6": "f32 (conv contractions: exact 3-way bf16 cut, 6 bf16 MFMAs per product, fp32 accumulate)",
                      "bf16x3": "f32 (conv contractions as 3 x bf16 MFMA, fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: offline model (Bark-262 sliCQT), one 240 s stereo track "
                                   "(10,584,000 samples, 5 chunks) per GPU, %s, seeded synthetic weights"
                                   % (2 if args.wiener else 1, "norbert Wiener-EM niter=1" if args.wiener else "Wiener off (mix-phase)"),
                       "parallelism": "one track per rank, %d rank(s), %s" % (
                           world, "RCCL all-gather of stems" if (args.gather and world > 1) else "no data-path collective")},
            "roofline": roofline,
            "kernels_source": "warm-up steps (all kernels instrumented); the timed region instruments the roofline kernel only",
            "kernels": kernels,
        }
        if variants:
            result["variants"] = variants
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
