#!/usr/bin/env python3
"""Headline benchmark: real-time factor of the demix hot path on MI355X.

Metric (BASELINE.json): audio-seconds demixed / wall-seconds, 44.1 kHz stereo, offline
model, timed around the `separator(audio)` call with the model resident and warm, file
I/O excluded (the reference's own convention, xumx_slicq_v2/inference.py:28-31).

N = 1 (default), workload "track240" = BASELINE.json configs[1]: offline conv stack (Bark-262
sliCQT), ONE 240 s track (10,584,000 samples = 4 full 59.4 s chunks + a 98,240-sample tail),
Wiener off (mix-phase), seeded synthetic audio and weights (no dataset / checkpoint offline).
A "step" = one full pass sliCQT -> CDAE -> phasemix -> isliCQT over that track, input already
resident in HBM.  The same line carries, outside the timed region: the per-kernel HBM / MFMA
roofline fractions, configs[2] (Wiener-EM on) and configs[4] (the B = 16 training step) as
`variants`, the split-bf16 arithmetic variants, and the CPU oracle on this box's host cores.

N > 1, workload "testset50" = BASELINE.json configs[3]: 50 seeded track lengths in [150, 420] s
flattened into (track, chunk) work items, dealt longest-first to the ranks (one process per GPU,
RCCL), every rank runs its full-size chunks stacked along the batch axis, and the stems of every
item are all-gathered to every rank and placed into per-track (4, 1, 2, N_t) tensors -- the final
waveform concat of separator.py:231.  A "step" = the whole 50-track set once (fixed total work:
"scaling": "strong").  value = total audio-s / max-rank time.  Beside it (`variants`): the same
step without the collective, and the whole set on rank 0 alone (the single-GPU rate on the SAME
workload, for an efficiency figure that does not mix workloads).

`python3 bench.py --gpus N` (no launcher) starts its own ranks: a fresh
`python -m torch.distributed.run` child BEFORE anything touches the GPU, rank 0's JSON line is
relayed, the child's exit status is returned.  Under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE from the environment.  One JSON line on stdout from rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRACK_SAMPLES = 10_584_000        # 240 s at 44.1 kHz
CHUNK = 2_621_440
FS = 44100.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (split-bf16: three MFMAs per product)
TESTSET_SEED, TESTSET_TRACKS = 20260104, 50
METRIC = "real-time factor (audio-s demixed / wall-s), 44.1 kHz stereo, offline model"


def testset_lengths(ntracks=TESTSET_TRACKS):
    """a fixed seeded list of durations in [150 s, 420 s] (the reference does not state
    the MUSDB18-HQ test-set durations)."""
    import numpy as np
    rng = np.random.default_rng(TESTSET_SEED)
    return [int(d * FS) for d in rng.uniform(150.0, 420.0, ntracks)]


def algorithmic_work(plan, B, chunk_lengths, wiener):
    """Per-kernel ALGORITHMIC work of one pass over the given chunks: name -> (bound, amount)
    in bytes (hbm) or flops (mfma).  Per-unit figures are SURVEY.md 8(d): per channel-slice
    sliCQT = read 9030*4 + write 18640*8 B; CDAE flops formula; Wiener-EM 48 + 112 B per TF point (from the masks)."""
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
        # hand-written LDS slice FFTs (window / band-spectrum gather / overlap-add fused in)
        add("slice_rfft", "hbm", 2 * B * n * 4 + r2 * nbins * 8)
        add("slice_irfft", "hbm", r8 * sumFT * 8 + r8 * L * 4)
        add("slice_irfft_ola", "hbm", r8 * sumFT * 8 + 8 * B * n * 4)
        # rocFFT fallback path (other plans)
        add("slice_window", "hbm", 2 * B * n * 4 + r2 * L * 4)
        add("rfft_L", "hbm", r2 * L * 4 + r2 * nbins * 8)
        # per-band DFTs: bands with Lg >= 24 (XSQ_D4_MIN_LG_DEFAULT, csrc/slicqt.hip) on the radix-4 kernel (2*M*Lg^2 flops: four m-point DFTs),
        # the short ones on the dense GEMM (8*M*Lg^2)
        split = int(os.environ.get("XSQ_D4_MIN_LG", 24))
        long_, short_ = Lg[Lg >= split], Lg[Lg < split]
        add("band_analysis_dft4", "mfma", r2 * 2 * int((long_ * long_).sum()))
        add("band_analysis_gemm", "mfma", r2 * 8 * int((short_ * short_).sum()))
        add("magnitude_whiten", "hbm", r2 * sumFT * 12)
        f1 = f2 = f4 = 0
        for (_, F, T) in plan.blocks:
            kf = freq_filter(F)
            F1, F2 = F - kf + 1, F - 2 * kf + 2
            f1 += 2 * B * F1 * T1 * (2 * kf * T) * 50 * 4
            f2 += 2 * B * F2 * T2 * (50 * kf * 4) * 51 * 4
            f4 += 2 * B * F1 * T1 * 50 * (2 * kf * T) * 4
        add("cdae_l1_gemm", "mfma", f1)
        # layers 2/3 of long inputs run on the slab kernels (csrc/cdae_slab.h: T >= 86), short ones on the generic engine.
        # Layer 3 is a ConvTranspose: its algorithmic count is per INPUT position (F2 x T2 x 51 x 50 x kf x 4 MACs,
        # SURVEY.md 8(d)) = layer 2's; the gather form the kernel runs also visits the zero-padded border taps
        # (F1 x T1 output rows), which are not counted here.
        add("cdae_l2_slab" if T2 >= 86 else "cdae_l2_gemm", "mfma", f2)
        add("cdae_l3_slab" if T1 >= 86 else "cdae_l3_gemm", "mfma", f2)
        add("cdae_l4_gemm", "mfma", f4)
        add("band_synthesis_dft4", "mfma", r8 * 2 * int((long_ * long_).sum()))
        add("band_synthesis_gemm", "mfma", r8 * 8 * int((short_ * short_).sum()))
        add("spectrum_gather", "hbm", r8 * sumFT * 8 + r8 * nbins * 8)
        add("irfft_L", "hbm", r8 * nbins * 8 + r8 * L * 4)
        add("overlap_add", "hbm", r8 * L * 4 + 8 * B * n * 4)
        if wiener:
            # per time-frequency point (both channels, four targets).  From the masks (default, xsq_wiener_em_masked):
            # pass 1 reads 8 masks + 2 mix values = 48 B, pass 3 reads the same and writes 8 estimates = 112 B.  The
            # two-step form (XSQ_WIENER_MASKED=0) reads the 8 complex initial estimates instead: 80 B / 80 + 64 B.
            masked = os.environ.get("XSQ_WIENER_MASKED", "1") != "0"
            add("wiener_stats", "hbm", B * S * sumFT * (48 if masked else 80))
            add("wiener_apply", "hbm", B * S * sumFT * (112 if masked else 144))
    return w


# event name -> the kernel launched under it, as named by tools/summarize_profiles.py in profiles/*_kernel_stats.csv
_PMC_NAMES = {"cdae_l1_gemm": ["l1f<CdaeL1>", "gemm<CdaeL1Op>"], "cdae_l2_gemm": ["gemm<CdaeL2Op>"], "cdae_l3_gemm": ["gemm<CdaeL3Op>"],
              "cdae_l2_slab": ["wino<CdaeL2>", "slab<CdaeL2>"], "cdae_l3_slab": ["wino<CdaeL3>", "slab<CdaeL3>"],
              "cdae_l4_gemm": ["l4f<CdaeL4>", "gemm<CdaeL4Op>"], "band_synthesis_gemm": ["gemm<BandInvOp>"],
              "band_analysis_gemm": ["gemm<BandFwdOp>"], "band_synthesis_dft4": ["band_dft4s<inverse>", "band_dft4<inverse>"],
              "band_analysis_dft4": ["band_dft4s<forward>", "band_dft4<forward>"], "slice_irfft": ["k_slice_irfft"], "slice_rfft": ["k_slice_rfft"],
              "slice_irfft_ola": ["k_slice_irfft"],
              "overlap_add": ["k_overlap_add"], "magnitude_whiten": ["k_magnitude_whiten"],
              "wiener_stats": ["k_wiener_stats_masked", "k_wiener_stats"], "wiener_apply": ["k_wiener_apply_masked", "k_wiener_apply"],
              "place_rows": ["k_place_rows"]}


def pmc_traffic(kernel, tag=None):
    """(HBM bytes per STEP of `kernel` -- all its launches of one 240 s track --, launches per step) from the
    committed rocprofv3 PMC passes of this same command (tools/collect_profiles.sh -> tools/summarize_profiles.py;
    FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes, WRITE_SIZE as is; both are KB per dispatch
    averaged over the profiled launches; the profiled run holds `steps_profiled` steps).  None when no profile
    of the current kernels is committed."""
    import csv
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.csv"))
                   if (tag is None) == ("wiener" not in os.path.basename(f)))
    if not files or kernel not in _PMC_NAMES:
        return None
    total, launches, steps = 0.0, 0, 2
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if row["Kernel"] in _PMC_NAMES[kernel]:
                try:
                    n = int(float(row["launches"]))
                    total += n * (2.0 * float(row["fetch_KB_mean_raw"]) + float(row["write_KB_mean_raw"])) * 1024
                    launches += n
                    steps = int(float(row.get("steps_profiled") or 2))     # tools/collect_profiles.sh: --steps 1 --warmup 1
                except (KeyError, ValueError):
                    return None
    return (int(total / steps), launches / steps) if launches else None


def rocprof_launch_ms(kernel, tag=None):
    """Average launch duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/*_kernel_stats.csv, newest; tools/collect_profiles.sh traces it with the driver's --steps 20 --warmup 5): the
    figure a reader recomputes the roofline fraction from.  The bench line carries it beside the event time of the timed
    region instead of letting the two disagree silently (a 3 + 1 step trace, as the sets before r06z were, times its
    launches from cold and reads 10-15 % long).  None without a committed summary."""
    import csv
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv"))
                   if (tag is None) == ("wiener" not in os.path.basename(f)))
    if not files or kernel not in _PMC_NAMES:
        return None
    total, calls = 0.0, 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if row.get("Kernel") in _PMC_NAMES[kernel]:
                try:
                    total += float(row["TotalDurationNs"])
                    calls += int(float(row["Calls"]))
                except (KeyError, ValueError):
                    return None
    return (total / calls * 1e-6, os.path.basename(files[-1])) if calls else None


def pmc_issue(kernel, tag=None):
    """What the SQ counters of the committed rocprofv3 pass say bounds `kernel` (profiles/*_pmc_sq.csv, the same command
    as this bench): vector-ALU issue utilisation = 4 cycles x SQ_INSTS_VALU over the SIMD cycles of the kernel's busy
    time (SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs), matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES
    over the same base, and the share of wave cycles spent in s_waitcnt.  None without a committed profile."""
    import csv
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq.csv"))
                   if (tag is None) == ("wiener" not in os.path.basename(f)))
    if not files or kernel not in _PMC_NAMES:
        return None
    tot = {}
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if row["Kernel"] in _PMC_NAMES[kernel]:
                for c in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"):
                    try:
                        tot[c] = tot.get(c, 0.0) + float(row[c])
                    except (KeyError, ValueError):
                        return None
                break          # the summary repeats a kernel once per launch shape: the first row is the main launch
    if not tot or tot["SQ_BUSY_CYCLES"] <= 0 or tot["SQ_WAVE_CYCLES"] <= 0:
        return None
    return {"valu_issue": round(tot["SQ_INSTS_VALU"] / (8.0 * tot["SQ_BUSY_CYCLES"]), 3),
            "mfma_busy": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * tot["SQ_BUSY_CYCLES"]), 3),
            "waitcnt_share_of_wave_cycles": round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
            "source": os.path.basename(files[-1])}


SIMDS, PEAK_CLOCK_GHZ = 1024, 2.4     # 256 CUs x 4 SIMDs; the clock the fp32 MFMA peak of MI355X_MICROARCH.md is quoted at


def issue_bound(plan, B, chunk_lengths, winograd=True, l1f=True, l4f=True):
    """Per-kernel ISSUE BOUND of the fp32 MFMA kernels for one pass over the given chunks, in ms: on gfx950 an fp32 MFMA and the
    other vector instructions of a SIMD's waves take turns on one issue port (DESIGN.md "Vector issue": a loop's time is the SUM
    of its MFMA cycles and ~4 cycles per other vector instruction), so no schedule of the kernel AS COMPILED can beat
        sum over tiles and waves of (MFMA cycles + 4 x other vector instructions) / (1024 SIMDs x 2.4 GHz).
    The per-trip and outside-the-loops counts come from the assembly (profiles/isa_budget.json, tools/isa_budget.py); tile and
    trip counts are the host code's tilings restated here (csrc/cdae.hip get_*_tiles, csrc/slicqt.hip get_dft4_full_tiles).
    Structural assumptions, where one kernel holds several specialised bodies: layer 4's out-of-loop instructions are
    estimated (see there); the radix-4 kernel's out-of-loop count is its prologue plus one
    epilogue block per 16-column block (33 instructions for the synthesis; the rest pro rata for the analysis) and one
    peeled K-step.  Returns {event name: (bound_ms, mfma_cycle_share)} -- None without the committed budget."""
    import math
    path = os.path.join(ROOT, "profiles", "isa_budget.json")
    if not os.path.exists(path):
        return None
    K = json.load(open(path))["kernels"]
    from xumx_slicq_amd.weights import freq_filter
    Lg = plan.Lg.astype("int64")
    split = int(os.environ.get("XSQ_D4_MIN_LG", 24))
    tot = {}

    def add(name, waves, mfma, valu):
        m, v = tot.get(name, (0.0, 0.0))
        tot[name] = (m + waves * mfma, v + waves * 4.0 * valu)

    def engine(name, key, M, ncols_tiles, ksteps, waves=4):
        """generic tile engine: 128-row tiles, the loop of the assembly covers two 16-value K-steps"""
        k = K[key]
        lp = k["loops"][0]
        tiles = math.ceil(M / 128) * ncols_tiles
        add(name, tiles * waves, k["outside"]["mfma_cycles"] + lp["mfma_cycles"] * ksteps / 2.0, k["outside"]["valu"] + lp["valu"] * ksteps / 2.0)

    for n in chunk_lengths:
        n = max(n, plan.L // 2 + 1)
        S = plan.num_slices(n)
        T1, T2 = 2 * S - 1, 2 * S - 4
        for (_, F, T) in plan.blocks:
            kf = freq_filter(F)
            F1, F2 = F - kf + 1, F - 2 * kf + 2
            if l1f and "cdae_l1f" in K:
                # layer 1 as F(2, 2) along the hop (csrc/cdae_l1f.h): tiles of 64 output pairs (4 waves x 16) per target, chunks of
                # 16 k of the padded half-window order; the assembly's loop holds TWO chunks (72 MFMAs), an odd last chunk and
                # the prologue / epilogue are its out-of-loop part
                k = K["cdae_l1f"]
                lp = k["loops"][0]
                hop = T // 2
                nch = (2 * kf * ((hop + 3) // 4) + 3) // 4
                tiles = 4 * math.ceil(B * F1 * ((T1 + 1) // 2) / 64)
                tail_v = lp["valu"] / 2.0
                add("cdae_l1_gemm", tiles * 4, lp["mfma_cycles"] * (nch // 2) + (k["outside"]["mfma_cycles"] if nch & 1 else 0),
                    lp["valu"] * (nch // 2) + (k["outside"]["valu"] - (0 if nch & 1 else tail_v)))
            else:
                # layer 1: one 52-column tile per 128 rows and target; K = 2 kf T padded to 16
                engine("cdae_l1_gemm", "gemm<CdaeL1Op>", B * F1 * T1, 4, math.ceil(2 * kf * T / 16))
            # layers 2 / 3
            for name, To, Fo, key_w, key_s, key_g in (("cdae_l2", T2, F2, "cdae_wino<L2>", "cdae_slab<L2>", "gemm<CdaeL2Op>"),
                                                    ("cdae_l3", T1, F1, "cdae_wino<L3>", "cdae_slab<L3>", "gemm<CdaeL3Op>")):
                if winograd and (To + 1) // 2 >= 64:
                    k = K[key_w]
                    tiles = 4 * B * math.ceil(Fo * ((To + 1) // 2) / 64)
                    add(name + "_slab", tiles * 4, k["outside"]["mfma_cycles"] + kf * k["loops"][0]["mfma_cycles"], k["outside"]["valu"] + kf * k["loops"][0]["valu"])
                elif To >= 86:
                    k = K[key_s]
                    tiles = 4 * B * math.ceil(Fo * To / 256)
                    add(name + "_slab", tiles * 8, k["outside"]["mfma_cycles"] + kf * k["loops"][0]["mfma_cycles"], k["outside"]["valu"] + kf * k["loops"][0]["valu"])
                else:
                    engine(name + "_gemm", key_g, B * Fo * To, 4, kf * 13)
            if l4f and "cdae_l4f<4>" in K:
                # layer 4 as F(2, 2) along the hop (csrc/cdae_l4f.h): row tiles of 64 output pairs x one column tile of 16 NCB <= 64
                # columns; the four width classes are budgeted from probe kernels of their own (tools/isa_budget.py).  One tap: a
                # workgroup runs up to 6 consecutive row tiles against weights resident in LDS -- the assembly's loop with the
                # epilogue in it (the most vector instructions) is ONE row tile, the out-of-loop count is the workgroup's prologue.
                # Several taps (NCB <= 2): one row tile per workgroup; the other loop is one tap, prologue + epilogue ~ 300.
                cols = 16 * math.ceil(T / 16)
                rtiles = math.ceil(B * F * S / 64)
                for n0 in range(0, cols, 64):
                    ncb = min(4, (cols - n0) // 16)
                    k = K["cdae_l4f<%d>" % ncb]
                    tap_m = 39 * ncb * 32.0
                    run_lp = max(k["loops"], key=lambda lp: lp["valu"])
                    if kf == 1:
                        nruns = math.ceil(rtiles / 6)
                        add("cdae_l4_gemm", 4 * rtiles * 4, tap_m, run_lp["valu"])
                        add("cdae_l4_gemm", 4 * nruns * 4, 0.0, 250.0)
                    else:
                        tap_lp = min(k["loops"], key=lambda lp: lp["valu"])
                        add("cdae_l4_gemm", 4 * rtiles * 4, tap_m * kf, 300.0 + tap_lp["valu"] * kf)
                continue
            # layer 4: N = T columns in tiles of 64 with a last tile of 16 / 32 / 48 / 64; K = kf * 104 padded to 16
            k4 = K["gemm<CdaeL4Op>"]
            loops = {lp["mfma_cycles"] // 32: lp for lp in k4["loops"]}           # columns of the class -> its loop (512 cycles = 16 columns, ...)
            ks4 = math.ceil(kf * 104 / 16)
            for n0 in range(0, T, 64):
                w = min(64, 16 * math.ceil((T - n0) / 16))
                lp = loops[w]
                tiles = 4 * math.ceil(B * F * 2 * S / 128)
                # out of the loop: the binary holds four width classes x two epilogues (masks only | estimates as well); what the
                # separator's path runs is ~200 instructions of prologue and the masks-only epilogue, 6 per element (sigmoid 5 +
                # store), w / 2 elements per lane -- an estimate, the static count (8,476 over all bodies) cannot tell them apart
                add("cdae_l4_gemm", tiles * 4, 16.0 * w + lp["mfma_cycles"] * (ks4 - 1) / 2.0, 200.0 + 6.0 * w / 2 + lp["valu"] * ks4 / 2.0)
        # band kernels: rows = channel-slices; radix-4 bands in 32-row tiles of 4 waves; short bands on the dense engine
        # (N = 2 Lg columns, K = 2 Lg).  Default (band_dft4s.h): the four m-point DFTs contracted over input PAIRS -- K-steps of
        # 8 pairs over K2 = m / 2 + 1, 16 MFMAs (512 cycles) per 16-column block of the K2 outputs and K-step; the budget's loop
        # and its epilogue blocks are the three-block form's, scaled to the band's blocks.  XSQ_D4_SYM=0 (band_dft4.h): K-steps of
        # 8 complex over m, 256 MFMA cycles per 16-column block of the 2 m real outputs and K-step.
        sym = not (os.environ.get("XSQ_D4_SYM", "1") == "0") and "band_dft4s<forward>" in K
        for name, key, rows, eblk in (("band_analysis_dft4", "band_dft4<forward>", 2 * B * S, None), ("band_synthesis_dft4", "band_dft4<inverse,masked>", 8 * B * S, 33)):
            if sym:
                # (the analysis binary holds two exclusive load paths -- Hermitian reflection for the two edge bands -- and the
                # whitened-magnitude epilogue in two formats: its static count cannot tell what runs.  Its budget is the
                # synthesis kernel's plus the window products, 16 per K-step, and the magnitudes, ~64 per epilogue block.)
                k = K["band_dft4s<inverse,masked>"]
                lp = k["loops"][0]
                fwd = eblk is None
                e = 95 + (64 if fwd else 0)                    # vector instructions of one epilogue block (of three in the binary)
                for lg in Lg[Lg >= split]:
                    m = int(lg) // 4
                    k2 = m // 2 + 1
                    ncb, ksteps = math.ceil(k2 / 16), math.ceil(k2 / 8)
                    tiles = math.ceil(rows / 32)
                    add(name, tiles * 4, 512.0 * ncb * ksteps,
                        k["outside"]["valu"] + (16 if fwd else 0) - (3 - ncb) * 95 + (64 * ncb if fwd else 0) + (lp["valu"] + (16 if fwd else 0)) * (ksteps - 1))
                continue
            k = K[key]
            lp = k["loops"][0]
            e = eblk if eblk is not None else (k["outside"]["valu"] - 300) / 10.0
            for lg in Lg[Lg >= split]:
                m = int(lg) // 4
                ncb, ksteps = math.ceil(2 * m / 16), math.ceil(m / 8)
                tiles = math.ceil(rows / 32)
                add(name, tiles * 4, 256.0 * ncb * ksteps, k["outside"]["valu"] - (10 - ncb) * e + lp["valu"] * (ksteps - 1))
        for name, key, rows in (("band_analysis_gemm", "gemm<BandFwdOp>", 2 * B * S), ("band_synthesis_gemm", "gemm<BandInvOp>", 8 * B * S)):
            for lg in Lg[Lg < split]:
                engine(name, key, rows, math.ceil(2 * int(lg) / 64), math.ceil(2 * int(lg) / 16))
    return {k: ((m + v) / (SIMDS * PEAK_CLOCK_GHZ * 1e9) * 1e3, m / (m + v)) for k, (m, v) in tot.items()}


def cpu_baseline(threads, full=False):
    """The CPU oracle (a port of the reference, pinned to it by tests/golden) timed on this box's host cores.  Default
    (`full`): the bench's own 240 s track -- the same workload as `value`, the oracle's literal chunk loop over 4 full
    chunks + the tail, ~40 s of host time on 16 threads (6.3 x real time, profiles/r07u_bench_cpu_baseline_full.json).
    --cpu-baseline-clip: one 30 s clip through the same configuration (~3 s; reads ~1.7x faster per audio-second: its working
    set is a tenth of a chunk's)."""
    import torch
    from oracle import separator as osep
    from oracle import slicqt as oslicqt
    from xumx_slicq_amd.synth import synth_audio
    from xumx_slicq_amd.weights import seeded_state_dict
    torch.set_num_threads(threads)
    plan = oslicqt.make_plan()
    sd = seeded_state_dict([(F, T) for (_, F, T) in plan.blocks])
    n = TRACK_SAMPLES if full else 30 * 44100
    x = synth_audio(n, seed=20260101)
    osep.separate(plan, sd, x[..., :44100], causal=False, wiener=False)   # warm
    t0 = time.perf_counter()
    osep.separate(plan, sd, x, causal=False, wiener=False)
    dt = time.perf_counter() - t0
    what = ("the bench's own 240 s stereo track (10,584,000 samples, 5 chunks)" if full
            else "30 s stereo clip (1,323,000 samples)")
    return {"value": round(n / FS / dt, 3), "unit": "x real-time", "cores": threads, "kind": "port",
            "sample_matches_workload": bool(full), "seconds": round(dt, 2),
            "sample": what + ", offline conv stack + mix-phase, oracle/ torch-CPU fp32 "
                      "(the port, timed here; for comparison only: the reference ITSELF ran this configuration's full "
                      "240 s track at 1.87 x real-time on 8 cores of the build container, BASELINE.md section 2)"}


def roofline_issue_table(bound, prof_step, steps_in_prof, wiener=False):
    """`roofline_issue`: every kernel's measured time against its vector-issue bound.  MFMA kernels: the STATIC budget of the
    assembly (issue_bound above, `source: "isa"`).  Kernels without matrix work (the LDS-resident slice FFTs, the Wiener-EM
    kernels): the DYNAMIC count -- SQ_INSTS_VALU of the committed PMC pass of this same command x 4 issue cycles over the
    kernel's SIMD cycles (`source: "pmc"`): branchy straight-line code whose executed path a static count cannot pick
    (interior / edge slices, exclusive load paths), where the counter says exactly what was issued.  `coverage` (last
    element) = the share of the step's kernel time that carries a bound."""
    out = []
    total = sum(ms for ms, _ in prof_step.values())
    covered = 0.0
    for k, (ms, _launches) in sorted(prof_step.items(), key=lambda kv: -kv[1][0]):
        if ms <= 0:
            continue
        if bound and k in bound:
            tb, share = bound[k]
            out.append({"kernel": k, "ms_per_step": round(ms / steps_in_prof, 4), "t_issue_bound_ms": round(tb, 4),
                        "frac_of_issue_bound": round(tb / (ms / steps_in_prof), 4), "mfma_share_of_issue_cycles": round(share, 3),
                        "source": "isa"})
            covered += ms
            continue
        p = pmc_issue(k, "wiener" if wiener else None)
        if p and p.get("mfma_busy", 0) < 0.01:
            out.append({"kernel": k, "ms_per_step": round(ms / steps_in_prof, 4),
                        "t_issue_bound_ms": round(p["valu_issue"] * ms / steps_in_prof, 4), "frac_of_issue_bound": p["valu_issue"],
                        "mfma_share_of_issue_cycles": 0.0, "source": "pmc (%s)" % p["source"]})
            covered += ms
    if out and total > 0:
        out.append({"coverage": round(covered / total, 4), "of": "sum of the step's kernel times (kernels.ms_per_step)"})
    return out


def roofline_tables(work, prof_step, steps_in_prof, wiener=False):
    """Per-kernel fractions from one fully instrumented step: HBM-bound kernels against 8 TB/s, matrix kernels
    against the fp32 MFMA peak.  achieved = algorithmic work of the step's launches / their summed duration."""
    hbm, mfma = [], []
    for k, (ms, launches) in sorted(prof_step.items(), key=lambda kv: -kv[1][0]):
        if k not in work or ms <= 0:
            continue
        bound, amount = work[k]
        sec = ms / steps_in_prof * 1e-3
        if bound == "hbm":
            ach = amount / sec / 1e9
            tr = pmc_traffic(k, "wiener" if wiener else None)
            hbm.append({"kernel": k, "ms_per_step": round(ms / steps_in_prof, 4), "achieved": round(ach, 1), "unit": "GB/s",
                        "peak": HBM_PEAK_GBS, "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(amount),
                        "traffic_per_step": tr[0] if tr else None, "pmc": pmc_issue(k, "wiener" if wiener else None)})
        else:
            ach = amount / sec / 1e12
            mfma.append({"kernel": k, "ms_per_step": round(ms / steps_in_prof, 4), "achieved": round(ach, 2), "unit": "TFLOP/s",
                         "peak": FP32_MFMA_PEAK_TFLOPS, "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                         "algorithmic_flops_per_step": int(amount), "pmc": pmc_issue(k, "wiener" if wiener else None)})
    return hbm, mfma


def executed_mfma_flops(plan, B, chunk_lengths, kernel, winograd):
    """Flops the matrix pipe EXECUTES per step under `kernel` where they differ from the algorithmic count of SURVEY.md 8(d):
    layers 2 / 3 as Winograd F(2, 4) along the time taps -- 5 products per output pair, input channel (52 stored) and matrix
    column (48 of the 50 / 51 channels; the rest on the vector ALU) instead of 8 per pair, real channel and column."""
    if kernel in ("band_synthesis_dft4", "band_analysis_dft4") and os.environ.get("XSQ_D4_SYM", "1") != "0":
        # band_dft4s.h: the four m-point DFTs of a band contracted over input pairs -- per row and residue two real
        # (m / 2 + 1)^2 matrices on (Re, Im) instead of one complex m x m product: 4 x 4 (m / 2 + 1)^2 x 2 flops per row
        rows_per_slice = 8 if kernel == "band_synthesis_dft4" else 2
        split = int(os.environ.get("XSQ_D4_MIN_LG", 24))
        total = 0
        for n in chunk_lengths:
            S = plan.num_slices(max(n, plan.L // 2 + 1))
            for lg in plan.Lg:
                if lg >= split:
                    total += rows_per_slice * B * S * 4 * 4 * (int(lg) // 8 + 1) ** 2 * 2
        return total
    if kernel not in ("cdae_l2_slab", "cdae_l3_slab") or not winograd:
        return None
    from xumx_slicq_amd.weights import freq_filter
    total = 0
    for n in chunk_lengths:
        S = plan.num_slices(max(n, plan.L // 2 + 1))
        T1, T2 = 2 * S - 1, 2 * S - 4
        To = T2 if kernel == "cdae_l2_slab" else T1
        if (To + 1) // 2 < 64:
            continue
        for (_, F, T) in plan.blocks:
            kf = freq_filter(F)
            Fo = F - 2 * kf + 2 if kernel == "cdae_l2_slab" else F - kf + 1
            total += 4 * B * Fo * ((To + 1) // 2) * kf * 5 * 52 * 48 * 2
    return total


def dominant_roofline(dom, prof, work, steps, dt, precision="fp32", wiener=False):
    ms, launches = prof[dom]
    bound, amount = work[dom]
    per_launch = amount * steps / launches          # algorithmic work per launch
    avg_s = ms / launches * 1e-3
    if bound == "hbm":
        ach, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        ach, peak, unit = per_launch / avg_s / 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s"
        if precision != "fp32" and dom.startswith("cdae_"):      # useful flops at 3 / 6 bf16 MFMAs per product
            peak = round(BF16_MFMA_PEAK_TFLOPS / (3.0 if precision == "bf16x3" else 6.0), 1)
    tr = pmc_traffic(dom, "wiener" if wiener else None)
    out = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
           "frac": round(ach / peak, 4), "traffic": int(tr[0] / tr[1]) if tr else None,
           "avg_launch_ms": round(ms / launches, 4), "launches": launches,
           "share_of_step": round(ms / (dt * 1e3), 4)}
    rp = rocprof_launch_ms(dom, "wiener" if wiener else None)
    if rp:           # the same fraction from the committed rocprofv3 average (one launch of this kernel per step: same work)
        out["rocprof"] = {"avg_launch_ms": round(rp[0], 4), "frac": round(ach * avg_s * 1e3 / rp[0] / peak, 4), "source": rp[1]}
    return out


def self_launch(args):
    """`python3 bench.py --gpus N` without a launcher: start the ranks as a fresh child process BEFORE this one
    has made any GPU call (no exec of a process that has initialised the GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def timed_steps(step, steps, world, dist, dev, split=None):
    """K steps between barrier + synchronize, max over ranks.  `split` (a dict) receives what tells a GPU-bound step
    from a host-bound one: two HIP events per step on the launch stream (before its first launch, after its last one)
    -> `gpu_span_ms` (mean / max over the steps: first launch to last completion, gaps included), and the wall time
    the host needed to ENQUEUE the K steps (`host_enqueue_ms` per step, read before the final synchronize).  GPU-bound:
    host_enqueue << ms_per_step ~ gpu_span; host-bound: host_enqueue ~ ms_per_step > the sum of the kernels."""
    import torch
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if split is not None else None
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        if ev:
            ev[i][0].record()
        out = step()
        if ev:
            ev[i][1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ev:
        spans = [a.elapsed_time(b) for a, b in ev]
        split.update({"gpu_span_ms": {"mean": round(sum(spans) / len(spans), 4), "max": round(max(spans), 4), "min": round(min(spans), 4)},
                      "host_enqueue_ms": round(t_enq / steps * 1e3, 4),
                      "gpu_total_ms": round(ev[0][0].elapsed_time(ev[-1][1]) / steps, 4)})
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["track240", "testset50"],
                    help="default: track240 (configs[1]) at N = 1, testset50 (configs[3]) at N > 1")
    ap.add_argument("--tracks", type=int, default=TESTSET_TRACKS, help="testset50: number of tracks")
    ap.add_argument("--stack", type=int, default=4, help="testset50: work items per round / stacked pass")
    ap.add_argument("--wiener", action="store_true", help="BASELINE configs[2]: Wiener-EM on (default off = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--cpu-baseline-clip", action="store_true",
                    help="time the CPU oracle on a 30 s clip (~3 s of host time) instead of the whole 240 s track (the workload of "
                         "`value`: ~40 s of host time on 16 threads)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x6", "bf16x3"],
                    help="arithmetic of the convolution contractions for the headline value (default: exact fp32)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra measurements outside the timed region")
    ap.add_argument("--graph", action="store_true",
                    help="track240: replay the step from a captured HIP graph (Separator.forward_graphed)")
    ap.add_argument("--gather-at-1", action="store_true",
                    help="testset50 on ONE rank with the exchange machinery on (process group of one rank on nccl, exchange "
                         "blocks, one-rank all-gather, placement launches): what packing + placement cost without any link")
    ap.add_argument("--exchange", default="sendrecv", choices=["sendrecv", "allgather"],
                    help="testset50: how the stems reach every rank -- sendrecv (default): every rank keeps the same flat per-track "
                         "layout, kernels write their rows in place, one grouped RCCL send/recv per pass moves rows owner -> peers "
                         "(xsq_exchange_rows); allgather: in-place all_gather_into_tensor per pass + one placement launch per exchange")
    ap.add_argument("--no-verify", action="store_true",
                    help="testset50: skip the bitwise check of sampled tracks against Separator.forward (outside the timed region)")
    ap.add_argument("--no-gather", action="store_true",
                    help="testset50: headline without the all-gather (default: with it; the other one is a variant)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    # stdout carries ONE line, the JSON, and it has to be the last thing on it: libraries write there as well (RCCL
    # prints a version banner at teardown, gloo its connection notes).  File descriptor 1 is pointed at stderr for the
    # whole run and the line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    workload = args.workload or ("track240" if world == 1 else "testset50")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP library is the product path and there is no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("XSQ_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1 and backend == "nccl" and local_world > ndev:
        # RCCL refuses two ranks on one device ("duplicate GPU") deep inside communicator setup: say it here instead
        sys.exit("bench.py: %d ranks on this node but only %d visible device(s): one process per GPU is the contract "
                 "(RCCL cannot put two ranks on one device).  Use --gpus <= %d, or XSQ_DIST_BACKEND=gloo for the "
                 "functional several-ranks-per-device path of the tests." % (local_world, ndev, ndev))
    local_dev = local_rank % ndev                           # (several ranks per GPU only with the gloo backend)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    import torch.distributed as dist
    if world == 1 and args.gather_at_1:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import contextlib
    from xumx_slicq_amd.separator import seeded_separator
    with contextlib.redirect_stdout(sys.stderr):     # keep stdout to the one JSON line
        sep = seeded_separator(realtime=False, wiener=args.wiener, device=dev, chunk_size=CHUNK)
    sep.xumx_model.set_precision(args.precision)

    if workload == "track240":
        result = bench_track(args, sep, dev, world, rank, dist)
    else:
        result = bench_testset(args, sep, dev, world, rank, dist)
    if world > 1:
        dist.barrier()
    from xumx_slicq_amd.sharding import close_row_exchanges
    close_row_exchanges()                 # the library's own RCCL communicator(s): before the process group goes
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        result.update(provenance())
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    os.close(json_fd)


# XSQ_* variables that do not touch the arithmetic or the kernel selection of the measured path
NEUTRAL_ENV = ("XSQ_DIST_BACKEND", "XSQ_RCCL_LIB")


def provenance():
    """`env` and `library` of the JSON line: every XSQ_* variable set in this process (two dozen of them select kernels,
    fusions or diagnostic paths: csrc getenv, separator.py / model.py / transforms.py) and what library is loaded (path, ABI,
    its own build string).  `nondefault` is True when any variable other than the transport switches is set or the library is
    not the in-tree default build: such a line is an A/B arm, not the product's number."""
    from xumx_slicq_amd import _lib
    env = _lib.xsq_environment()
    lib = _lib.build_info()
    odd = sorted(k for k in env if k not in NEUTRAL_ENV)
    plain_build = lib["build"].split("flags=")[-1].split(";")[0].strip() == "no-packed-fp32-ops"
    return {"env": {"xsq": env, "selecting_kernels_or_paths": odd},
            "library": lib,
            "nondefault": bool(odd) or not lib["default_path"] or not plain_build}


DTYPES = {"fp32": "f32", "bf16x6": "f32 (conv contractions: exact 3-way bf16 cut, 6 bf16 MFMAs per product, fp32 accumulate)",
          "bf16x3": "f32 (conv contractions as 3 x bf16 MFMA, fp32 accumulate)"}


def instrumented_warmup(step, warmup, serial=None):
    """Warm-up steps; every kernel is timed with HIP events on its launch stream in the LAST one (the first
    ones build tile tables etc.).  `serial(flag)`: called with True around that step so that the caller can
    take the short tail pass off its side stream -- a kernel timed while another stream's kernels share the
    chip reads long, and the per-kernel roofline fractions are meant per kernel.  Returns {kernel: (ms,
    launches)} of that step."""
    import torch
    from xumx_slicq_amd import _lib
    _lib.profile_filter(None)
    _lib.profile_enable(True)
    warmup = max(warmup, 1)       # the roofline needs one instrumented (untimed) step even with --warmup 0
    for i in range(warmup):
        last = i == warmup - 1
        if last:
            torch.cuda.synchronize()
            _lib.profile_reset()
            if serial:
                serial(True)
        step()
        if last and serial:
            torch.cuda.synchronize()
            serial(False)
    torch.cuda.synchronize()
    return _lib.profile_read()


def bench_track(args, sep, dev, world, rank, dist):
    """configs[1] / [2]: one 240 s track per rank (weak scaling, no collective)."""
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.sharding import chunk_items
    from xumx_slicq_amd.synth import synth_audio
    track = synth_audio(TRACK_SAMPLES, seed=20260101 + rank).to(dev)      # resident in HBM before timing starts
    run = sep.forward_graphed if args.graph else sep

    def step():
        return run(track)

    # Warm-up steps: every kernel is timed with HIP events on its launch stream (the per-kernel table and the
    # choice of the dominant kernel).  Timed region: only the dominant kernel keeps its two events per launch --
    # event records around all launches of a step cost ~0.1 ms of it (tools/prof_overhead.py).
    def serial(on):                       # the instrumented step runs the tail pass after the stacked pass, not beside it
        sep.overlap_tail = not on

    prof_all = instrumented_warmup(step, args.warmup, None if args.graph else serial)
    dom = max(prof_all, key=lambda k: prof_all[k][0]) if prof_all else None
    _lib.profile_filter(dom)
    _lib.profile_reset()
    split = {}
    dt, out = timed_steps(step, args.steps, world, dist, dev, split)
    prof = _lib.profile_read()          # the dominant kernel only, over the timed region
    _lib.profile_enable(False)
    _lib.profile_filter(None)
    nwarm = 1
    if prof_all is None:                # no warm-up step to take the table from
        prof_all, nwarm = prof, args.steps
        dom = max(prof, key=lambda k: prof[k][0]) if prof else None

    plan = sep.nsgt.nsgt.plan
    my_items = [it.length for it in chunk_items([TRACK_SAMPLES], CHUNK)]
    variants = {}
    if world == 1 and not args.no_variants and rank == 0:
        if not args.graph:
            variants["hip_graph"] = variant_graph(args, sep, track, out)
        if args.precision == "fp32":
            variants.update(variant_precisions(args, sep, step, out))
        if not args.wiener:
            variants["wiener"] = variant_wiener(args, dev, track, plan, my_items)
        if not args.wiener and args.precision == "fp32":
            try:
                variants["winograd_f44"] = variant_winograd_f44(args, dev, track, out)
            except Exception as e:                       # noqa: BLE001 -- an A/B arm must not cost the headline line
                variants["winograd_f44"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        variants["train_step"] = variant_train_step(args, sep, dev)
        variants["train_step_bf16"] = variant_train_step(args, sep, dev, precision="bf16")
        if not args.wiener:
            for name, fn in (("cold_start", lambda: variant_cold_start(args)), ("cli", lambda: variant_cli(args, sep, dev))):
                try:
                    variants[name] = fn()
                except Exception as e:                   # noqa: BLE001 -- a side measurement must not cost the headline line
                    variants[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank != 0:
        return None
    audio_s = world * args.steps * TRACK_SAMPLES / FS
    work = algorithmic_work(plan, 1, my_items, args.wiener)
    roofline = dominant_roofline(dom, prof, work, args.steps, dt, args.precision, args.wiener) if dom else None
    hbm, mfma = roofline_tables(work, prof_all, nwarm, args.wiener)
    wino = bool(getattr(sep.xumx_model, "winograd", 1)) and not (int(os.environ.get("XSQ_CDAE_VARIANT", "0")) & 2048) and args.precision == "fp32"
    wmask = int(getattr(sep.xumx_model, "winograd", 7))
    wino = bool(wmask & 1) and wino
    issue = roofline_issue_table(issue_bound(plan, 1, my_items, winograd=wino, l1f=bool(wmask & 2), l4f=bool(wmask & 4) and os.environ.get("XSQ_WIENER_MASKED", "1") != "0") if args.precision == "fp32" else None,
                                 prof_all, nwarm, args.wiener)
    if roofline and issue:
        for row in issue:
            if row.get("kernel") == roofline["kernel"]:
                roofline["issue_bound"] = {"t_issue_bound_ms": row["t_issue_bound_ms"],
                                           "frac_of_issue_bound": round(row["t_issue_bound_ms"] / (roofline["avg_launch_ms"] * roofline["launches"] / args.steps), 4)}
        executed = executed_mfma_flops(plan, 1, my_items, roofline["kernel"], wino)
        if executed:
            roofline["mfma_flops_executed_per_launch"] = int(executed * args.steps / roofline["launches"])
            roofline["frac_by_executed_flops"] = round(executed * args.steps / roofline["launches"] / (roofline["avg_launch_ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
    kernels = {k: {"ms_per_step": round(v[0] / nwarm, 4), "launches_per_step": v[1] / nwarm}
               for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
    result = {
        "metric": METRIC,
        "value": round(audio_s / dt, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPES[args.precision], "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: offline model (Bark-262 sliCQT), one 240 s stereo track "
                               "(10,584,000 samples, 5 chunks) per GPU, %s, seeded synthetic weights"
                               % (2 if args.wiener else 1, "norbert Wiener-EM niter=1" if args.wiener else "Wiener off (mix-phase)"),
                   "parallelism": "one track per rank, %d rank(s), no data-path collective" % world},
        "gpu_span_ms": split.get("gpu_span_ms"), "host_enqueue_ms": split.get("host_enqueue_ms"),
        "gpu_total_ms": split.get("gpu_total_ms"),
        "timing_note": "ms_per_step: wall clock over the K steps incl. the final synchronize; gpu_span_ms: HIP events around each step on "
                       "the launch stream (first launch -> last completion); gpu_total_ms: first event to last event / K; host_enqueue_ms: "
                       "host wall time to issue one step (one C call, xsq_separator_forward).  host_enqueue << ms_per_step = GPU-bound",
        "roofline": roofline,
        "roofline_hbm": hbm,
        "roofline_mfma": mfma,
        "roofline_issue": issue,
        "roofline_issue_note": "fp32 MFMAs and the other vector instructions of a SIMD share one issue port on gfx950: t_issue_bound = sum over tiles "
                               "and waves of (MFMA cycles + 4 x other vector instructions) / (1024 SIMDs x 2.4 GHz), instruction counts from the assembly "
                               "(profiles/isa_budget.json, tools/isa_budget.py); frac_of_issue_bound = t_issue_bound / measured",
        "kernels_source": "last warm-up step: all kernels instrumented and the tail pass serialised behind the stacked pass (clean per-kernel times; their sum exceeds ms_per_step, whose timed region overlaps the two passes and instruments the roofline kernel only)",
        "kernels": kernels,
    }
    if variants:
        result["variants"] = variants
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1), full=not args.cpu_baseline_clip)
    return result


def _timed_loop(fn, steps):
    """(seconds per step, host enqueue seconds per step, last result) of `steps` back-to-back calls."""
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, t_enq / steps, out


def variant_graph(args, sep, track, out):
    """The same step replayed from a captured HIP graph (Separator.forward_graphed): no host-side launch work,
    bitwise the eager result."""
    import torch
    for _ in range(max(1, args.warmup)):
        g = sep.forward_graphed(track)
    torch.cuda.synchronize()
    same = bool(torch.equal(g, out))
    dt, enq, _ = _timed_loop(lambda: sep.forward_graphed(track), args.steps)
    info = {"what": "the headline step as one HIP graph replay (Separator.forward_graphed): the captured kernels read the caller's tensor through a "
                    "device pointer slot (xsq_separator_forward_indirect) -- no copy of the 85 MB input into a static buffer",
            "static_input_copy": bool(next(iter(sep._graphs.values()))[1] is not None) if getattr(sep, "_graphs", None) else None,
            "value": round(TRACK_SAMPLES / FS / dt, 2), "unit": "x real-time",
            "ms_per_step": round(dt * 1e3, 3), "host_enqueue_ms": round(enq * 1e3, 4), "bitwise_equal_to_eager": same}
    sep.drop_graphs()
    return info


def variant_precisions(args, sep, step, out):
    """The same step with the convolution contractions on the split-bf16 matrix path, and its stems against the
    fp32 stems just produced (outside the timed region, N = 1 only)."""
    import torch
    ref_out = out.clone()
    variants = {}
    what = {"bf16x6": "conv contractions as 6 x bf16 MFMA on fp32 operands cut exactly into three bf16 pieces (dropped terms <= 2^-23 |ab|: fp32-grade), fp32 accumulate; everything else unchanged",
            "bf16x3": "conv contractions as 3 x bf16 MFMA on hi/lo-split fp32 operands (~2^-17 per product), fp32 accumulate; everything else unchanged"}
    for prec in ("bf16x6", "bf16x3"):
        sep.xumx_model.set_precision(prec)
        for _ in range(max(1, args.warmup)):
            step()
        tv, enq, vout = _timed_loop(step, args.steps)
        d = (vout - ref_out).double()
        variants[prec] = {
            "what": what[prec],
            "value": round(TRACK_SAMPLES / FS / tv, 2), "ms_per_step": round(tv * 1e3, 3), "host_enqueue_ms": round(enq * 1e3, 4),
            "stems_vs_fp32": {"rms": float(d.pow(2).mean().sqrt()), "max_abs": float(d.abs().max()),
                              "bar": "1e-4 rms / 1e-3 max-abs (BASELINE.json north_star)"}}
        del vout, d
    sep.xumx_model.set_precision("fp32")
    return variants


def variant_winograd_f44(args, dev, track, out):
    """A/B arm: layers 2 / 3 as Winograd F(4, 4) (csrc/cdae_wino4.h, bit 8 of xsq_model_set_winograd; its weights exist only in a
    model created with XSQ_WINO4=1) against the default F(2, 4): the same track, the stems' distance to the headline stems."""
    import contextlib
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.separator import seeded_separator
    os.environ["XSQ_WINO4"] = "1"
    try:
        with contextlib.redirect_stdout(sys.stderr):
            sep4 = seeded_separator(realtime=False, wiener=False, device=dev, chunk_size=CHUNK)
        sep4.xumx_model.set_winograd(15)
        o4 = sep4(track)                                  # (the model handle is created here, with the variable still set)
    finally:
        os.environ.pop("XSQ_WINO4", None)                 # (the line's `env` block describes the headline run)
    for _ in range(max(1, args.warmup)):
        o4 = sep4(track)
    _lib.profile_enable(True); _lib.profile_reset()
    sep4.overlap_tail = False
    sep4(track)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    sep4.overlap_tail = True
    dt, enq, o4 = _timed_loop(lambda: sep4(track), args.steps)
    d = (o4 - out).double()
    return {"what": "A/B arm, off by default: CDAE layers 2 / 3 as Winograd F(4, 4) along the time taps (7 MFMA products per output quad instead "
                    "of 10; one 512-thread workgroup per CU) instead of F(2, 4); everything else unchanged",
            "value": round(TRACK_SAMPLES / FS / dt, 2), "unit": "x real-time", "ms_per_step": round(dt * 1e3, 3),
            "kernels_ms": {k: round(v[0], 4) for k, v in prof.items() if k in ("cdae_l2_slab", "cdae_l3_slab")},
            "stems_vs_default": {"rms": float(d.pow(2).mean().sqrt()), "max_abs": float(d.abs().max())}}


def variant_wiener(args, dev, track, plan, my_items):
    """BASELINE configs[2]: the same track with the norbert Wiener-EM post-filter (niter = 1), fp32."""
    import contextlib
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.separator import seeded_separator
    with contextlib.redirect_stdout(sys.stderr):
        sepw = seeded_separator(realtime=False, wiener=True, device=dev, chunk_size=CHUNK)

    def step():
        return sepw(track)

    def serial(on):
        sepw.overlap_tail = not on

    prof_all = instrumented_warmup(step, max(2, args.warmup), serial)
    dom = max(prof_all, key=lambda k: prof_all[k][0])
    _lib.profile_filter(dom)
    _lib.profile_reset()
    dt1, enq, _ = _timed_loop(step, args.steps)
    dt = dt1 * args.steps
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    _lib.profile_filter(None)
    work = algorithmic_work(plan, 1, my_items, True)
    hbm, _ = roofline_tables(work, prof_all, 1, True)
    return {"what": "BASELINE configs[2]: offline model + norbert Wiener-EM (niter=1), same 240 s track, fp32",
            "value": round(args.steps * TRACK_SAMPLES / FS / dt, 2), "unit": "x real-time",
            "ms_per_step": round(dt / args.steps * 1e3, 3), "host_enqueue_ms": round(enq * 1e3, 4),
            "roofline": dominant_roofline(dom, prof, work, args.steps, dt, "fp32", True),
            "roofline_hbm": [h for h in hbm if h["kernel"].startswith("wiener")],
            "kernels_ms": {k: round(v[0], 4) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}}


def variant_train_step(args, sep, dev, batch=16, seq_dur=2.0, precision="fp32"):
    """BASELINE configs[4] (SURVEY config 5): one training.loop step -- train-mode forward, ComplexMSE + MaskSum,
    backward incl. the differentiable Wiener-EM, AdamW -- on a batch of 16 two-second chunks, offline model."""
    import contextlib
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.separator import seeded_separator
    from xumx_slicq_amd.synth import synth_audio
    from xumx_slicq_amd.training import Trainer
    with contextlib.redirect_stdout(sys.stderr):
        sept = seeded_separator(realtime=False, device=dev)
    tr = Trainer(sept.xumx_model, (sept.nsgt, sept.insgt, sept.cnorm), device=dev, precision=precision)
    n = int(seq_dur * FS)
    y_t = torch.stack([0.5 * synth_audio(n, seed=700 + j, nb_samples=batch) for j in range(4)]).to(dev)
    x = y_t.sum(0)
    losses = [tr.step(x, y_t)[0] for _ in range(max(2, args.warmup))]
    torch.cuda.synchronize()
    # (1) the reference's loop: the loss is looked at after every step (loss.item(), training.py:110)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step(x, y_t)[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # (2) pipelined: step k + 1 is issued before the loss of step k is looked at (Trainer.step(wait=False))
    t0 = time.perf_counter()
    prev = None
    for _ in range(args.steps):
        cur = tr.step(x, y_t, wait=False)
        if prev is not None:
            losses.append(prev[0])
        prev = cur
    losses.append(prev[0])
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / args.steps
    # (3) per-kernel table: an extra instrumented pass (event records cost ~0.2 ms per step and are not in (1) / (2))
    _lib.profile_filter(None)
    _lib.profile_enable(True)
    _lib.profile_reset()
    for _ in range(2):
        tr.step(x, y_t)
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    kern = {k: round(ms / 2, 4) for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    dom = next(iter(kern))
    # algorithmic flops: the four convolution layers forward (SURVEY.md 8(d) formula at B = 16, S = 11), their data
    # gradients (the mirrored layers: same contraction sizes) and their weight gradients (same again)
    w = algorithmic_work(sep.nsgt.nsgt.plan, batch, [n], False)
    fwd = sum(v for k, (_, v) in w.items() if k.startswith("cdae_"))
    flops = 3 * fwd
    return {"what": "BASELINE configs[4]: training.py step, CDAE fwd+bwd with the X-UMX combined loss (ComplexMSE 14 "
                    "combinations + MaskSum), differentiable Wiener-EM, AdamW; batch = 16 chunks of 2 s (S = 11), offline "
                    "model, %s; incl. the five forward sliCQTs (mix + 4 targets) of the batch"
                    % ("fp32" if precision == "fp32" else "forward / data-gradient / weight-gradient contractions on bf16-rounded operands (one v_mfma_f32_32x32x16_bf16 per "
                       "product, fp32 accumulate: the arithmetic of the reference's bf16 autocast convolutions, training.py:473-476), everything else fp32"),
            "ms_per_step": round(dt * 1e3, 3), "chunks_per_s": round(batch / dt, 1), "steps_per_s": round(1.0 / dt, 2),
            "ms_per_step_pipelined": round(dtp * 1e3, 3),
            "loss_readback": "ms_per_step: looked at after every step, as loss.item() in training.py:110; "
                             "ms_per_step_pipelined: step k + 1 issued before the loss of step k is looked at",
            "algorithmic_flops_per_step": int(flops),
            "flops_note": "forward + data-gradient + weight-gradient contractions of the four convolution layers "
                          "(3 x %.1f GFLOP); BatchNorm, loss, Wiener-EM, sliCQTs and AdamW are not counted" % (fwd / 1e9),
            "achieved_tflops": round(flops / dt / 1e12, 2), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            "frac": round(flops / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "frac_note": "of the fp32 MFMA peak (the fp32 arm runs there; the bf16 arm's contractions run on the bf16 pipe, 16x that peak: "
                         "its step is bound by operand delivery, BatchNorm, loss and transforms)",
            "parity": ("tests/test_training.py::test_hip_training_step_at_config_size_matches_the_oracle (this batch, fp32 and bf16x6)" if precision == "fp32"
                       else "tests/test_training.py::test_hip_training_step_bf16_arm_at_config_size (THIS batch, B = 16 x 88,200: the reference's own autograd under "
                            "bf16 autocast, tests/golden/training_step_bf16_b16.npz) and ::test_hip_training_step_bf16_arm_sits_inside_the_reference_autocast_spread (B = 2)"),
            "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)],
            "dominant_kernel": {"kernel": dom, "ms_per_step": kern[dom], "share_of_step": round(kern[dom] / (dt * 1e3), 4)},
            "kernels_ms": dict(list(kern.items())[:12])}


def variant_cold_start(args):
    """`Separator.load(model_path=<reference-style dir>)` -> first stems of a 10 s clip in a FRESH child process
    (tools/cold_start.py; separator.py:50-93), split into phases.  The metric excludes it by the reference's own convention
    (inference.py:28-31); at ~5 ms per track it is what a user of the CLI waits for."""
    import tempfile
    tool = os.path.join(ROOT, "tools", "cold_start.py")
    with tempfile.TemporaryDirectory(prefix="xsq_model_") as d:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, tool, "--make-dir", d], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        runs = []
        for _ in range(2):                       # the second child finds the files and code objects in the page cache
            r = subprocess.run([sys.executable, tool, "--model-path", d], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return {"error": r.stderr[-400:]}
            runs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    out = dict(runs[-1])
    out["what"] = ("fresh process: import torch -> HIP init -> import package -> Separator.load(model_path) -> first stems of a 10 s clip, "
                   "synchronised; second of two child runs (files in the page cache); first run: %.0f ms" % runs[0]["cold_start_ms"])
    return out


def _write_pcm16(path, audio, rate=44100):
    """(2, N) float tensor -> 16-bit PCM wav (the format of a MUSDB18-HQ track)."""
    import struct
    import numpy as np
    a = (audio.clamp(-1, 1) * 32767.0).round().to("cpu").numpy().astype("<i2")
    data = np.ascontiguousarray(a.T).tobytes()
    fmt = struct.pack("<HHIIHH", 1, 2, rate, rate * 4, 4, 16)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sI4s", b"RIFF", 4 + 8 + len(fmt) + 8 + len(data), b"WAVE"))
        f.write(struct.pack("<4sI", b"fmt ", len(fmt)) + fmt)
        f.write(struct.pack("<4sI", b"data", len(data)) + data)


def variant_cli(args, sep, dev, ntracks=8):
    """`python -m xumx_slicq_amd` over a directory (inference.py:118-146): 8 synthetic 240 s 16-bit stereo wavs on a tmpfs
    -> 4 float32 stem wavs each, through the pipelined loop (xumx_slicq_amd.inference.demix_directory: decode | H2D | demix |
    GPU interleave | D2H | encode, overlapped across tracks, pinned staging).  Next to the rate: each stage alone on one
    track, and the bound they give for a perfectly overlapped pipeline."""
    import shutil
    import tempfile
    import torch
    from xumx_slicq_amd import audio as xaudio
    from xumx_slicq_amd.inference import demix_directory
    from xumx_slicq_amd.synth import synth_audio
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 6e9 else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="xsq_cli_", dir=base)
    try:
        os.makedirs(os.path.join(d, "in"))
        for i in range(ntracks):
            _write_pcm16(os.path.join(d, "in", "track%02d.wav" % i), 0.5 * synth_audio(TRACK_SAMPLES, seed=900 + i)[0])
        wavs = sorted(os.path.join(d, "in", f) for f in os.listdir(os.path.join(d, "in")))
        rates = []
        for rep in range(2):
            out = os.path.join(d, "out%d" % rep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = demix_directory(sep, wavs, out, device=dev, quiet=True)
            torch.cuda.synchronize()
            rates.append(ntracks / (time.perf_counter() - t0))
            if rep == 0:
                shutil.rmtree(out)
        gpu_ms = sum(x[2] for x in done) / len(done)
        # the stages alone, on one track
        hin_flat = torch.empty(2 * TRACK_SAMPLES, dtype=torch.float32).pin_memory()
        xaudio.load_audio_into(wavs[0], lambda numel: hin_flat)
        t0 = time.perf_counter()
        hin, _rate = xaudio.load_audio_into(wavs[0], lambda numel: hin_flat)
        decode_ms = (time.perf_counter() - t0) * 1e3
        hout = torch.empty(4, TRACK_SAMPLES, 2, dtype=torch.float32).pin_memory()
        x = hin.to(dev)
        y = torch.empty(4, TRACK_SAMPLES, 2, device=dev)
        torch.cuda.synchronize()

        def timed(fn, n=3):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        h2d_ms = timed(lambda: x.copy_(hin, non_blocking=True))
        d2h_ms = timed(lambda: hout.copy_(y, non_blocking=True))
        est = sep(x[None])
        inter_ms = timed(lambda: est[:, 0].transpose(1, 2).contiguous())
        wdir = os.path.join(d, "w")
        os.makedirs(wdir)
        t0 = time.perf_counter()
        for k in range(4):
            xaudio.save_wav_float_interleaved(os.path.join(wdir, "t%d.wav" % k), hout[k], 44100)
        write_ms = (time.perf_counter() - t0) * 1e3
        readers, writers = 3, 4
        # the same write as the pipeline issues it: `writers` threads, one track (four wavs) each, at the same time -- a tmpfs does
        # not scale with the threads (page allocation), so the stage's rate is measured, not write_ms / writers
        import threading

        def write_track(i):
            for k in range(4):
                xaudio.save_wav_float_interleaved(os.path.join(wdir, "p%d_%d.wav" % (i, k)), hout[k], 44100)
        ths = [threading.Thread(target=write_track, args=(i,)) for i in range(writers)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        write_par_ms = (time.perf_counter() - t0) * 1e3 / writers
        bound = max(decode_ms / readers, h2d_ms, d2h_ms, gpu_ms + inter_ms, write_par_ms)
        return {"what": "python -m xumx_slicq_amd over %d synthetic 240 s 16-bit stereo wavs in %s -> 4 float32 stem wavs per track; "
                        "pipelined loop (3 reader threads decoding straight into pinned buffers, 4 writer threads -- one stem each --, channel interleave on the GPU; pinned staging pools kept across calls: the second of two passes is reported, the first allocates them)" % (ntracks, base),
                "cli_tracks_per_s": round(rates[-1], 2), "cli_tracks_per_s_first_pass": round(rates[0], 2),
                "x_real_time_end_to_end": round(rates[-1] * TRACK_SAMPLES / FS, 1),
                "separator_ms_per_track": round(gpu_ms, 3),
                "stages_alone_ms_per_track": {"read + decode_pcm16_to_pinned_float (1 thread)": round(decode_ms, 1), "h2d_85MB_pinned": round(h2d_ms, 2),
                                              "d2h_339MB_pinned": round(d2h_ms, 2), "gpu_interleave": round(inter_ms, 3),
                                              "write_4_wavs (1 thread)": round(write_ms, 1),
                                              "write_4_wavs per track with %d threads writing at once" % writers: round(write_par_ms, 1)},
                "pipeline_bound_ms_per_track": round(bound, 2),
                # an n-track run also pays one pass through every stage (fill + drain) around its n - 1 steady intervals
                "run_bound_ms_per_track": round((decode_ms + h2d_ms + gpu_ms + inter_ms + d2h_ms + write_par_ms + (ntracks - 1) * bound) / ntracks, 2),
                "ratio_to_run_bound": round(1e3 / rates[-1] / ((decode_ms + h2d_ms + gpu_ms + inter_ms + d2h_ms + write_par_ms + (ntracks - 1) * bound) / ntracks), 2),
                "bound_note": "max over the stages: decode / 3 reader threads, H2D, D2H, demix + interleave, the write stage at its measured rate with 4 threads writing at once",
                "ratio_to_bound": round(1e3 / rates[-1] / bound, 2)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _release():
    """Hand cached blocks of a finished side measurement back to the device (several ranks may share one GPU in the gloo
    functional mode: eight caching allocators, each keeping a dead 6 GB stem allocation, do not fit)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


def bench_testset(args, sep, dev, world, rank, dist):
    """configs[3]: the 50-track set as one chunk batch over the ranks, stems all-gathered (see module docstring)."""
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.sharding import ShardedDemixer
    from xumx_slicq_amd.synth import synth_audio_device
    lengths = testset_lengths(args.tracks)
    total_s = sum(lengths) / FS
    cache = {}
    requested_exchange = args.exchange

    def get_chunk(it):          # resident in HBM before timing starts; a rank only materialises its own items
        key = (it.track, it.chunk)
        if key not in cache:
            cache[key] = synth_audio_device(it.length, seed=20260101 + 64 * it.track + it.chunk, device=dev)
        return cache[key]

    def get_chunk_fresh(it):    # the same audio, not kept (the verification reads whole tracks once)
        return synth_audio_device(it.length, seed=20260101 + 64 * it.track + it.chunk, device=dev)

    gather = not args.no_gather
    if world == 1 and args.gather_at_1:
        gather = "always"
    # sendrecv needs the library's own RCCL communicator.  Whether it is usable is decided by ALL ranks together (an all-reduce
    # of every rank's local outcome inside RowExchange / ShardedDemixer.settle): either every rank exchanges in place or every
    # rank falls back to all-gather + placement, and the `collective` block says which and why.
    dmx = ShardedDemixer(sep, lengths, get_chunk, dev, gather=gather, stack=args.stack, exchange=args.exchange, fallback=True)
    gather = dmx.gather
    for q in dmx.plan.rounds:
        for p in q[rank]:
            get_chunk(p.item)
    torch.cuda.synchronize()
    exchange_note = dmx.settle()          # first exchange as a warm-up step; its outcome is a collective decision too
    if exchange_note:
        print("bench.py: " + exchange_note, file=sys.stderr)
    args.exchange = dmx.exchange

    prof_all = instrumented_warmup(dmx.run, args.warmup)
    dom = max(prof_all, key=lambda k: prof_all[k][0]) if prof_all else None
    _lib.profile_filter(dom)
    _lib.profile_reset()
    dt, _ = timed_steps(dmx.run, args.steps, world, dist, dev)
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    _lib.profile_filter(None)
    verified = verify_testset(dmx, sep, lengths, get_chunk_fresh, dist, world, rank, dev) if not args.no_verify else None
    _release()

    variants = {}
    other_ms = None
    if not args.no_variants:
        # (a) the same step with the other exchange setting
        if world > 1:
            other = ShardedDemixer(sep, lengths, get_chunk, dev, gather=not gather, stack=args.stack, exchange=args.exchange, fallback=True)
            for _ in range(max(1, args.warmup)):
                other.run()
            dto, _ = timed_steps(other.run, args.steps, world, dist, dev)
            variants["no_gather" if gather else "gather"] = {
                "what": ("the same sharded step WITHOUT the all-gather: every rank keeps the stems of its own items (no data-path collective)"
                         if gather else "the same sharded step WITH the RCCL all-gather of all stems to all ranks + placement"),
                "value": round(args.steps * total_s / dto, 2), "unit": "x real-time", "ms_per_step": round(dto / args.steps * 1e3, 3)}
            other_ms = dto / args.steps * 1e3
            del other
            _release()
        # (a2) the same step through the OTHER exchange (A/B of the two forms of the waveform concat)
        if (world > 1 or dmx.gather) and gather:
            alt_name = "allgather" if args.exchange == "sendrecv" else "sendrecv"
            from xumx_slicq_amd.sharding import ExchangeUnavailable
            try:            # (raised on every rank together or on none: RowExchange decides collectively)
                alt = ShardedDemixer(sep, lengths, get_chunk, dev, gather=("always" if world == 1 else True), stack=args.stack, exchange=alt_name)
                alt.settle()
            except ExchangeUnavailable as e:
                alt = None
                variants["exchange_" + alt_name] = {"what": "exchange = %s not available on every rank: %s" % (alt_name, str(e)[:200])}
            if alt is not None:
                for _ in range(max(1, args.warmup)):
                    alt.run()
                dta, _ = timed_steps(alt.run, args.steps, world, dist, dev)
                variants["exchange_" + alt_name] = {
                    "what": "the same step with exchange = %s (%s)" % (alt_name, EXCHANGE_WHAT[alt_name]),
                    "value": round(args.steps * total_s / dta, 2), "unit": "x real-time", "ms_per_step": round(dta / args.steps * 1e3, 3)}
                del alt
                _release()
        # (b) the whole set on rank 0 alone: the single-GPU rate on the SAME workload
        if world > 1:
            dist.barrier()
        if rank == 0:
            if world > 1:
                solo = ShardedDemixer(sep, lengths, get_chunk, dev, gather=False, stack=args.stack, solo=True)
                for q in solo.plan.rounds:
                    for p in q[0]:
                        get_chunk(p.item)
                solo.run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nsolo = max(1, min(args.steps, 3))
                for _ in range(nsolo):
                    solo.run()
                torch.cuda.synchronize()
                ds = (time.perf_counter() - t0) / nsolo
                rtf1 = total_s / ds
                variants["single_rank_same_workload"] = {
                    "what": "the whole 50-track set on rank 0 alone (no collective), timed after the multi-rank region",
                    "value": round(rtf1, 2), "unit": "x real-time", "ms_per_step": round(ds * 1e3, 3),
                    "efficiency_of_headline": round(args.steps * total_s / dt / (world * rtf1), 4)}
                del solo
                _release()
        if world > 1:
            dist.barrier()
    if world == 1 and dmx.gather and not args.no_variants:      # --gather-at-1: the same set without the exchange machinery
        plain = ShardedDemixer(sep, lengths, get_chunk, dev, gather=False, stack=args.stack)
        for _ in range(max(1, args.warmup)):
            plain.run()
        dto, _ = timed_steps(plain.run, args.steps, world, dist, dev)
        other_ms = dto / args.steps * 1e3
        variants["no_gather"] = {"what": "the same set, kernels writing straight into the per-track tensors (no exchange blocks, no placement)",
                                 "value": round(args.steps * total_s / dto, 2), "unit": "x real-time", "ms_per_step": round(other_ms, 3)}
        del plain
        _release()
    collective = collective_block(dmx, dist, world, rank, dev, dt / args.steps * 1e3, other_ms, gather) if (world > 1 or dmx.gather) else None
    if rank != 0:
        return None
    plan = sep.nsgt.nsgt.plan
    my_items = [it.length for q in dmx.plan.rounds for it in (p.item for p in q[0])]
    work = algorithmic_work(plan, 1, my_items, args.wiener)
    roofline = dominant_roofline(dom, prof, work, args.steps, dt, args.precision, args.wiener) if dom else None
    hbm, mfma = roofline_tables(work, prof_all, 1, args.wiener) if prof_all else ([], [])
    stems_gb = 8 * 4 * sum(lengths) / 1e9
    result = {
        "metric": METRIC,
        "value": round(args.steps * total_s / dt, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": DTYPES[args.precision], "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %d seeded track lengths in [150, 420] s (%.0f s of stereo audio, %d chunk "
                               "work items of <= 2,621,440 samples) as one chunk batch, offline model (Bark-262 sliCQT), %s, "
                               "seeded synthetic weights" % (len(lengths), total_s, sum(len(q) for q in dmx.plan.queues),
                                                              "norbert Wiener-EM niter=1" if args.wiener else "Wiener off (mix-phase)"),
                   "parallelism": "chunk items dealt longest-first to %d rank(s) (imbalance %.4f), %d items per stacked round, %s"
                                  % (world, dmx.plan.imbalance(), args.stack,
                                     ("%.1f GB of stems per step to every rank; %s" % (stems_gb, EXCHANGE_WHAT[dmx.exchange]))
                                     if dmx.gather else "no data-path collective")},
        "roofline": roofline,
        "roofline_hbm": hbm,
        "roofline_mfma": mfma,
    }
    if verified is not None:
        result["verified"] = verified
    if collective:
        collective["exchange_requested"] = requested_exchange
        collective["exchange_note"] = exchange_note or "the requested exchange ran on every rank (collective decision: construction and first exchange succeeded everywhere)"
        result["collective"] = collective
    if variants:
        result["variants"] = variants
    return result


def verify_testset(dmx, sep, lengths, get_chunk, dist, world, rank, dev, k=4):
    """Outside the timed region: K sampled tracks of the step's result against `Separator.forward` of the whole track on
    this rank -- bitwise (the sharded path stacks items of DIFFERENT tracks per pass and writes them through row offsets
    into one flat allocation of > 2^32 floats; per-track forward stacks a track's own chunks).  The sample holds the first
    and the last track, the track whose span of the flat allocation crosses 2^32 elements, and the longest one.  A rank
    that holds only its own items (no gather) compares those spans.  Every rank checks; the line reports the AND."""
    import torch
    from xumx_slicq_amd.sharding import all_ranks_ok, chunk_items
    nt = len(lengths)
    cross = next((t for t in range(nt) if dmx.track_off[t] < (1 << 32) <= dmx.track_off[t + 1]), None)
    pick = []
    for t in (0, nt - 1, cross, max(range(nt), key=lambda i: lengths[i])):
        if t is not None and t not in pick:
            pick.append(t)
    pick = pick[:k]
    whole = dmx.gather or world == 1
    mine = {(p.item.track, p.item.chunk) for rnd in dmx.plan.rounds for p in rnd[dmx.rank]}
    ok, worst = True, 0.0
    for t in pick:
        items = chunk_items([lengths[t]], sep.chunk_size)
        x = torch.cat([get_chunk(type(it)(t, it.chunk, it.start, it.length)) for it in items], dim=-1)
        ref = sep(x)
        got = dmx.out[t]
        for it in items:
            if whole or (t, it.chunk) in mine:
                a, b = got[..., it.start:it.start + it.length], ref[..., it.start:it.start + it.length]
                if not torch.equal(a, b):
                    ok = False
                    worst = max(worst, float((a - b).abs().max()))
        del x, ref
    torch.cuda.synchronize()
    agree = all_ranks_ok(ok, None, dev, world) if world > 1 else ok
    return {"tracks": pick, "bitwise": bool(agree), "against": "Separator.forward of the whole track on the checking rank",
            "scope": "whole tracks on every rank" if whole else "the spans of each rank's own items",
            "track_crossing_2^32_flat_elements": cross, "flat_elements": int(dmx.track_off[-1]),
            "max_abs_diff_this_rank": worst}


EXCHANGE_WHAT = {"sendrecv": "sendrecv-inplace: same flat per-track layout on every rank, kernels write their rows in place, one grouped "
                             "ncclSend / ncclRecv per pass kind and round moves rows owner -> peers at identical offsets (xsq_exchange_rows)",
                 "allgather": "allgather+place: in-place all_gather_into_tensor per pass kind and round, async beside the next pass; "
                              "one xsq_place_rows launch per exchange"}
XGMI_LINK_GBPS = 153.0            # MI355X_MICROARCH.md: 7 point-to-point xGMI links per GPU, ~153 GB/s each way


def collective_block(dmx, dist, world, rank, dev, step_ms, other_ms, gather):
    """What a reader of an N > 1 line needs to judge the exchange without guessing: who ran where, over which library,
    how many bytes a rank takes in per step, how much of the step the exchange left exposed (step with the all-gather
    minus the same step without it, both timed here), and what the bytes cost at the links' peak -- a fully connected
    node gives a rank one link per peer, so an all-gather can use (world - 1) links at once."""
    import torch
    me = {"rank": rank, "device": dev.index, "name": torch.cuda.get_device_name(dev),
          "pci": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None), "host": socket.gethostname()}
    devices = [None] * world
    dist.all_gather_object(devices, me)
    # every rank must end a gathered step holding the SAME bits of every track: an exact integer checksum of this rank's stems
    # (the flat per-track allocation read as int32) against the other ranks' -- what the exchange is for, checked where it ran
    replicas = None
    if dmx.gather:
        dmx.run()
        torch.cuda.synchronize()
        from xumx_slicq_amd.sharding import checksum_int32
        mine = checksum_int32(dmx.flat)
        sums = [None] * world
        dist.all_gather_object(sums, mine)
        replicas = {"stems_checksum_int32_sum": sums[0], "identical_on_all_ranks": all(v == sums[0] for v in sums)}
    if rank != 0:
        return None
    acct = dmx.plan.exchange_bytes() if dmx.gather else {"collectives_per_step": 0, "bytes_in_per_rank_per_step": 0,
                                                          "stem_bytes_in_per_rank_per_step": 0, "largest_collective_bytes_per_rank": 0}
    if dmx.gather and getattr(dmx, "exchange", "allgather") == "sendrecv":
        # rows travel as they are: no padding of ragged blocks, the wire bytes ARE the stems of the other ranks
        acct = dict(acct, bytes_in_per_rank_per_step=acct["stem_bytes_in_per_rank_per_step"],
                    largest_collective_bytes_per_rank=max((int(t[:, 3].sum()) * 4 for t in dmx._xtable.values()), default=0))
    backend = dist.get_backend()
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:
        ver = None
    with_ms, without_ms = (step_ms, other_ms) if gather else (other_ms, step_ms)
    links = max(1, world - 1)
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else " (host-staged functional path, not a measurement of xGMI)"),
            "world": world, "devices": devices, "nccl_version": ver,
            "exchange": "sendrecv-inplace" if getattr(dmx, "exchange", "allgather") == "sendrecv" else "allgather+place",
            "replicas": replicas,
            "op": EXCHANGE_WHAT[getattr(dmx, "exchange", "allgather")],
            **acct,
            "step_ms_with_gather": round(with_ms, 3) if with_ms is not None else None,
            "step_ms_without_gather": round(without_ms, 3) if without_ms is not None else None,
            "exposed_wait_ms": round(with_ms - without_ms, 3) if with_ms is not None and without_ms is not None else None,
            "xgmi_link_GBps": XGMI_LINK_GBPS, "links_per_rank": links,
            "model_ms_at_link_bw": round(acct["bytes_in_per_rank_per_step"] / (links * XGMI_LINK_GBPS * 1e9) * 1e3, 3)}


if __name__ == "__main__":
    main()
