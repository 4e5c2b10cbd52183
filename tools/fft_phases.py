#!/usr/bin/env python3
"""Phase timeline of k_slice_irfft from in-kernel s_memrealtime stamps (diagnostic build only):
   make -C xumx_slicq_amd/csrc OBJDIR=../../build/stamp OUT=../../build/libstamp.so EXTRA=-DXSQ_FFT_STAMP=1
   XSQ_LIB=$PWD/build/libstamp.so python tools/fft_phases.py
Prints, for the LAST inverse launch of one 240 s track (the odd-slice launch of the stacked pass or the tail), the median
duration of every phase per workgroup, how many workgroups ran concurrently, and how the phases of concurrently running
workgroups line up in time (are they in lockstep?)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xumx_slicq_amd import _lib  # noqa: E402
from xumx_slicq_amd.separator import seeded_separator  # noqa: E402
from xumx_slicq_amd.synth import synth_audio  # noqa: E402

sep = seeded_separator(realtime=False, wiener=False)
sep.overlap_tail = False
x = synth_audio(4 * 2_621_440, seed=1).cuda()          # four full chunks: one stacked pass, no tail
for _ in range(3):
    sep(x)
torch.cuda.synchronize()
rows = 4672
buf = np.zeros((rows, 8), dtype=np.uint64)
fn = _lib.lib.xsq_debug_fft_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf.ctypes.data, rows) == 0
t = buf[:, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0                                  # 100 MHz -> microseconds
names = ["gather", "pre-process", "43-point stage", "steps 2 / 3", "output issue"]
d = np.diff(us, axis=1)
print("launch span %.1f us, %d workgroups" % (us[:, 5].max(), rows))
for i, n in enumerate(names):
    print("  %-16s median %6.2f us   p10 %6.2f   p90 %6.2f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  %-16s median %6.2f us" % ("row total", np.median(us[:, 5] - us[:, 0])))
g = (buf[:, 6:8].astype(np.int64) - t0) / 100.0       # stamps 6 / 7: first value of phases 0-1 / 2-3 has arrived
print("  inside the gather: start -> first data of phases 0/1 %.2f us, accumulate 0/1 + request 2/3 -> first data %.2f us, accumulate 2/3 %.2f us"
      % (np.median(g[:, 0] - us[:, 0]), np.median(g[:, 1] - g[:, 0]), np.median(us[:, 1] - g[:, 1])))
# concurrency and phase alignment: at a few instants, what fraction of the running workgroups is in which phase?
for when in np.linspace(0.15, 0.85, 8) * us[:, 5].max():
    run = (us[:, 0] <= when) & (us[:, 5] > when)
    ph = [(run & (us[:, i] <= when) & (us[:, i + 1] > when)).sum() for i in range(5)]
    print("  t = %6.1f us: %4d running;  in phase: %s" % (when, run.sum(), "  ".join("%s %3d" % (n.split()[0], c) for n, c in zip(names, ph))))
