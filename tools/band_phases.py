#!/usr/bin/env python3
"""Phase timeline of the radix-4 synthesis kernel (band_dft4_full_kernel<false>) from in-kernel s_memrealtime stamps:
   make -C xumx_slicq_amd/csrc OBJDIR=../../build/stamp OUT=../../build/libstamp.so EXTRA=-DXSQ_D4_STAMP=1
   XSQ_LIB=$PWD/build/libstamp.so python tools/band_phases.py
Per class of band width (16-column blocks ncb): tiles, K-steps, median prologue (start -> first operands staged),
K loop, epilogue, and the K loop's time per K-step against the MFMA time it holds (8 ncb v_mfma_f32_16x16x4_f32 of 32
cycles per wave and K-step)."""
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
ntiles = 1 << 17
buf = np.zeros((ntiles, 8), dtype=np.uint64)
fn = _lib.lib.xsq_debug_d4_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf.ctypes.data, ntiles) == 0
ok = buf[:, 3] > 0
b = buf[ok].astype(np.int64)
t0 = b[:, 0].min()
us = (b[:, :4] - t0) / 100.0
ncb, ks = b[:, 4], b[:, 5]
print("synthesis launch: %d tiles stamped, span %.1f us" % (ok.sum(), us[:, 3].max()))
print("  ncb  tiles  K-steps  prologue  K loop  epilogue   total | per K-step  MFMA per K-step (us at 2.1 GHz)")
tot = {}
for c in sorted(set(ncb)):
    m = ncb == c
    d = np.diff(us[m], axis=1)
    k = int(np.median(ks[m]))
    print("  %3d  %5d  %7d  %8.2f  %6.2f  %8.2f  %6.2f | %10.2f  %6.2f" % (c, m.sum(), k, np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]),
          np.median(us[m][:, 3] - us[m][:, 0]), np.median(d[:, 1]) / max(k - 1, 1), 8 * c * 32 / 2100.0))
    tot[int(c)] = float((us[m][:, 3] - us[m][:, 0]).sum())
s = sum(tot.values())
dall = np.diff(us, axis=1)
print("  share of the summed tile time by phase: prologue %.3f, K loop %.3f, epilogue %.3f; summed %.0f us over %.1f us = %.0f tiles resident on average"
      % (*(dall.sum(axis=0) / dall.sum()), dall.sum(), us[:, 3].max(), dall.sum() / us[:, 3].max()))
print("  share of the summed tile time by ncb:", {c: round(v / s, 3) for c, v in tot.items()})
for when in np.linspace(0.2, 0.8, 4) * us[:, 3].max():
    run = (us[:, 0] <= when) & (us[:, 3] > when)
    ph = [(run & (us[:, i] <= when) & (us[:, i + 1] > when)).sum() for i in range(3)]
    print("  t = %6.1f us: %4d tiles resident; prologue %d, K loop %d, epilogue %d" % (when, run.sum(), *ph))
