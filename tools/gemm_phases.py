#!/usr/bin/env python3
"""Phase timeline of the tile engine running layer 4 (grouped_gemm_kernel<CdaeL4Op, 1, 2>; the operator marked STAMPED) from in-kernel s_memrealtime stamps:
   make -C xumx_slicq_amd/csrc OBJDIR=../../build/stamp OUT=../../build/libstamp.so EXTRA=-DXSQ_GEMM_STAMP=1
   XSQ_LIB=$PWD/build/libstamp.so python tools/gemm_phases.py
Per (tile kind, K-steps): tiles, median prologue (start -> first K-step staged), K loop, epilogue, and the K loop's time
per K-step against the MFMA time it holds per wave (kind 0: 64 columns = 16 v_mfma_f32_32x32x2_f32 of 64 cycles; 1: 32
columns = 8; 3: 48 columns = 8 + 8 v_mfma_f32_16x16x4_f32 of 32 cycles; 2: 16 columns = 8 of the latter)."""
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
fn = _lib.lib.xsq_debug_gemm_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf.ctypes.data, ntiles) == 0
ok = buf[:, 3] > 0
b = buf[ok].astype(np.int64)
t0 = b[:, 0].min()
us = (b[:, :4] - t0) / 100.0
ncb, ks = b[:, 4], b[:, 5]
print("layer-4 launch: %d tiles stamped, span %.1f us" % (ok.sum(), us[:, 3].max()))
print(" kind  K-steps  tiles  prologue  K loop  epilogue   total | per K-step  MFMA per K-step (us at 2.1 GHz)")
tot = {}
mf = {0: 16 * 64 if os.environ.get('XSQ_STAMPED_LAYER', '4') == '4' else 8 * 64 + 8 * 32, 1: 8 * 64, 3: 8 * 64 + 8 * 32, 2: 8 * 32}      # layer 1 (XW = 1): 32 + 16 + vector columns
for c, k in sorted(set(zip(ncb.tolist(), ks.tolist()))):
    m = (ncb == c) & (ks == k)
    d = np.diff(us[m], axis=1)
    print("  %3d  %7d  %5d  %8.2f  %6.2f  %8.2f  %6.2f | %10.2f  %6.2f" % (c, k, m.sum(), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]),
          np.median(us[m][:, 3] - us[m][:, 0]), np.median(d[:, 1]) / max(k, 1), mf[int(c)] / 2100.0))
    tot[(int(c), int(k))] = float((us[m][:, 3] - us[m][:, 0]).sum())
s = sum(tot.values())
print("  share of the summed tile time by (kind, K-steps):", {c: round(v / s, 3) for c, v in tot.items()})
for when in np.linspace(0.2, 0.8, 4) * us[:, 3].max():
    run = (us[:, 0] <= when) & (us[:, 3] > when)
    ph = [(run & (us[:, i] <= when) & (us[:, i + 1] > when)).sum() for i in range(3)]
    print("  t = %6.1f us: %4d tiles resident; prologue %d, K loop %d, epilogue %d" % (when, run.sum(), *ph))
# ---- per CU: how many workgroups are resident, and how long does a freed slot stay empty? ----
hw, xcc = b[:, 6], b[:, 7] & 0xF
cu_key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)      # (xcc, se, sh, cu)
print("  distinct CUs seen: %d" % len(set(cu_key.tolist())))
res, gaps = [], []
for key in list(set(cu_key.tolist()))[:64]:
    m = cu_key == key
    st, en = np.sort(us[m][:, 0]), np.sort(us[m][:, 3])
    # resident count sampled at every start
    for t_ in st[len(st) // 4: 3 * len(st) // 4]:
        res.append(((us[m][:, 0] <= t_) & (us[m][:, 3] > t_)).sum())
    # gap: time from an end to the next start on the same CU (a freed slot being refilled)
    for e_ in en[len(en) // 4: 3 * len(en) // 4]:
        nxt = st[st > e_]
        if len(nxt):
            gaps.append(nxt[0] - e_)
print("  resident workgroups per CU at a start (middle half of the launch): median %d, mean %.2f, max %d" % (np.median(res), np.mean(res), np.max(res)))
print("  end of a workgroup -> next start on the same CU: median %.2f us, p90 %.2f us" % (np.median(gaps), np.percentile(gaps, 90)))
