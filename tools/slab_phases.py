#!/usr/bin/env python3
"""Phase timeline of the fp32 slab kernel running layer 3 (cdae_slab_kernel<true, 3, true>) from in-kernel stamps:
   make -C xumx_slicq_amd/csrc OBJDIR=../../build/stamp OUT=../../build/libstamp.so EXTRA=-DXSQ_SLAB_STAMP=1
   XSQ_LIB=$PWD/build/libstamp.so python tools/slab_phases.py
Per frequency-filter height kf (1 / 3 / 5 slabs per tile): tiles, median prologue, slot loop, epilogue, the part of the
slot loop spent in the slab-fetch slots, and the slot loop against the MFMA time it holds per wave (13 chunks of 16 k per
df: 8 v_mfma_f32_32x32x2_f32 of 64 cycles + 8 v_mfma_f32_16x16x4_f32 of 32 cycles each)."""
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
x = synth_audio(int(os.environ.get("XSQ_STAMP_CHUNKS", "4")) * 2_621_440, seed=1).cuda()
for _ in range(3):
    sep(x)
torch.cuda.synchronize()
ntiles = 1 << 16
raw = np.zeros(ntiles * 12, dtype=np.uint64)
buf = raw[:ntiles * 8].reshape(ntiles, 8)
buf2 = raw[ntiles * 8:].reshape(ntiles, 4)
fn = _lib.lib.xsq_debug_slab_stamps
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
occ = fn(raw.ctypes.data, ntiles)
ok = buf[:, 3] > 0
b = buf[ok].astype(np.int64)
us = (b[:, :4] - b[:, 0].min()) / 100.0
kf, fetch = b[:, 4], b[:, 5] / 100.0
print("layer-3 slab launch: %d tiles stamped, span %.1f us, workgroups per CU by the runtime: %d" % (ok.sum(), us[:, 3].max(), occ))
print("  kf  tiles  prologue  slot loop  (slab-fetch slots)  epilogue   total | MFMA time per wave in the slot loop (us at 2.1 GHz)")
for c in sorted(set(kf.tolist())):
    m = kf == c
    d = np.diff(us[m], axis=1)
    print("  %2d  %5d  %8.2f  %9.2f  %18.2f  %8.2f  %6.2f | %6.2f" % (c, m.sum(), np.median(d[:, 0]), np.median(d[:, 1]), np.median(fetch[m]), np.median(d[:, 2]),
          np.median(us[m][:, 3] - us[m][:, 0]), c * 13 * (8 * 64 + 8 * 32) / 2100.0))
hw = b[:, 6]
cu_key = (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)
res, gaps = [], []
for key in list(set(cu_key.tolist()))[:64]:
    m = cu_key == key
    st, en = np.sort(us[m][:, 0]), np.sort(us[m][:, 3])
    for t_ in st[len(st) // 4: 3 * len(st) // 4]:
        res.append(((us[m][:, 0] <= t_) & (us[m][:, 3] > t_)).sum())
    for e_ in en[len(en) // 4: 3 * len(en) // 4]:
        nxt = st[st > e_]
        if len(nxt):
            gaps.append(nxt[0] - e_)
# (the CU key lacks the XCC id in this build -- its stamp slot holds the "lane setup done" time -- so eight CUs share a key)
print("  resident workgroups per group of 8 same-numbered CUs at a start: mean %.2f, max %d" % (np.mean(res), np.max(res)))
tot = (us[:, 3] - us[:, 0]).sum()
print("  of the summed tile time: prologue %.3f, slot loop %.3f (slab-fetch slots %.3f), epilogue %.3f"
      % ((us[:, 1] - us[:, 0]).sum() / tot, (us[:, 2] - us[:, 1]).sum() / tot, fetch.sum() / tot, (us[:, 3] - us[:, 2]).sum() / tot))
p2 = buf2[ok].astype(np.int64)
for c in sorted(set(kf.tolist())):
    m = kf == c
    t00 = b[m][:, 0]
    print("  kf %d prologue detail (us from the start): request 0 of the slab issued %.2f, lane setup done %.2f, all requests issued %.2f, first slab data %.2f, request 3 issued %.2f, barrier passed %.2f"
          % (c, np.median((p2[m][:, 0] - t00) / 100.0), np.median((b[m][:, 7] - t00) / 100.0), *[np.median((p2[m][:, i] - t00) / 100.0) for i in range(1, 4)],
             np.median((b[m][:, 1] - t00) / 100.0)))
