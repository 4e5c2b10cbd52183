"""A/B of the four-target layer-1 tiles (csrc/cdae_l1q.h) against the per-target tiles, same process, interleaved rounds."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio

dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)


def run(quad, n=20):
    sep.xumx_model.l1_quad = quad
    for _v, h in sep.xumx_model._handles.values():
        _lib.lib.xsq_model_set_l1_quad(h, int(quad))
    for _ in range(3):
        sep(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        sep(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    _lib.profile_filter("cdae_l1_gemm"); _lib.profile_enable(True); _lib.profile_reset()
    for _ in range(5):
        sep(x)
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_read()["cdae_l1_gemm"]
    _lib.profile_enable(False); _lib.profile_filter(None)
    return dt, ms / 5


for rnd in range(3):
    a, b, c = run(0), run(2), run(4)
    print(f"round {rnd}: per-target tiles step {a[0]:.3f} ms, layer 1 {a[1]:.3f} | two targets per tile {b[0]:.3f}, {b[1]:.3f} | four {c[0]:.3f}, {c[1]:.3f}", flush=True)
