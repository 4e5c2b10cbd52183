#!/usr/bin/env python3
"""Small-input latency of the hot path (streaming use; BASELINE configs[0] on the
GPU): eager launches vs HIP-graph replay.  Prints one JSON line per case."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xumx_slicq_amd.separator import seeded_separator  # noqa: E402
from xumx_slicq_amd.synth import synth_audio  # noqa: E402

CASES = [
    ("bark262 realtime model, 10 s clip (BASELINE configs[0] shape)", dict(realtime=True), 441000),
    ("bark262 realtime model, 32768-sample chunk (demixui.py:49-51)", dict(realtime=True), 32768),
    ("bark262 offline + Wiener-EM, 32768-sample chunk", dict(realtime=False), 32768),
    ("mel32 realtime model, 32768-sample chunk (rocFFT backend)", dict(realtime=True, fscale="mel", fbins=32, fmin=115.5), 32768),
]


def med(f, n=40):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for name, cfg, n in CASES:
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        sep = seeded_separator(**cfg)
    x = synth_audio(n, seed=1).cuda()
    for _ in range(3):
        sep(x)
    eager = med(lambda: sep(x))
    sep.forward_graphed(x)
    graph = med(lambda: sep.forward_graphed(x))
    print(json.dumps({"case": name, "samples": n, "audio_ms": round(n / 44.1, 1), "eager_ms": round(eager, 3),
                      "graph_ms": round(graph, 3), "rtf_graph": round(n / 44.1 / graph, 1)}), flush=True)
