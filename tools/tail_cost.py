"""What the short tail chunk of the 240 s track costs: the track with its tail beside the stacked pass (default), behind
it (overlap_tail = False), and the four full chunks alone.  Interleaved rounds; ms per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio

dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener="--wiener" in sys.argv, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)
x4 = x[..., :4 * 2_621_440].contiguous()
xt = x[..., 4 * 2_621_440:].contiguous()


def run(inp, overlap, n=20):
    sep.overlap_tail = overlap
    for _ in range(3):
        sep(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        sep(inp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rnd in range(3):
    a, b, c, d = run(x, True), run(x, False), run(x4, True), run(xt, True)
    print(f"round {rnd}: track, tail beside {a:.3f} | track, tail behind {b:.3f} | four full chunks {c:.3f} | tail alone {d:.3f}"
          f" | exposed tail cost {a - c:.3f} (serial {b - c:.3f})", flush=True)
