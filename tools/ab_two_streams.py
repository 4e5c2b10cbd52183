#!/usr/bin/env python3
"""A/B: the four full chunks of the 240 s track as ONE stacked pass (the product schedule) against TWO passes of two chunks
issued on two streams (two Separator instances, own workspaces).  Question: do kernels of two independent passes fill each
other's idle issue slots?  Prints ms per 240 s track for both."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio_device

CHUNK = 2621440
dev = torch.device("cuda", 0)
N = 10584000
x = synth_audio_device(N, 20260101, dev)
sepA = seeded_separator(realtime=False, wiener=False, device=dev, chunk_size=CHUNK)
sepB = seeded_separator(realtime=False, wiener=False, device=dev, chunk_size=CHUNK)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
a = x[..., : 2 * CHUNK].contiguous()
b = x[..., 2 * CHUNK:].contiguous()

def one():
    return sepA(x)

def two():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    with torch.cuda.stream(sA): ya = sepA(a)
    with torch.cuda.stream(sB): yb = sepB(b)
    cur.wait_stream(sA); cur.wait_stream(sB)
    return ya, yb

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for rep in range(2):
    print("one stacked pass (B=4) + tail: %.3f ms" % timeit(one))
    print("two passes (B=2 | B=2 + tail) on two streams: %.3f ms" % timeit(two))
y = one(); ya, yb = two()
print("bitwise equal:", bool(torch.equal(y[..., : 2 * CHUNK], ya)), bool(torch.equal(y[..., 2 * CHUNK:], yb)))
