"""Step time with and without the per-kernel HIP-event instrumentation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)
for _ in range(3): sep(x)
for rep in range(2):
    for on in (False, True):
        _lib.profile_enable(on); _lib.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): sep(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        _lib.profile_read(); _lib.profile_enable(False)
        print(f"events {'on ' if on else 'off'}: {dt*1e3:.3f} ms/step", flush=True)
