"""fp32 vs the bf16 modes of the convolution contractions on the 240 s bench track: stem error against the fp32
path (itself pinned to the reference at 1.1e-7 RMS) and step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
modes = sys.argv[1:] or ["bf16x6", "bf16x3"]
for wiener in (False, True):
    sep = seeded_separator(realtime=False, wiener=wiener, device=dev)
    x = synth_audio(10_584_000, seed=20260101).to(dev)
    res = {}
    for prec in ["fp32"] + modes:
        sep.xumx_model.set_precision(prec)
        for _ in range(2):
            out = sep(x)
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            out = sep(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        prof = _lib.profile_read(); _lib.profile_enable(False)
        res[prec] = out.clone()
        print(f"wiener={wiener} {prec}: {dt*1e3:.3f} ms/step", {k: round(v[0] / 5, 3) for k, v in prof.items() if "cdae" in k}, flush=True)
    for prec in modes:
        d = (res[prec] - res["fp32"]).double()
        print(f"wiener={wiener}: {prec} vs fp32 stems: rms {d.pow(2).mean().sqrt():.3e} max {d.abs().max():.3e} "
              f"(stem rms {res['fp32'].double().pow(2).mean().sqrt():.3e}, max {res['fp32'].abs().max():.3e})", flush=True)
