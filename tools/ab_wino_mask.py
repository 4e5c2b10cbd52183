"""Same-box interleaved A/B of the fast-convolution forms (xsq_model_set_winograd bit mask): per-kernel ms of the bench track,
step time, and the stems' distance between the arms.  python3 tools/ab_wino_mask.py 1 3 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("XSQ_WINO4", "1")      # build the F(4, 4) weights with the model (bit 8 of the mask; csrc/cdae_wino4.h)
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
masks = [int(a) for a in sys.argv[1:3]] or [1, 3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sep = seeded_separator(realtime=False, wiener=False, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)
outs = {}
for m in masks:
    sep.xumx_model.set_winograd(m)
    outs[m] = sep(x).clone()
torch.cuda.synchronize()
d = (outs[masks[0]] - outs[masks[1]]).double()
print(f"stems mask {masks[0]} vs {masks[1]}: rms {float(d.pow(2).mean().sqrt()):.3e} max {float(d.abs().max()):.3e}", flush=True)
for r in range(reps):
    for m in masks:
        sep.xumx_model.set_winograd(m)
        for _ in range(2): sep(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): sep(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        _lib.profile_enable(True); _lib.profile_reset()
        sep.overlap_tail = False
        for _ in range(3): sep(x)
        torch.cuda.synchronize()
        prof = _lib.profile_read(); _lib.profile_enable(False)
        sep.overlap_tail = True
        print(f"rep {r} mask {m}: {dt*1e3:.3f} ms/step", {k: round(v[0] / 3, 3) for k, v in prof.items() if "cdae" in k}, flush=True)
