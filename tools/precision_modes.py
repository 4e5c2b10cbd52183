"""The three arithmetic modes of the convolution contractions against the CPU oracle (fp32 torch restatement of the
reference) on a 300,000-sample clip, and their step times on the 240 s bench track."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import separator as osep
from oracle import slicqt as oslicqt
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.weights import seeded_state_dict
dev = torch.device("cuda", 0)
plan = oslicqt.make_plan()
sd = seeded_state_dict([(F, T) for (_, F, T) in plan.blocks])
sep = seeded_separator(realtime=False, wiener=False, device=dev)
x = synth_audio(300000, seed=77)
ref = osep.separate(plan, sd, x, causal=False, wiener=False).double()
outs = {}
for prec in ("fp32", "bf16x6", "bf16x3"):
    sep.xumx_model.set_precision(prec)
    o = sep(x.to(dev)).cpu().double(); outs[prec] = o
    d = o - ref
    print(f"{prec:7s} vs CPU oracle: rms {d.pow(2).mean().sqrt():.3e} max {d.abs().max():.3e}", flush=True)
for prec in ("bf16x6", "bf16x3"):
    d = outs[prec] - outs["fp32"]
    print(f"{prec:7s} vs fp32 path : rms {d.pow(2).mean().sqrt():.3e} max {d.abs().max():.3e}", flush=True)
xb = synth_audio(10_584_000, seed=20260101).to(dev)
for prec in ("fp32", "bf16x6", "bf16x3"):
    sep.xumx_model.set_precision(prec)
    for _ in range(2): sep(xb)
    torch.cuda.synchronize(); _lib.profile_enable(True); _lib.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5): sep(xb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print(f"{prec:7s} {dt*1e3:.3f} ms/step", {k: round(v[0] / 5, 3) for k, v in prof.items() if "cdae" in k}, flush=True)
