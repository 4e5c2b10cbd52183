import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)
def go(prec, overlap):
    sep.xumx_model.set_precision(prec); sep.overlap_tail = overlap
    out = sep(x); torch.cuda.synchronize(); return out.clone()
for prec in ("bf16x3", "fp32"):
    ref = go(prec, False)
    bad = 0
    n = int(os.environ.get("TRIALS", "30"))
    for trial in range(n):
        o = go(prec, True)
        if not torch.equal(o, ref): bad += 1
    print(os.environ.get("XSQ_LIB", "product lib"), prec, f"overlap runs differing from serial: {bad} / {n}", flush=True)
