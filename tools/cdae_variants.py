import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
sep.xumx_model.set_precision("bf16x3")
x = synth_audio(10_584_000, seed=20260101).to(dev)
for _ in range(2): out = sep(x)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_reset()
t0 = time.perf_counter()
for _ in range(5): out = sep(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
prof = _lib.profile_read()
print("variant", os.environ.get("XSQ_CDAE_VARIANT", "0"), f"{dt*1e3:.3f} ms/step", {k: round(v[0] / 5, 3) for k, v in prof.items() if "cdae" in k}, "checksum", float(out.double().abs().sum()), flush=True)
