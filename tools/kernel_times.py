"""Per-kernel ms/step of the bench step (fp32), for diagnostic library builds (XSQ_LIB=...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("XSQ_WINO4", "1")      # build the F(4, 4) weights with the model (bit 8 of the mask; csrc/cdae_wino4.h)
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
sep.xumx_model.set_precision(os.environ.get("PREC", "fp32"))
if os.environ.get("WINO_MASK"):     # xsq_model_set_winograd bit mask (default 7; 15 = F(4, 4) for layers 2 / 3)
    sep.xumx_model.set_winograd(int(os.environ["WINO_MASK"]))
if os.environ.get("NO_TAIL"):       # four full chunks, one stacked pass, nothing on the side stream: clean per-kernel times
    sep.overlap_tail = False
    x = synth_audio(4 * 2_621_440, seed=20260101).to(dev)
else:
    x = synth_audio(10_584_000, seed=20260101).to(dev)
for _ in range(2): out = sep(x)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_reset()
t0 = time.perf_counter()
for _ in range(5): out = sep(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
prof = _lib.profile_read()
want = sys.argv[1:] or None
print(os.path.basename(os.environ.get("XSQ_LIB", "product")), f"{dt*1e3:.3f} ms/step", {k: round(v[0] / 5, 3) for k, v in prof.items() if want is None or any(w in k for w in want)}, flush=True)
