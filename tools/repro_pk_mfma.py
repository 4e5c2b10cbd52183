import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener=False, device=dev)
um, eng = sep.xumx_model, sep.insgt.nsgt.nsgt
cs = 2621440
x = synth_audio(10_584_000, seed=20260101).to(dev)
a = x[..., :4 * cs].reshape(1, 2, 4, cs).permute(2, 0, 1, 3).reshape(4, 2, cs).contiguous()
tail = x[..., 4 * cs:].contiguous()
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
um.set_precision(os.environ.get("PREC", "bf16x3"))
ar, lead, S = eng.forward(a); torch.cuda.synchronize()
ref = ar.clone()
with torch.cuda.stream(side):
    Xt = sep.nsgt(tail); um.masks_arena(Xt)
torch.cuda.synchronize()
bad = 0
for trial in range(20):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(4):
            um.masks_arena(Xt)
    ar, lead, S = eng.forward(a)
    main.wait_stream(side); torch.cuda.synchronize()
    bad += int(not torch.equal(ar, ref))
print(os.environ.get("XSQ_LIB", "product"), os.environ.get("PREC", "bf16x3"), f"forward transforms corrupted: {bad} / 20", flush=True)
