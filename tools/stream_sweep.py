"""Step time of the 240 s track for different (max_stack, pass_streams) splits of the stacked chunks."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.synth import synth_audio
dev = torch.device("cuda", 0)
sep = seeded_separator(realtime=False, wiener="--wiener" in sys.argv, device=dev)
x = synth_audio(10_584_000, seed=20260101).to(dev)
ref = None
for (ms, ns, ot) in [(8, 1, True), (8, 1, False), (2, 1, True), (2, 2, True), (1, 2, True), (1, 4, True), (4, 1, True)]:
    sep.max_stack, sep.pass_streams, sep.overlap_tail = ms, ns, ot
    for _ in range(2):
        out = sep(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = sep(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if ref is None:
        ref = out.clone()
    print(f"max_stack={ms} pass_streams={ns} overlap_tail={ot}: {dt*1e3:.3f} ms  bitwise_equal={torch.equal(out, ref)}", flush=True)
