"""Training-step timing (BASELINE config 5: batch of 16 two-second chunks, offline model, fp32).
Prints one JSON line: steps/s, chunks/s, per-kernel milliseconds of one step."""
import argparse, json, os, sys, time
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xumx_slicq_amd import _lib
from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.training import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--seq-dur", type=float, default=2.0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--realtime", action="store_true")
ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x6"])
ap.add_argument("--no-profile", action="store_true", help="no per-kernel events: the clean wall time")
ap.add_argument("--pipelined", action="store_true", help="look at a step's loss after the next step has been issued")
a = ap.parse_args()
n = int(a.seq_dur * 44100)
with redirect_stdout(sys.stderr):
    sep = seeded_separator(realtime=a.realtime)
tr = Trainer(sep.xumx_model, (sep.nsgt, sep.insgt, sep.cnorm), precision=a.precision)
y_t = torch.stack([0.5 * synth_audio(n, seed=700 + j, nb_samples=a.batch) for j in range(4)]).cuda()
x = y_t.sum(0)
losses = []
for _ in range(a.warmup):
    losses.append(tr.step(x, y_t)[0])
torch.cuda.synchronize()
_lib.profile_enable(not a.no_profile); _lib.profile_reset()
t0 = time.perf_counter()
prev = None
for _ in range(a.steps):
    if a.pipelined:
        cur = tr.step(x, y_t, wait=False)
        if prev is not None:
            losses.append(prev[0])
        prev = cur
    else:
        losses.append(tr.step(x, y_t)[0])
if prev is not None:
    losses.append(prev[0])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
prof = _lib.profile_read()
_lib.profile_enable(False)
kern = {k: round(ms / a.steps, 3) for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0])}
print(json.dumps({"metric": "training steps/s", "value": 1.0 / dt, "ms_per_step": dt * 1e3, "chunks_per_s": a.batch / dt,
                  "batch": a.batch, "seq_dur": a.seq_dur, "model": "realtime" if a.realtime else "offline",
                  "dtype": "f32" if a.precision == "fp32" else "f32 (GEMM contractions as bf16x6)", "losses": [round(l, 5) for l in losses], "kernels_ms": kern,
                  "hbm_peak_gb": torch.cuda.max_memory_allocated() / 2**30}))
