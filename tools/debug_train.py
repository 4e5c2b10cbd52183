"""Debug aid: HIP training gradients vs the oracle's autograd, worst keys first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.separator import seeded_separator
from xumx_slicq_amd.training import Trainer
from xumx_slicq_amd.weights import seeded_state_dict
from oracle import loss as oloss, slicqt as oslicqt

realtime = (sys.argv[1] if len(sys.argv) > 1 else "realtime") == "realtime"
n = 44100
y_t = torch.stack([0.5 * synth_audio(n, seed=600 + j, nb_samples=2) for j in range(4)])
x = y_t.sum(0)
sep = seeded_separator(realtime=realtime)
tr = Trainer(sep.xumx_model, (sep.nsgt, sep.insgt, sep.cnorm))
print("hip", tr.step(x, y_t, apply_update=False))
G = tr.gradients()
plan = oslicqt.make_plan()
sd = seeded_state_dict([(F, T) for (_, F, T) in plan.blocks], seed=1234)
loss, mse, msk, R = oloss.training_gradients(plan, sd, x, y_t, causal=realtime, wiener=not realtime)
print("oracle", loss, mse, msk)
rows = []
for k, r in R.items():
    g = G[k]
    err = float((g - r).abs().max()); sc = float(r.abs().max())
    rows.append((err / (sc + 1e-30), err, sc, k))
rows.sort(reverse=True)
for rel, err, sc, k in rows[:40]:
    print(f"{rel:9.2e} err {err:9.2e} scale {sc:9.2e} {k}")
bykind = {}
for rel, err, sc, k in rows:
    kind = k.split(".", 2)[2] if k.count(".") > 2 else k
    kind = ".".join(kind.split(".")[-2:])
    a = bykind.setdefault(kind, [0.0, 0.0])
    a[0] = max(a[0], rel if sc > 1e-6 else 0.0); a[1] = max(a[1], err)
for kind, (rel, err) in sorted(bykind.items()):
    print(f"{kind:24s} worst rel {rel:9.2e} worst abs {err:9.2e}")
