"""Third training-step fixture (tests/golden/training_step_bf16.npz): the reference's step AS training.py RUNS IT -- forward
and loss under ``torch.autocast("cpu", dtype=torch.bfloat16)`` (/root/reference/xumx_slicq_v2/training.py:68-70,473-476),
``loss.backward()`` outside the context -- on fixture A's batch (B = 2 clips of 1 s, seeds 600..603), offline model
(Wiener-EM), produced by the REFERENCE's own autograd (imported from /root/reference, never copied).

Stored next to the autocast results: the same step in fp32 (what training_step.npz holds) so that the test can take its
tolerances from the fixture's OWN fp32-vs-bf16 spread: a bf16 arm is right when it sits as close to the autocast reference
as bf16 rounding allows, and that distance is not a number to guess.  Development container only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_train_bf16

``--b16``: the same pair of steps on bench.py's configs[4] batch itself -- B = 16 chunks of 88,200 samples, seeds 700..703 as
tests/test_training.py:_inputs16 -- into tests/golden/training_step_bf16_b16.npz (losses, every tensor's gradient norm in
both arithmetics, the per-tensor fp32-vs-bf16 distance, twelve full tensors): the 2.7 ms bench number's own shape.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.weights import seeded_state_dict

OUT = os.path.join(ROOT, "tests", "golden")
KEEP = ["sliced_umx.0.input_mean", "sliced_umx.0.input_scale", "sliced_umx.0.cdaes.1.0.weight", "sliced_umx.0.cdaes.1.1.weight",
        "sliced_umx.0.cdaes.2.3.weight", "sliced_umx.0.cdaes.2.6.weight", "sliced_umx.0.cdaes.3.9.weight", "sliced_umx.0.cdaes.3.9.bias",
        "sliced_umx.1.cdaes.0.3.weight", "sliced_umx.2.cdaes.1.6.weight", "sliced_umx.33.cdaes.2.0.weight", "sliced_umx.69.cdaes.3.9.weight"]


def main():
    b16 = "--b16" in sys.argv
    torch.set_num_threads(8)
    from xumx_slicq_v2.transforms import NSGTBase, make_filterbanks, ComplexNorm
    from xumx_slicq_v2.model import Unmix
    sys.modules.setdefault("auraloss", types.SimpleNamespace(time=types.SimpleNamespace(SDSDRLoss=lambda: None)))
    from xumx_slicq_v2.loss import ComplexMSELossCriterion, MaskSumLossCriterion

    base = NSGTBase("bark", 262, 32.9, fs=44100.0, device="cpu")
    enc, dec = make_filterbanks(base, 44100.0)
    cnorm = ComplexNorm()
    with torch.no_grad():
        jag, _ = base.predict_input_size(1, 2, 2.0)
    sd = seeded_state_dict([(b.shape[2], b.shape[4]) for b in jag], seed=1234)
    n = 88200 if b16 else 44100
    if b16:
        y_t = torch.stack([0.5 * synth_audio(n, seed=700 + j, nb_samples=16) for j in range(4)])
    else:
        y_t = torch.stack([0.5 * synth_audio(n, seed=600 + j, nb_samples=2) for j in range(4)])
    x = y_t.sum(0)

    def step(autocast: bool):
        m = Unmix(cnorm(jag), realtime=False)
        m.load_state_dict(sd, strict=True)
        m.train()
        dtypes = {}
        hooks = [mod.register_forward_hook(lambda _m, _i, out, name=name: dtypes.__setitem__(name, str(out.dtype)))
                 for name, mod in m.named_modules() if name.startswith("sliced_umx.1.cdaes.0.")]
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
        with ctx:                                   # training.py:68-103: transforms, model and both criteria inside the context
            Xc = enc(x)
            Yest, Ymask = m([c.clone() for c in Xc], return_masks=True)
            with torch.no_grad():
                Ytgt = enc(y_t)
            mse = ComplexMSELossCriterion()(Yest, Ytgt)
            msk = MaskSumLossCriterion()(Ymask)
            loss = mse + msk
        for h in hooks:
            h.remove()
        loss.backward()                             # outside the context (training.py:105-108)
        return m, float(mse), float(msk), dtypes

    d = dict(n=n)
    grads = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        m, mse, msk, dtypes = step(ac)
        print(tag, "mse", mse, "mask", msk)
        print("  layer output dtypes of block 1 / target 0:", dtypes)
        d[f"{tag}_mse"], d[f"{tag}_mask"] = mse, msk
        names, norms = [], []
        for k, p_ in m.named_parameters():
            names.append(k)
            norms.append(float(p_.grad.double().norm()))
        d["param_names"] = np.array(names)
        d[f"{tag}_grad_norms"] = np.array(norms)
        grads[tag] = {k: p_.grad.detach().double().clone() for k, p_ in m.named_parameters()}
        for k in KEEP:
            d[f"{tag}_grad::{k}"] = grads[tag][k].float().numpy()
        if ac:
            d["bf16_layer_dtypes"] = np.array([f"{k}={v}" for k, v in sorted(dtypes.items())])
    # the fixture's own spread: per tensor |g_bf16 - g_fp32| / |g_fp32|
    rel = np.array([float((grads["bf16"][k] - grads["fp32"][k]).norm() / max(float(grads["fp32"][k].norm()), 1e-30)) for k in d["param_names"]])
    d["rel_diff_bf16_vs_fp32"] = rel
    print("relative gradient difference bf16 autocast vs fp32: median %.3e, 90%% %.3e, 99%% %.3e, max %.3e" %
          (np.median(rel), np.quantile(rel, 0.9), np.quantile(rel, 0.99), rel.max()))
    print("loss difference: mse %.3e, mask %.3e (relative)" % (abs(d["bf16_mse"] - d["fp32_mse"]) / d["fp32_mse"],
                                                              abs(d["bf16_mask"] - d["fp32_mask"]) / d["fp32_mask"]))
    path = os.path.join(OUT, "training_step_bf16_b16.npz" if b16 else "training_step_bf16.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
