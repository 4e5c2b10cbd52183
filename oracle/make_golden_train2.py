"""Second training-step fixture (tests/golden/training_step_b.npz), produced by the REFERENCE's own autograd
like section (7b) of oracle/make_golden.py, on another seeded batch.

Why a second one: with ~3e6 BatchNorm outputs per step, a handful of pre-activations always lie within fp32
rounding of a ReLU kink (in fixture A the HIP path lands on the other side for block 32 / target 1, realtime, and
block 36 / target 0, offline), and the tensors upstream of such a ReLU can only be held loosely.  This script takes
the first input seed whose near-kink groups (min |BatchNorm output| < RISK) lie in OTHER blocks than fixture A's,
records the per-group minima (the test derives its loose set from them), and stores the same quantities as (7b):
both loss terms, the gradient norm of every trainable tensor, full gradients of a few tensors -- including the ones
fixture A has to treat loosely.  Every trainable tensor is then held at 2e-3 by at least one of the two fixtures.
Development container only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_train2
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
RISK = 3e-6                      # |pre-activation| below this may fall on the other side of the kink in another summation order
KINKS_A = {"realtime": "sliced_umx.32.cdaes.1", "offline": "sliced_umx.36.cdaes.0"}


def main():
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
    n = 44100

    def run(seed0, rt, backward):
        y_t = torch.stack([0.5 * synth_audio(n, seed=seed0 + j, nb_samples=2) for j in range(4)])
        x = y_t.sum(0)
        m = Unmix(cnorm(jag), realtime=rt)
        m.load_state_dict(sd, strict=True)
        m.train()
        minima = {}
        hooks = []
        for name, mod in m.named_modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                hooks.append(mod.register_forward_hook(
                    lambda _m, _i, out, name=name: minima.__setitem__(name, float(out.detach().abs().min()))))
        Xc = enc(x)
        Yest, Ymask = m([c.clone() for c in Xc], return_masks=True)
        for h in hooks:
            h.remove()
        if not backward:
            return minima, None
        with torch.no_grad():
            Ytgt = enc(y_t)
        mse = ComplexMSELossCriterion()(Yest, Ytgt)
        msk = MaskSumLossCriterion()(Ymask)
        (mse + msk).backward()
        return minima, (m, float(mse), float(msk))

    def risk_groups(minima):
        return sorted({k.rsplit(".", 1)[0] for k, v in minima.items() if v < RISK})

    chosen = None
    for seed0 in range(610, 1000, 10):
        ok, info = True, {}
        for tag, rt in (("realtime", True), ("offline", False)):
            minima, _ = run(seed0, rt, backward=False)
            rg = risk_groups(minima)
            info[tag] = rg
            blocks_a = KINKS_A[tag].split(".cdaes.")[0]
            if any(g.startswith(blocks_a + ".") for g in rg):
                ok = False
        print("seed", seed0, info, "OK" if ok else "", flush=True)
        if ok:
            chosen = seed0
            break
    assert chosen is not None
    d = dict(n=n, seed0=chosen, risk=RISK)
    keep_keys = ["sliced_umx.0.input_mean", "sliced_umx.0.input_scale", "sliced_umx.0.cdaes.1.0.weight",
                 "sliced_umx.1.cdaes.0.3.weight", "sliced_umx.69.cdaes.3.9.weight",
                 # the tensors fixture A can only bound loosely (upstream of its kinks)
                 "sliced_umx.32.input_mean", "sliced_umx.32.input_scale", "sliced_umx.32.cdaes.1.0.weight",
                 "sliced_umx.32.cdaes.1.1.weight", "sliced_umx.32.cdaes.1.3.weight", "sliced_umx.32.cdaes.1.4.bias",
                 "sliced_umx.36.input_mean", "sliced_umx.36.input_scale", "sliced_umx.36.cdaes.0.0.weight",
                 "sliced_umx.36.cdaes.0.1.weight", "sliced_umx.36.cdaes.0.3.weight", "sliced_umx.36.cdaes.0.4.bias",
                 "sliced_umx.36.cdaes.0.6.weight", "sliced_umx.36.cdaes.0.7.weight"]
    for tag, rt in (("realtime", True), ("offline", False)):
        minima, (m, mse, msk) = run(chosen, rt, backward=True)
        d[f"{tag}_mse"], d[f"{tag}_mask"] = mse, msk
        names, norms = [], []
        for k, p_ in m.named_parameters():
            names.append(k)
            norms.append(float(p_.grad.double().norm()))
        d[f"{tag}_grad_norms"] = np.array(norms)
        d["param_names"] = np.array(names)
        for k in keep_keys:
            d[f"{tag}_grad::{k}"] = dict(m.named_parameters())[k].grad.numpy()
        d[f"{tag}_bn_names"] = np.array(sorted(minima))
        d[f"{tag}_bn_min_abs"] = np.array([minima[k] for k in sorted(minima)])
        d[f"{tag}_risk_groups"] = np.array(risk_groups(minima) or [""])
        print("training step B", tag, "mse", mse, "mask", msk, "risk groups", risk_groups(minima))
    np.savez_compressed(os.path.join(OUT, "training_step_b.npz"), **d)
    print("training_step_b.npz", os.path.getsize(os.path.join(OUT, "training_step_b.npz")))


if __name__ == "__main__":
    main()
