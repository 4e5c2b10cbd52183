"""Oracle: losses of the reference's training / validation loop (forward).

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU restatement of
``/root/reference/xumx_slicq_v2/loss.py`` (ComplexMSELossCriterion :37-76,
MaskSumLossCriterion :79-96) and of the validation body of ``training.loop``
(training.py:66-103 with train=False and the SDR term off)."""
from __future__ import annotations

from itertools import combinations

import torch

from . import model as omodel
from . import slicqt as oslicqt


def complex_mse(pred, target) -> torch.Tensor:
    """loss.py:37-76: mean over blocks of (1/14) sum over the 4C1+4C2+4C3 target subsets of
    mean((sum pred - sum target)^2)."""
    total = 0.0
    for p, t in zip(pred, target):
        acc = 0.0
        for r in (1, 2, 3):
            for S in combinations(range(4), r):
                acc = acc + torch.mean((sum(p[j] for j in S) - sum(t[j] for j in S)) ** 2)
        total = total + acc / 14.0
    return total / len(pred)


def mask_sum(masks) -> torch.Tensor:
    """loss.py:79-96: the four masks of every TF point should sum to one."""
    return sum(torch.mean((m.sum(dim=0) - 1.0) ** 2) for m in masks) / len(masks)


def validation_step(plan, sd, x, y_targets):
    """training.py:66-103 (train=False): offline model, Wiener-EM on.  Returns (loss, mse, mask)."""
    with torch.no_grad():
        X = oslicqt.forward(plan, x)
        Y, masks = omodel.unmix(sd, X, causal=False, wiener=True)
        Yt = oslicqt.forward(plan, y_targets)
        mse, msk = float(complex_mse(Y, Yt)), float(mask_sum(masks))
    return mse + msk, mse, msk


TRAINABLE_SUFFIXES = ("input_mean", "input_scale", ".weight", ".bias")


def training_gradients(plan, sd, x, y_targets, causal: bool, wiener: bool, minima=None):
    """One forward + backward of training.loop (training.py:66-108, train=True, SDR term off, fp32):
    BatchNorm on batch statistics, loss = ComplexMSE + MaskSum, torch autograd.  Returns
    (loss, mse, mask, {key: grad}) for every trainable tensor of the state_dict.  ``minima`` (optional dict) receives,
    per BatchNorm key, the smallest |BatchNorm output| of the batch: how close the step came to a ReLU kink."""
    params = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and k.endswith(TRAINABLE_SUFFIXES)) else v)
              for k, v in sd.items()}
    with torch.no_grad():
        X = oslicqt.forward(plan, x)
        Yt = oslicqt.forward(plan, y_targets)
    Y, masks = omodel.unmix(params, X, causal=causal, wiener=wiener, training=True, minima=minima)
    mse, msk = complex_mse(Y, Yt), mask_sum(masks)
    loss = mse + msk
    loss.backward()
    grads = {k: v.grad for k, v in params.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    return float(loss.detach()), float(mse.detach()), float(msk.detach()), grads
