"""Training step (BASELINE config 5 shape family): loss + gradients of
training.loop (training.py:66-108) vs fixtures produced by the reference's own autograd
(tests/golden/training_step.npz: B = 2 clips of 1 s, realtime and offline models)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from xumx_slicq_amd.synth import synth_audio


def _inputs(n):
    y_t = torch.stack([0.5 * synth_audio(n, seed=600 + j, nb_samples=2) for j in range(4)])
    return y_t.sum(0), y_t


def _check(g, tag, mse, msk, grads, rtol):
    assert abs(mse - float(g[f"{tag}_mse"])) < 1e-4 * float(g[f"{tag}_mse"])
    assert abs(msk - float(g[f"{tag}_mask"])) < 1e-4 * float(g[f"{tag}_mask"])
    names = [str(k) for k in g["param_names"]]
    norms = dict(zip(names, g[f"{tag}_grad_norms"]))
    worst = 0.0
    for k in names:
        got = float(grads[k].double().norm())
        assert abs(got - norms[k]) <= rtol * norms[k] + 2e-7, (k, got, norms[k])
    for key in g.files:
        if key.startswith(f"{tag}_grad::"):
            k = key.split("::", 1)[1]
            ref = torch.from_numpy(g[key])
            err = float((grads[k].cpu() - ref).abs().max())
            scale = float(ref.abs().max()) + 1e-12
            worst = max(worst, err / scale)
            # (with batch-statistics BN the loss is invariant to input_scale of single-bin blocks: those
            #  gradients are pure rounding noise around 1e-9, hence the absolute floor)
            assert err <= rtol * scale + 2e-7, (k, err, scale)
    return worst


@pytest.mark.parametrize("tag,causal,wiener", [("realtime", True, False), ("offline", False, True)])
def test_oracle_training_gradients_match_reference(oracle_plan, seeded_sd, tag, causal, wiener):
    from oracle import loss as oloss
    g = load_golden("training_step.npz")
    x, y_t = _inputs(int(g["n"]))
    loss, mse, msk, grads = oloss.training_gradients(oracle_plan, seeded_sd, x, y_t, causal=causal, wiener=wiener)
    _check(g, tag, mse, msk, grads, rtol=2e-3)
