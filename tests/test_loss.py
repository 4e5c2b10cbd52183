"""Validation half of a training.loop step (forward only): losses vs the
reference-generated scalars (tests/golden/validation_step.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from xumx_slicq_amd.synth import synth_audio


def _inputs(n):
    y_t = torch.stack([0.5 * synth_audio(n, seed=500 + j, nb_samples=2) for j in range(4)])
    return y_t.sum(0), y_t


def test_oracle_losses_match_reference(oracle_plan, seeded_sd):
    from oracle import loss as oloss
    g = load_golden("validation_step.npz")
    x, y_t = _inputs(int(g["n"]))
    loss, mse, msk = oloss.validation_step(oracle_plan, seeded_sd, x, y_t)
    assert abs(mse - float(g["mse"])) < 1e-5 * float(g["mse"]) + 1e-7
    assert abs(msk - float(g["mask"])) < 1e-5 * float(g["mask"]) + 1e-7
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    # closed form used by the kernel: the 14 squared subset sums equal 4*s2 + 3*s1^2
    e = torch.randn(4, 1000, dtype=torch.float64)
    from itertools import combinations
    direct = sum(sum(e[j] for j in S) ** 2 for r in (1, 2, 3) for S in combinations(range(4), r))
    assert torch.allclose(direct, 4 * (e ** 2).sum(0) + 3 * e.sum(0) ** 2)


@pytest.mark.gpu
def test_hip_losses_match_reference_and_oracle(oracle_plan, seeded_sd):
    from oracle import loss as oloss
    from oracle import model as omodel
    from oracle import slicqt as oslicqt
    from xumx_slicq_amd.loss import ComplexMSELossCriterion, MaskSumLossCriterion, validation_step
    from xumx_slicq_amd.separator import seeded_separator
    g = load_golden("validation_step.npz")
    x, y_t = _inputs(int(g["n"]))
    sep = seeded_separator(realtime=False)
    loss, mse, msk = validation_step(sep.xumx_model, (sep.nsgt, sep.insgt, sep.cnorm), x.cuda(), y_t.cuda())
    assert abs(mse - float(g["mse"])) < 1e-4 * float(g["mse"])
    assert abs(msk - float(g["mask"])) < 1e-4 * float(g["mask"])
    assert abs(loss - float(g["loss"])) < 1e-4 * float(g["loss"])
    # the criteria as separate calls on arbitrary (non-arena) block lists, vs the oracle
    rng = np.random.default_rng(0)
    shapes = [(3, 16), (1, 28), (2, 40)]
    pred = [torch.from_numpy(rng.standard_normal((4, 2, 2, F, 5, T, 2)).astype(np.float32)) for F, T in shapes]
    targ = [torch.from_numpy(rng.standard_normal((4, 2, 2, F, 5, T, 2)).astype(np.float32)) for F, T in shapes]
    msks = [torch.from_numpy(rng.uniform(0, 1, (4, 2, 2, F, 5, T)).astype(np.float32)) for F, T in shapes]
    a = float(ComplexMSELossCriterion()([p.cuda() for p in pred], [t.cuda() for t in targ]))
    b = float(MaskSumLossCriterion()([m.cuda() for m in msks]))
    assert abs(a - float(oloss.complex_mse(pred, targ))) < 1e-5 * a
    assert abs(b - float(oloss.mask_sum(msks))) < 1e-5 * b
