"""Dataset statistics vs the reference's own training.get_statistics
(tests/golden/statistics.npz was produced by calling that function)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from xumx_slicq_amd.synth import synth_audio


def _tracks(g):
    return [synth_audio(int(n), seed=900 + i)[0] for i, n in enumerate(g["lens"])]


def test_oracle_statistics_match_reference(oracle_plan):
    from oracle import statistics as ostat
    g = load_golden("statistics.npz")
    assert bool(g["via_reference_function"])
    means, stds = ostat.get_statistics(oracle_plan, _tracks(g))
    assert np.allclose(np.concatenate(means), g["means"], rtol=1e-5, atol=1e-6)
    assert np.allclose(np.concatenate(stds), g["stds"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_statistics_match_reference_and_feed_unmix():
    from xumx_slicq_amd.separator import build_models
    from xumx_slicq_amd.statistics import get_statistics
    from xumx_slicq_amd.model import Unmix
    g = load_golden("statistics.npz")
    xumx, encoder, _ = build_models()
    means, stds = get_statistics(encoder, [t.cuda() for t in _tracks(g)])
    assert len(means) == len(stds) == 70 and means[1].shape == (86,)
    assert np.allclose(np.concatenate(means), g["means"], rtol=2e-5, atol=1e-5)
    assert np.allclose(np.concatenate(stds), g["stds"], rtol=2e-5, atol=1e-5)
    # the constructor path the reference uses them for (model.py:192-203): stored as -mean and 1/std
    jag, _ = encoder[0].nsgt.predict_input_size(1, 2, 2.0)
    m = Unmix(encoder[2](jag), input_means=means, input_scales=stds)
    assert torch.allclose(m.sliced_umx[1].input_mean.detach(), torch.from_numpy(-means[1]).float())
    assert torch.allclose(m.sliced_umx[1].input_scale.detach(), torch.from_numpy(1.0 / stds[1]).float())
