"""Oracle: dataset statistics for the input whitening.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates ``training.get_statistics``
(/root/reference/xumx_slicq_v2/training.py:115-154): per block, sklearn StandardScaler statistics
(mean, population std) per frequency bin of the channel-mean magnitude over all frames of all tracks,
std floored at 1e-4 of the block's largest std (:151-153)."""
from __future__ import annotations

import numpy as np
import torch

from . import slicqt as oslicqt


def get_statistics(plan, tracks):
    nb = len(plan.blocks)
    n = np.zeros(nb)
    s1 = [np.zeros(F) for (_, F, _) in plan.blocks]
    s2 = [np.zeros(F) for (_, F, _) in plan.blocks]
    for x in tracks:                                       # x: (channels, samples)
        X = oslicqt.complex_norm(oslicqt.forward(plan, x[None]))      # list of (1, C, F, S, T)
        for b, Xb in enumerate(X):
            m = Xb.flatten(-2, -1).mean(1)[0].double().numpy()       # (F, frames): channel mean, :141-147
            n[b] += m.shape[1]
            s1[b] += m.sum(1)
            s2[b] += (m * m).sum(1)
    means = [a / k for a, k in zip(s1, n)]
    stds = [np.sqrt(np.maximum(b / k - mu * mu, 0.0)) for b, k, mu in zip(s2, n, means)]
    return means, [np.maximum(s, 1e-4 * np.max(s)) for s in stds]
