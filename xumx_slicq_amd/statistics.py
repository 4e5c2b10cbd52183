"""Dataset statistics for the input whitening of ``Unmix`` -- the counterpart of
``training.get_statistics`` (/root/reference/xumx_slicq_v2/training.py:115-154): per block, mean and
standard deviation per frequency bin of the channel-mean sliCQT magnitude over all frames of all
tracks (sklearn ``StandardScaler.partial_fit`` semantics: population std), with the reference's
floor ``std = max(std, 1e-4 * max(std))`` per block."""
from __future__ import annotations

from typing import Iterable, List, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .phase import _tables, _workspace


def get_statistics(encoder, tracks: Iterable[Tensor]) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """tracks: iterable of (channels, samples) or (1, channels, samples) mixes on a ROCm device.
    Returns (means, stds): one float64 array of F_b entries per block -- what
    ``Unmix(..., input_means=means, input_scales=stds)`` takes (model.py:192-203)."""
    nsgt = encoder[0]
    eng = nsgt.nsgt.nsgt
    table = eng.table
    F, T = _tables(table)
    sumF = int(F.sum())
    n = np.zeros(len(table), dtype=np.float64)
    acc = np.zeros((sumF, 2), dtype=np.float64)
    for x in tracks:
        x = x[None] if x.dim() == 2 else x
        if x.shape[0] != 1:
            raise ValueError("statistics are collected one track at a time (batch of 1)")
        arena, lead, S = eng.forward(x)
        C = lead[-1]
        with torch.cuda.device(arena.device):
            out = torch.empty(sumF, 2, dtype=torch.float64, device=arena.device)
            ws = _workspace(arena.device, 32 * sumF)
            _lib.check(_lib.lib.xsq_magnitude_stats(len(table), F.ctypes.data, T.ctypes.data, arena.data_ptr(), C, S,
                                                    out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "xsq_magnitude_stats")
        acc += out.cpu().numpy()
        n += S * T.astype(np.float64)
    means, stds, o = [], [], 0
    for b, (Fb, _) in enumerate(table.shapes):
        s1, s2 = acc[o:o + Fb, 0], acc[o:o + Fb, 1]
        mean = s1 / n[b]
        std = np.sqrt(np.maximum(s2 / n[b] - mean * mean, 0.0))
        means.append(mean)
        stds.append(np.maximum(std, 1e-4 * np.max(std)))
        o += Fb
    return means, stds
