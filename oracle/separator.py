"""Oracle: Separator.forward (chunk loop) restated on top of the oracle stages.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
``/root/reference/xumx_slicq_v2/separator.py:133-232``.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import model as omodel
from . import slicqt as oslicqt

CHUNK_SIZE = 2621440          # separator.py:53


def separate(plan: oslicqt.Plan, sd: Dict[str, torch.Tensor], audio: torch.Tensor,
             causal: bool, wiener: bool, chunk_size: int = CHUNK_SIZE) -> torch.Tensor:
    """audio (B, 2, N) fp32 -> (4, B, 2, N) fp32 (targets first).

    separator.py:147-158 chunking; 162-168 zero-pad to sllen/2+1 samples;
    170-174 nsgt -> unmix -> insgt(n_samples); 231 hard concat.
    """
    N = audio.shape[-1]
    outs = []
    with torch.no_grad():
        for start in range(0, N, chunk_size):
            a = audio[..., start:min(start + chunk_size, N)]
            n = a.shape[-1]
            min_samples = plan.L // 2 + 1
            if n < min_samples:
                a = torch.cat([a, torch.zeros(*a.shape[:-1], min_samples - n)], dim=-1)
            X = oslicqt.forward(plan, a)
            Y, _ = omodel.unmix(sd, X, causal=causal, wiener=wiener)
            outs.append(oslicqt.inverse(plan, Y, n))
    return torch.cat(outs, dim=-1)
