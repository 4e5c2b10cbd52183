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


def separate_both(plan: oslicqt.Plan, sd: Dict[str, torch.Tensor], audio: torch.Tensor, causal: bool,
                  chunk_size: int = CHUNK_SIZE):
    """``separate`` with BOTH post-filters from one forward transform and one set of CDAE masks per chunk: the masks
    (model.py:213-262) do not depend on the post-filter chosen at model.py:264-268.  Returns (mix-phase stems,
    Wiener-EM stems); each is what ``separate(..., wiener=False / True)`` returns (tests/test_oracle_golden.py holds that)."""
    N = audio.shape[-1]
    outs = ([], [])
    with torch.no_grad():
        for start in range(0, N, chunk_size):
            a = audio[..., start:min(start + chunk_size, N)]
            n = a.shape[-1]
            min_samples = plan.L // 2 + 1
            if n < min_samples:
                a = torch.cat([a, torch.zeros(*a.shape[:-1], min_samples - n)], dim=-1)
            X = oslicqt.forward(plan, a)
            Y, masks = omodel.unmix(sd, X, causal=causal, wiener=False)
            outs[0].append(oslicqt.inverse(plan, Y, n))
            del Y
            Yw, _ = omodel.unmix(sd, X, causal=causal, wiener=True, masks=masks)
            outs[1].append(oslicqt.inverse(plan, Yw, n))
    return torch.cat(outs[0], dim=-1), torch.cat(outs[1], dim=-1)
