"""Mirrors of /root/reference/xumx_slicq_v2/loss.py on the HIP library (forward only):
``ComplexMSELossCriterion`` (loss.py:37-76) and ``MaskSumLossCriterion`` (loss.py:79-96), plus
``validation_step`` = the body of ``training.loop`` with ``train=False`` (training.py:66-103).
The SDR criterion (auraloss, off by default: ``--sdr-mcoef -1``) is out of scope."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .arena import BlockTable
from .phase import _tables, _workspace


def _per_block_losses(pred: Sequence[Tensor], target: Sequence[Tensor], masks: Optional[Sequence[Tensor]]):
    """(nblocks, 2) float64 on the device: per block (complex MSE over the 14 combinations, mask-sum MSE)."""
    table = BlockTable([(p.shape[-4], p.shape[-2]) for p in pred])
    P, lead, S = table.as_arena(list(pred))
    Tg, lead_t, S_t = table.as_arena(list(target))
    if lead != lead_t or S != S_t or len(lead) != 3 or lead[0] != 4 or lead[2] != 2:
        raise ValueError(f"expected (4, nb_samples, 2, F, S, T, 2) blocks for both arguments; got {lead} / {lead_t}")
    if P.device.type != "cuda":
        raise _lib.XsqError("the loss kernels run on a ROCm device only; there is no CPU fallback")
    B = lead[1]
    M = None
    if masks is not None:
        M = torch.cat([m.to(torch.float32).reshape(-1) for m in masks]) if not _is_run(masks) else \
            torch.as_strided(masks[0], (table.numel(8 * B, S, complex_=False),), (1,))
    F, T = _tables(table)
    with torch.cuda.device(P.device):
        out = torch.empty(len(table), 2, dtype=torch.float64, device=P.device)
        nbytes = _lib.lib.xsq_loss_workspace(len(table), F.ctypes.data, T.ctypes.data, B, S)
        ws = _workspace(P.device, nbytes)
        _lib.check(_lib.lib.xsq_loss_forward(len(table), F.ctypes.data, T.ctypes.data, P.data_ptr(), Tg.data_ptr(),
                                             M.data_ptr() if M is not None else None, B, S, out.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "xsq_loss_forward")
    return out


def _is_run(ts: Sequence[Tensor]) -> bool:
    ptr = ts[0].data_ptr()
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != ptr:
            return False
        ptr += t.numel() * 4
    return True


class ComplexMSELossCriterion:
    """loss.py:37-76."""

    def __call__(self, pred_complex: List[Tensor], target_complex: List[Tensor]) -> Tensor:
        return _per_block_losses(pred_complex, target_complex, None)[:, 0].mean().float()


class MaskSumLossCriterion:
    """loss.py:79-96."""

    def __call__(self, Ymasks: List[Tensor]) -> Tensor:
        # the mask term only needs the masks; reuse the fused kernel with pred == target == any complex arena
        zeros = [torch.zeros(*m.shape, 2, dtype=torch.float32, device=m.device) for m in Ymasks]
        return _per_block_losses(zeros, zeros, Ymasks)[:, 1].mean().float()


@torch.no_grad()
def validation_step(unmix, encoder, x: Tensor, y_targets: Tensor):
    """training.py:66-103 with train=False (unmix.eval(), no grad) and the SDR term off:
    x (B, 2, N) mix, y_targets (4, B, 2, N) -> (loss, mse_loss, mask_loss) as python floats.
    Both criteria come out of one fused streaming pass over the estimate / target / mask arenas."""
    nsgt, insgt, cnorm = encoder
    Xcomplex = nsgt(x)
    Ycomplex_ests, Ymasks = unmix(Xcomplex, return_masks=True)
    Ycomplex_targets = nsgt(y_targets)
    per_block = _per_block_losses(Ycomplex_ests, Ycomplex_targets, Ymasks)
    mse, mask = (float(v) for v in per_block.mean(dim=0).cpu())
    return mse + mask, mse, mask
