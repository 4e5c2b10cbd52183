"""Drop-in mirror of /root/reference/xumx_slicq_v2/phase.py on the HIP library:
blockwise_wiener, blockwise_phasemix_sep, abs_of_real_complex (+ the list
helpers wiener / phasemix_sep).  ROCm tensors only; inputs are never modified
(the reference's _atan2 writes into its X argument).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .arena import BlockTable

_WS = {}


def _workspace(device, nbytes):
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)          # per stream: see SliCQEngine.workspace
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _WS[key] = None
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _tables(table: BlockTable):
    F = np.asarray([s[0] for s in table.shapes], dtype=np.int32)
    T = np.asarray([s[1] for s in table.shapes], dtype=np.int32)
    return F, T


def _need_gpu(t: Tensor, who: str):
    if t.device.type != "cuda":
        raise _lib.XsqError(f"{who} runs on a ROCm device only (got '{t.device}'); there is no CPU fallback")


def wiener_em_arena(table: BlockTable, X: Tensor, Y: Tensor, B: int, S: int, win_len: int = 5000,
                    batch_group: int = 0):
    """One EM iteration in place on the estimates arena Y (8B channels) given the
    mix arena X (2B channels).  phase.py:43-59 + norbert/__init__.py:153-260.
    ``batch_group``: runs of that many batch items share the window maximum (0 = whole batch)."""
    F, T = _tables(table)
    with torch.cuda.device(X.device):
        nbytes = _lib.lib.xsq_wiener_workspace(len(table), F.ctypes.data, T.ctypes.data, B, S, win_len)
        if nbytes == 0:
            raise _lib.XsqError("xsq_wiener_workspace: bad arguments")
        ws = _workspace(X.device, nbytes)
        _lib.check(_lib.lib.xsq_wiener_em(len(table), F.ctypes.data, T.ctypes.data, X.data_ptr(), Y.data_ptr(),
                                          B, S, win_len, int(batch_group), ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr()),
                   "xsq_wiener_em")


def wiener_em_masked_arena(table: BlockTable, X: Tensor, masks: Tensor, Y: Tensor, B: int, S: int, win_len: int = 5000,
                           batch_group: int = 0):
    """The same iteration fed by the sigmoid masks (real arena, 8B channels): the initial estimate mask * X
    (model.py:262-264) is formed while the two passes load, Y (8B channels, complex) is only written.
    Same bits as ``xsq_cdae_forward(Y)`` + ``wiener_em_arena``; a third less HBM traffic."""
    F, T = _tables(table)
    with torch.cuda.device(X.device):
        nbytes = _lib.lib.xsq_wiener_workspace(len(table), F.ctypes.data, T.ctypes.data, B, S, win_len)
        if nbytes == 0:
            raise _lib.XsqError("xsq_wiener_workspace: bad arguments")
        ws = _workspace(X.device, nbytes)
        _lib.check(_lib.lib.xsq_wiener_em_masked(len(table), F.ctypes.data, T.ctypes.data, X.data_ptr(), masks.data_ptr(),
                                                 Y.data_ptr(), B, S, win_len, int(batch_group), ws.data_ptr(), ws.numel(),
                                                 _lib.stream_ptr()),
                   "xsq_wiener_em_masked")


def _one_block(mix_slicqt: Tensor, slicqtgrams: Tensor):
    if mix_slicqt.dim() != 6 or mix_slicqt.shape[-1] != 2 or mix_slicqt.shape[1] != 2:
        raise ValueError(f"mix must be (nb_samples, 2, F, S, T, 2); got {tuple(mix_slicqt.shape)}")
    B, _, F, S, T, _ = mix_slicqt.shape
    if tuple(slicqtgrams.shape) != (4, B, 2, F, S, T):
        raise ValueError(f"magnitudes must be (4, {B}, 2, {F}, {S}, {T}); got {tuple(slicqtgrams.shape)}")
    _need_gpu(mix_slicqt, "phase")
    return BlockTable([(F, T)]), B, S


def blockwise_phasemix_sep(X_block: Tensor, Ymag_block: Tensor) -> Tensor:
    """phase.py:96-113: Y = Ymag * exp(i angle(X)).  (B,2,F,S,T,2), (4,B,2,F,S,T) -> (4,B,2,F,S,T,2)."""
    table, B, S = _one_block(X_block, Ymag_block)
    F, T = _tables(table)
    X = X_block.contiguous().float()
    mag = Ymag_block.contiguous().float()
    with torch.cuda.device(X.device):
        Y = torch.empty(*Ymag_block.shape, 2, dtype=torch.float32, device=X.device)
        _lib.check(_lib.lib.xsq_phasemix(1, F.ctypes.data, T.ctypes.data, X.data_ptr(), mag.data_ptr(),
                                         Y.data_ptr(), B, S, _lib.stream_ptr()), "xsq_phasemix")
    return Y


def blockwise_wiener(mix_slicqt: Tensor, slicqtgrams: Tensor, wiener_win_len_param: int = 5000) -> Tensor:
    """phase.py:18-69.  (B,2,F,S,T,2), (4,B,2,F,S,T) -> (4,B,2,F,S,T,2)."""
    table, B, S = _one_block(mix_slicqt, slicqtgrams)
    X = mix_slicqt.contiguous().float()
    Y = blockwise_phasemix_sep(X, slicqtgrams)
    nb_frames = S * mix_slicqt.shape[4]
    win = int(wiener_win_len_param) if wiener_win_len_param else nb_frames
    wiener_em_arena(table, X.view(-1), Y.view(-1), B, S, win)
    return Y


def abs_of_real_complex(Xcomplex_real_view: Tensor) -> Tensor:
    """phase.py:116-118."""
    return torch.sqrt(Xcomplex_real_view[..., 0] ** 2 + Xcomplex_real_view[..., 1] ** 2)


def wiener(mix_slicqt: List[Tensor], slicqtgrams: List[Tensor], wiener_win_len: int = 5000):
    """phase.py:7-15."""
    return [blockwise_wiener(m, s, wiener_win_len) for m, s in zip(mix_slicqt, slicqtgrams)]


def phasemix_sep(X: List[Tensor], Ymag: List[Tensor]):
    """phase.py:121-126."""
    return [blockwise_phasemix_sep(x, y) for x, y in zip(X, Ymag)]
