"""Oracle: per-block CDAE, mix-phase and norbert Wiener-EM post-filters.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU fp32 restatement of
``/root/reference/xumx_slicq_v2/model.py`` (Unmix / _SlicedUnmixCDAE /
_CausalConv2d), ``phase.py`` and the ``norbert.wiener`` call the reference makes
(``iterations=1, use_softmask=False``).  Weights come in as a plain
``{state_dict key: tensor}`` mapping with the reference's key layout
(SURVEY.md 8(a) M2), so the same mapping loads into the reference ``Unmix``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # torch BatchNorm2d default, model.py:137,152,165
WIENER_WIN = 5000      # phase.py:18


def freq_filter(nb_f_bins: int) -> int:
    """model.py:112-117."""
    if nb_f_bins < 10:
        return 1
    if nb_f_bins < 20:
        return 3
    return 5


def abs_of_real_complex(X: torch.Tensor) -> torch.Tensor:
    """phase.py:116-118."""
    return torch.sqrt(X[..., 0] ** 2 + X[..., 1] ** 2)


def _bn(x, sd, key, training=False, minima=None):
    if training:      # batch statistics (nn.BatchNorm2d in train mode); running stats are not touched here
        y = F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], training=True, eps=BN_EPS)
        if minima is not None:       # distance of the closest pre-activation to the ReLU kink (tests: subgradient ambiguity)
            minima[key] = float(y.detach().abs().min())
        return y
    return F.batch_norm(
        x, sd[key + ".running_mean"], sd[key + ".running_var"],
        sd[key + ".weight"], sd[key + ".bias"], training=False, eps=BN_EPS)


def cdae_masks(sd: Dict[str, torch.Tensor], b: int, mag: torch.Tensor,
               causal: bool, training: bool = False, minima=None) -> torch.Tensor:
    """Sigmoid masks of the four target CDAEs of block ``b``.

    mag (B, 2, F, S, T) fp32 -> (4, B, 2, F, S, T).  model.py:213-261 (whiten
    236-242, four-layer stack 130-181, crop 251); causal first layer
    model.py:274-290.
    """
    B, C, Fb, S, T = mag.shape
    pre = f"sliced_umx.{b}."
    x = mag.reshape(B, C, Fb, S * T)
    x = (x + sd[pre + "input_mean"][None, None, :, None]) * sd[pre + "input_scale"][None, None, :, None]
    hop = T // 2
    out = []
    for t in range(4):
        p = f"{pre}cdaes.{t}."
        y = F.pad(x, (T - 1, 0)) if causal else x
        y = F.conv2d(y, sd[p + "0.weight"], stride=(1, hop))
        y = F.relu(_bn(y, sd, p + "1", training, minima))
        y = F.conv2d(y, sd[p + "3.weight"])
        y = F.relu(_bn(y, sd, p + "4", training, minima))
        y = F.conv_transpose2d(y, sd[p + "6.weight"])
        y = F.relu(_bn(y, sd, p + "7", training, minima))
        y = F.conv_transpose2d(y, sd[p + "9.weight"], sd[p + "9.bias"], stride=(1, hop))
        y = torch.sigmoid(y)
        y = y[..., :Fb, : S * T]
        out.append(y.reshape(B, C, Fb, S, T))
    return torch.stack(out)


def phasemix_sep(X: torch.Tensor, Ymag: torch.Tensor) -> torch.Tensor:
    """phase.py:96-113 with the hand-rolled _atan2 (72-93) replaced by
    torch.atan2 on a copy (the reference mutates X where re=im=0).
    X (B,2,F,S,T,2), Ymag (4,B,2,F,S,T) -> (4,B,2,F,S,T,2)."""
    ph = torch.atan2(X[..., 1], X[..., 0])
    return torch.stack((Ymag * torch.cos(ph), Ymag * torch.sin(ph)), dim=-1)


def _em_one_iteration(y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """norbert/__init__.py:10-150 with iterations=1, eps=finfo(float32).eps.
    y (B,N,F,C,J) complex64, x (B,N,F,C) complex64 -> (B,N,F,C,J)."""
    eps = torch.finfo(torch.float32).eps
    # get_local_gaussian_model, :458-494
    v = y.abs().pow(2).mean(3)                                  # (B,N,F,J)
    Cj = y.unsqueeze(4) * y.unsqueeze(3).conj()                 # (B,N,F,C,D,J)
    R = Cj.sum(1) / (v.sum(1) + eps)[:, :, None, None, :]       # (B,F,C,D,J)
    # get_mix_model + regulariser, :131-132,144,416-437
    Cxx = torch.einsum("znbs,zbcds->znbcd", v.to(R.dtype), R)
    Cxx = Cxx + math.sqrt(eps) * torch.eye(2, dtype=R.dtype)
    # _invert 2x2, :337-346
    det = Cxx[..., 0, 0] * Cxx[..., 1, 1] - Cxx[..., 0, 1] * Cxx[..., 1, 0]
    inv_det = det.reciprocal()
    inv = torch.empty_like(Cxx)
    inv[..., 0, 0] = inv_det * Cxx[..., 1, 1]
    inv[..., 1, 0] = -inv_det * Cxx[..., 1, 0]
    inv[..., 0, 1] = -inv_det * Cxx[..., 0, 1]
    inv[..., 1, 1] = inv_det * Cxx[..., 0, 0]
    # wiener_gain :353-388, apply_filter :391-413
    G = torch.einsum("zbcds,znbde->znbces", R, inv) * v[..., None, None, :]
    return torch.einsum("znbces,znbe->znbcs", G, x)


def norbert_wiener(v: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """norbert/__init__.py:153-260 with iterations=1, use_softmask=False.
    v (B,N,F,C,J) >= 0, x (B,N,F,C) complex -> (B,N,F,C,J) complex."""
    y = v * torch.exp(1j * torch.angle(x[..., None]))            # :250
    max_abs = max(1.0, float(x.abs().max()) * 0.1)               # :257
    return _em_one_iteration(y / max_abs, x / max_abs) * max_abs  # :258-260


def blockwise_wiener(X: torch.Tensor, Ymag: torch.Tensor,
                     win_len: int = WIENER_WIN) -> torch.Tensor:
    """phase.py:18-69.  X (B,2,F,S,T,2), Ymag (4,B,2,F,S,T) -> (4,B,2,F,S,T,2).
    Frames are the flattened (S,T) axis, cut in windows of <= win_len."""
    B, C, Fb, S, T, _ = X.shape
    x = torch.view_as_complex(X.reshape(B, C, Fb, S * T, 2).contiguous())
    x = x.permute(0, 3, 2, 1)                                    # (B,N,F,C)
    v = Ymag.reshape(4, B, C, Fb, S * T).permute(1, 4, 3, 2, 0)  # (B,N,F,C,J)
    N = S * T
    wl = win_len if win_len else N
    y = torch.zeros(B, N, Fb, C, 4, dtype=torch.complex64)
    for p in range(0, N, wl):
        y[:, p:p + wl] = norbert_wiener(v[:, p:p + wl], x[:, p:p + wl])
    y = torch.view_as_real(y).permute(4, 0, 3, 2, 1, 5).contiguous()
    return y.reshape(4, B, C, Fb, S, T, 2)


def unmix(sd: Dict[str, torch.Tensor], X_list: List[torch.Tensor],
          causal: bool, wiener: bool, training: bool = False, minima=None,
          masks: List[torch.Tensor] = None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """model.py:69-82 (Unmix.forward, return_masks=True).

    ``causal`` selects _CausalConv2d for layer 1 (the reference's
    ``Unmix(realtime=True)``); ``wiener`` selects blockwise_wiener over
    blockwise_phasemix_sep (model.py:264-268).  The reference ties the two
    together (realtime => causal + phasemix, offline => non-causal + Wiener);
    BASELINE config 2 is offline conv stack + phasemix, reachable there by
    flipping ``.realtime`` on the built blocks.
    """
    given, (Ys, masks) = masks, ([], [])       # ``masks``: reuse the CDAE output of an earlier call on the same X (the
    for b, X in enumerate(X_list):           # other post-filter on the same masks: oracle/separator.py separate_both)
        mag = abs_of_real_complex(X)
        m = given[b] if given is not None else cdae_masks(sd, b, mag, causal, training, minima)
        Ymag = m * mag
        Ys.append(blockwise_wiener(X, Ymag) if wiener else phasemix_sep(X, Ymag))
        masks.append(m)
    return Ys, masks
