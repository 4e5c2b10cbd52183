"""Seeded synthetic CDAE weights in the reference's ``state_dict`` layout.

The pretrained ``.pth`` files of the reference are Git-LFS pointers and there
is no network, so parity tests, the smoke run and the bench all use weights
drawn by one rule from a NumPy PCG64 stream ("Weights").  The
keys and their order are exactly those of the reference ``Unmix.state_dict()``
(/root/reference/xumx_slicq_v2/model.py:130-203), so a real
checkpoint drops into the same loaders and this mapping loads into the
reference model with ``load_state_dict``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterable, Tuple

import numpy as np
import torch

HIDDEN_1 = 50   # model.py:92
HIDDEN_2 = 51   # model.py:93
TIME_FILTER_2 = 4  # model.py:99
NB_TARGETS = 4
NB_CHANNELS = 2


def freq_filter(nb_f_bins: int) -> int:
    """Frequency kernel height of a block (model.py:112-117)."""
    if nb_f_bins < 10:
        return 1
    if nb_f_bins < 20:
        return 3
    return 5


def state_dict_spec(blocks: Iterable[Tuple[int, int]]):
    """Yield (key, shape, kind) in the reference's state_dict order.

    ``blocks`` is the block table [(F_b, T_b)].  ``kind`` picks the draw rule.
    """
    for b, (F, T) in enumerate(blocks):
        kf = freq_filter(F)
        pre = f"sliced_umx.{b}."
        yield pre + "input_mean", (F,), "in_mean"
        yield pre + "input_scale", (F,), "in_scale"
        for t in range(NB_TARGETS):
            p = f"{pre}cdaes.{t}."
            yield p + "0.weight", (HIDDEN_1, NB_CHANNELS, kf, T), "conv"
            for k in _bn(p + "1", HIDDEN_1):
                yield k
            yield p + "3.weight", (HIDDEN_2, HIDDEN_1, kf, TIME_FILTER_2), "conv"
            for k in _bn(p + "4", HIDDEN_2):
                yield k
            # ConvTranspose2d weights are (in, out, kH, kW)
            yield p + "6.weight", (HIDDEN_2, HIDDEN_1, kf, TIME_FILTER_2), "convT"
            for k in _bn(p + "7", HIDDEN_1):
                yield k
            yield p + "9.weight", (HIDDEN_1, NB_CHANNELS, kf, T), "convT_out"
            yield p + "9.bias", (NB_CHANNELS,), "out_bias"


def _bn(prefix: str, n: int):
    yield prefix + ".weight", (n,), "bn_w"
    yield prefix + ".bias", (n,), "bn_b"
    yield prefix + ".running_mean", (n,), "bn_m"
    yield prefix + ".running_var", (n,), "bn_v"
    yield prefix + ".num_batches_tracked", (), "bn_n"


def seeded_state_dict(blocks: Iterable[Tuple[int, int]], seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """Draw every tensor of the state_dict, in key order, from PCG64(seed).

    Convolutions get a Kaiming-style uniform bound so activations stay O(1)
    and the sigmoid masks are neither saturated nor flat; BatchNorm running
    statistics and affine terms are non-trivial so that folding them is tested.
    """
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape, kind in state_dict_spec(blocks):
        if kind == "bn_n":
            sd[key] = torch.tensor(100, dtype=torch.long)
            continue
        if kind == "conv":            # (out, in, kH, kW): fan_in = in*kH*kW
            bound = np.sqrt(6.0 / (shape[1] * shape[2] * shape[3]))
            a = rng.uniform(-bound, bound, shape)
        elif kind == "convT":         # (in, out, kH, kW): fan_in = in*kH*kW
            bound = np.sqrt(6.0 / (shape[0] * shape[2] * shape[3]))
            a = rng.uniform(-bound, bound, shape)
        elif kind == "convT_out":     # strided: two time taps per output sample
            bound = np.sqrt(6.0 / (shape[0] * shape[2] * 2))
            a = rng.uniform(-bound, bound, shape)
        elif kind == "out_bias":
            a = rng.uniform(-0.5, 0.5, shape)
        elif kind == "bn_w":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind in ("bn_b", "bn_m"):
            a = rng.uniform(-0.2, 0.2, shape)
        elif kind == "bn_v":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "in_mean":       # stored as -mean (model.py:192-195)
            a = rng.uniform(-1.0, 0.0, shape)
        elif kind == "in_scale":      # stored as 1/std (model.py:197-200)
            a = rng.uniform(0.5, 1.5, shape)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(a.astype(np.float32))
    return sd
