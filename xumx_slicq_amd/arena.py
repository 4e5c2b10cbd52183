"""Coefficient arena: the reference's ragged list of per-block tensors kept in
ONE device allocation (layout: include/xumx_slicq_hip.h, "Coefficient arena").

Block b of shape (*lead, F_b, S, T_b[, 2]) starts at element offset
prod(lead) * S * cum_b (* 2 for complex) -- so the list the reference API
requires is a set of zero-copy views, and the kernels see one buffer plus an
offset table.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


class BlockTable:
    def __init__(self, shapes: Sequence[Tuple[int, int]]):
        self.shapes = [(int(F), int(T)) for F, T in shapes]
        self.coefs_per_slice = sum(F * T for F, T in self.shapes)

    def __len__(self):
        return len(self.shapes)

    def numel(self, nchan: int, S: int, complex_: bool = True) -> int:
        return (2 if complex_ else 1) * nchan * S * self.coefs_per_slice

    def views(self, arena: Tensor, lead: Tuple[int, ...], S: int, complex_: bool = True) -> List[Tensor]:
        nchan = int(np.prod(lead)) if len(lead) else 1
        tail = (2,) if complex_ else ()
        out, off = [], 0
        for (F, T) in self.shapes:
            n = (2 if complex_ else 1) * nchan * F * S * T
            out.append(arena[off: off + n].view(*lead, F, S, T, *tail))
            off += n
        return out

    def as_arena(self, X_list: Sequence[Tensor]):
        """(arena, lead, S) of a complex block list; zero-copy when the list already
        is a run of views laid out back to back in one allocation."""
        if len(X_list) != len(self.shapes):
            raise ValueError(f"expected {len(self.shapes)} blocks, got {len(X_list)}")
        x0 = X_list[0]
        if x0.dim() < 5 or x0.shape[-1] != 2:
            raise ValueError(f"blocks must be (..., F, S, T, 2); got {tuple(x0.shape)}")
        lead = tuple(x0.shape[:-4])
        S = int(x0.shape[-3])
        nchan = int(np.prod(lead)) if len(lead) else 1
        total = self.numel(nchan, S)
        run = True
        ptr = x0.data_ptr()
        for X, (F, T) in zip(X_list, self.shapes):
            if tuple(X.shape) != (*lead, F, S, T, 2):
                raise ValueError(f"block has shape {tuple(X.shape)}, expected {(*lead, F, S, T, 2)}")
            if run and (X.dtype != torch.float32 or not X.is_contiguous() or X.data_ptr() != ptr):
                run = False
            ptr += X.numel() * 4
        if run:
            st = x0.untyped_storage()
            run = st.nbytes() - (x0.data_ptr() - st.data_ptr()) >= total * 4
        if run:
            arena = torch.as_strided(x0, (total,), (1,))
        else:
            arena = torch.cat([X.to(torch.float32).reshape(-1) for X in X_list])
        return arena, lead, S
