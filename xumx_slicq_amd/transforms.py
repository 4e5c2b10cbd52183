"""Drop-in mirrors of /root/reference/xumx_slicq_v2/transforms.py on top of the
HIP library: NSGTBase, NSGT_SL, INSGT_SL, ComplexNorm, make_filterbanks.

Same names, argument meaning and error behaviour as the reference.  The
arithmetic is in csrc/slicqt.hip; these classes only own device memory
(PyTorch allocations) and hand pointers across the C ABI.  There is no CPU
path: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .arena import BlockTable
from .plan import SliCQPlan, build_plan


def make_filterbanks(nsgt_base, sample_rate=44100.0):
    """transforms.py:11-18."""
    if sample_rate != 44100.0:
        raise ValueError("i was lazy and harcoded a lot of 44100.0, forgive me")
    return NSGT_SL(nsgt_base), INSGT_SL(nsgt_base)


class SliCQEngine:
    """Device-side plan handle (the role NSGT_sliced plays in the reference,
    nsgt/slicq.py:70-230).  One handle per device, created on first use."""

    def __init__(self, plan: SliCQPlan):
        self.plan = plan
        self.sl_len = plan.L
        self.tr_area = plan.tr
        self.fs = plan.fs
        self.ncoefs = plan.ncoefs
        self.fbins_actual = plan.nbands
        self.table = BlockTable(plan.block_shapes())
        self._handles = {}
        self._ws = {}
        self._fft_backend = 0
        import os
        # measured (r02c-e): synthesising the short bands inside the inverse slice FFT kernel is 0.03-0.05 ms SLOWER per
        # 240 s track than the dense GEMM + workspace round trip it replaces (the kernel is bound by its serial chain of
        # memory / LDS round trips, and the in-kernel stage adds three) -> off by default; XSQ_SHORT_INLINE=1 / set_short_inline
        self._short_inline = os.environ.get("XSQ_SHORT_INLINE", "0") != "0"
        self._packed_fft = False        # see set_packed_fft

    def set_fft_backend(self, backend: int):
        """0 = hand-written LDS slice FFT when the plan allows it (default), 1 = rocFFT."""
        self._fft_backend = int(backend)
        for h in self._handles.values():
            _lib.check(_lib.lib.xsq_plan_set_fft_backend(h, self._fft_backend), "xsq_plan_set_fft_backend")

    def set_band_radix4(self, on: bool):
        """True (default): long bands on the radix-4 DFT kernel; False: every band on the dense GEMM."""
        self._band_radix4 = bool(on)
        for h in self._handles.values():
            _lib.check(_lib.lib.xsq_plan_set_band_radix4(h, int(self._band_radix4)), "xsq_plan_set_band_radix4")

    def set_packed_fft(self, on: bool):
        """True: the hand-written slice FFT runs its butterflies on packed-fp32 vector instructions (half the vector
        instructions, bitwise the same results).  Diagnostic builds only (csrc/Makefile PACKED_FFT=1): the product library
        refuses (``XsqError``) -- a packed-fp32 transform next to split-bf16 MFMA waves of another stream returned wrong
        values on MI355X and it measured no faster."""
        on = bool(on)
        if on == getattr(self, "_packed_fft", False):
            return
        for h in self._handles.values():
            _lib.check(_lib.lib.xsq_plan_set_packed_fft(h, int(on)), "xsq_plan_set_packed_fft")
        self._packed_fft = on

    def set_short_inline(self, on: bool):
        """False (default): bands with Lg < 24 on the dense GEMM with a round trip through the workspace; True: the inverse
        transform synthesises them inside the slice-FFT kernel (A/B switch, same results to fp32 rounding; measured slower)."""
        self._short_inline = bool(on)
        for h in self._handles.values():
            _lib.check(_lib.lib.xsq_plan_set_short_inline(h, int(self._short_inline)), "xsq_plan_set_short_inline")

    # -- handle management ---------------------------------------------------
    def handle(self, device: torch.device):
        if device.type != "cuda":
            raise _lib.XsqError(
                f"the sliCQT runs on a ROCm device only (got a tensor on '{device}'); there is no CPU fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            p = self.plan
            out = C.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(_lib.lib.xsq_plan_create(
                    C.byref(out), p.L, p.tr, p.nbands,
                    p.Lg.ctypes.data, p.c.ctypes.data, p.g.ctypes.data, p.gd.ctypes.data, p.tw.ctypes.data),
                    "xsq_plan_create")
            h = out
            _lib.check(_lib.lib.xsq_plan_set_fft_backend(h, self._fft_backend), "xsq_plan_set_fft_backend")
            _lib.check(_lib.lib.xsq_plan_set_band_radix4(h, int(getattr(self, "_band_radix4", True))),
                       "xsq_plan_set_band_radix4")
            _lib.check(_lib.lib.xsq_plan_set_short_inline(h, int(getattr(self, "_short_inline", False))),
                       "xsq_plan_set_short_inline")
            _lib.check(_lib.lib.xsq_plan_set_packed_fft(h, int(getattr(self, "_packed_fft", False))), "xsq_plan_set_packed_fft")
            self._handles[idx] = h
        return h

    def demixer(self, device: torch.device):
        """The native whole-call handle of this plan on ``device`` (xsq_demixer: cached row-offset tables of the call
        shapes + the fork / join events of the tail stream); what ``Separator.forward`` issues a track through."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        d = self.__dict__.setdefault("_demixers", {}).get(idx)
        if d is None:
            h = self.handle(device)
            out = C.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(_lib.lib.xsq_demixer_create(C.byref(out), h), "xsq_demixer_create")
            d = self._demixers[idx] = out
        return d

    def workspace(self, device: torch.device, nbytes: int) -> Tensor:
        """Grow-only scratch buffer per (device, stream): calls issued on different streams may overlap
        (Separator.forward runs the tail chunk beside the stacked pass). PyTorch owns the memory."""
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = None
            self._ws[key] = None
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def __del__(self):
        try:
            for d in self.__dict__.get("_demixers", {}).values():
                _lib.lib.xsq_demixer_destroy(d)
            for h in self._handles.values():
                _lib.lib.xsq_plan_destroy(h)
        except Exception:
            pass

    def __deepcopy__(self, memo):
        """The device plan handles and workspaces stay with the original (a copied raw handle would be
        destroyed twice); the copy creates its own on first use."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k in ("_handles", "_ws", "_demixers") else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_handles"], st["_ws"] = {}, {}
        st.pop("_demixers", None)
        return st

    # -- transforms --------------------------------------------------------------
    def forward(self, x: Tensor, whiten=None):
        """x (*lead, n) fp32 on a ROCm device -> (arena, lead, S).  ``whiten`` = (xin device pointer, mean pointer,
        scale pointer, split flag) from ``Unmix.whitening_target``: the analysis kernels also write the CDAE's
        whitened magnitude there (xsq_slicqt_forward_xin)."""
        if x.dtype != torch.float32:
            x = x.float()
        lead = tuple(x.shape[:-1])
        n = x.shape[-1]
        xb = x.contiguous().view(-1, n)
        BC = xb.shape[0]
        h = self.handle(x.device)
        with torch.cuda.device(x.device):
            S = self.plan.num_slices(n)
            arena = torch.empty(self.table.numel(BC, S), dtype=torch.float32, device=x.device)
            nbytes = _lib.lib.xsq_slicqt_forward_workspace(h, BC, n)
            if nbytes == 0:
                raise _lib.XsqError("xsq_slicqt_forward_workspace: " + _lib.last_error())
            ws = self.workspace(x.device, nbytes)
            if whiten is None:
                _lib.check(_lib.lib.xsq_slicqt_forward(h, xb.data_ptr(), BC, n, arena.data_ptr(), ws.data_ptr(),
                                                       ws.numel(), _lib.stream_ptr()), "xsq_slicqt_forward")
            else:
                xin, mean, scale, split = whiten
                _lib.check(_lib.lib.xsq_slicqt_forward_xin(h, xb.data_ptr(), BC, n, arena.data_ptr(), xin, mean, scale,
                                                           int(split), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                           "xsq_slicqt_forward_xin")
        return arena, lead, S

    def backward(self, arena: Tensor, BC: int, S: int, length: int, out: Tensor = None,
                 row_offsets: Tensor = None) -> Tensor:
        """arena for BC channels, S slices -> (BC, length) fp32.  `arena` is left untouched.
        With ``out`` + ``row_offsets`` (device int64[BC], element offsets into ``out``) packed
        channel r lands at out.view(-1)[row_offsets[r] : row_offsets[r] + length]."""
        h = self.handle(arena.device)
        with torch.cuda.device(arena.device):
            y = out if out is not None else torch.empty(BC, length, dtype=torch.float32, device=arena.device)
            nbytes = _lib.lib.xsq_slicqt_inverse_workspace(h, BC, S)
            if nbytes == 0:
                raise _lib.XsqError("xsq_slicqt_inverse_workspace: " + _lib.last_error())
            ws = self.workspace(arena.device, nbytes)
            if out is not None:
                assert row_offsets is not None and row_offsets.dtype == torch.int64 and row_offsets.numel() == BC
                assert out.is_contiguous() and out.dtype == torch.float32
            _lib.check(_lib.lib.xsq_slicqt_inverse_rows(
                h, arena.data_ptr(), BC, S, length, y.data_ptr(),
                row_offsets.data_ptr() if row_offsets is not None else None,
                ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "xsq_slicqt_inverse_rows")
        return y

    def backward_masked(self, masks: Tensor, mix: Tensor, BC: int, BCx: int, S: int, length: int, out: Tensor,
                        row_offsets: Tensor) -> Tensor:
        """Inverse transform of masks[bc] * mix[bc % BCx] without materialising the product
        (the separator's mix-phase path): masks = real arena for BC channels, mix = complex arena
        for BCx channels.  Output placement as in ``backward``."""
        h = self.handle(masks.device)
        with torch.cuda.device(masks.device):
            nbytes = _lib.lib.xsq_slicqt_inverse_workspace(h, BC, S)
            if nbytes == 0:
                raise _lib.XsqError("xsq_slicqt_inverse_workspace: " + _lib.last_error())
            ws = self.workspace(masks.device, nbytes)
            assert row_offsets.dtype == torch.int64 and row_offsets.numel() == BC
            assert out.is_contiguous() and out.dtype == torch.float32
            _lib.check(_lib.lib.xsq_slicqt_inverse_masked(
                h, masks.data_ptr(), mix.data_ptr(), BC, BCx, S, length, out.data_ptr(), row_offsets.data_ptr(),
                ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "xsq_slicqt_inverse_masked")
        return out


class NSGTBase(nn.Module):
    """transforms.py:21-94.  Holds the plan; `.nsgt` is the device engine."""

    def __init__(self, scale, fbins, fmin, fmax=22050.0, fgamma=15.0, fs=44100.0, device="cuda"):
        super().__init__()
        self.fbins = fbins
        self.fmin = fmin
        self.fmax = fmax
        self.plan = build_plan(scale, fbins, fmin, fmax, fs)
        self.sllen, self.trlen = self.plan.L, self.plan.tr
        print(f"scale={scale}, fbins={fbins}, fmin={fmin:.2f}, fmax={fmax:.2f}, "
              f"sllen={self.sllen}, trlen={self.trlen}")
        self.nsgt = SliCQEngine(self.plan)
        self.M = self.nsgt.ncoefs
        self.fs = fs
        self.fbins_actual = self.nsgt.fbins_actual

    def predict_input_size(self, batch_size, nb_channels, seq_dur_s):
        """transforms.py:80-90.  The block shapes follow from the plan, so no
        transform is run (and the global RNG is left alone);
        the returned list holds zero tensors of the right shapes."""
        n = int(seq_dur_s * self.fs)
        S = self.plan.num_slices(n)
        x = torch.zeros((batch_size, nb_channels, n), dtype=torch.float32)
        jag = [torch.zeros((batch_size, nb_channels, F, S, T, 2), dtype=torch.float32)
               for (_, F, T) in self.plan.blocks]
        return jag, x


class NSGT_SL(nn.Module):
    """transforms.py:97-131."""

    def __init__(self, nsgt):
        super().__init__()
        self.nsgt = nsgt

    def forward(self, x: Tensor) -> List[Tensor]:
        """(nb_samples, nb_channels, nb_timesteps) -> list over blocks of
        (nb_samples, nb_channels, F_b, nb_slices, T_b, 2), views of one arena."""
        eng = self.nsgt.nsgt
        arena, lead, S = eng.forward(x)
        return eng.table.views(arena, lead, S)


class INSGT_SL(nn.Module):
    """transforms.py:134-178.  Accepts 6-D (B,C,F,S,T,2) or 7-D (4,B,C,F,S,T,2) blocks."""

    def __init__(self, nsgt):
        super().__init__()
        self.nsgt = nsgt

    def forward(self, X_list, length: int) -> Tensor:
        eng = self.nsgt.nsgt
        arena, lead, S = eng.table.as_arena(list(X_list))
        BC = int(np.prod(lead)) if len(lead) else 1
        y = eng.backward(arena, BC, S, int(length))
        return y.view(*lead, -1)


class ComplexNorm(nn.Module):
    """transforms.py:181-208: magnitude of a block list or of one tensor."""

    def forward(self, spec):
        if isinstance(spec, list):
            return [torch.abs(torch.view_as_complex(b)) for b in spec]
        if isinstance(spec, Tensor):
            return self.forward([spec])[0]
        raise ValueError(f"unsupported type for 'spec': {type(spec)}")
