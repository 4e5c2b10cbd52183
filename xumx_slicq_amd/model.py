"""Drop-in mirror of /root/reference/xumx_slicq_v2/model.py (Unmix,
_SlicedUnmixCDAE, _CausalConv2d) on top of the HIP library.

The modules hold the parameters under exactly the reference's ``state_dict``
keys (the same ``nn.Sequential`` skeleton), so reference checkpoints load
unchanged; they are parameter holders only.  ``Unmix.forward`` packs the
parameters once (BatchNorm folded in csrc/cdae.hip) and runs all 70 blocks x 4
targets in grouped HIP launches.  There is no CPU path.
"""
from __future__ import annotations

import copy
import ctypes as C
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.nn import BatchNorm2d, Conv2d, ConvTranspose2d, Parameter, ReLU, Sequential, Sigmoid

from . import _lib
from .arena import BlockTable
from .weights import freq_filter

import os

# xsq_model_set_precision modes (include/xumx_slicq_hip.h)
_PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x6": 2}

# Owners (models, trainers) whose contractions currently run on split-bf16 MFMAs.  While the set is non-empty no
# packed-fp32 slice FFT may be launched in this process (``split_bf16_active``; SliCQEngine.set_packed_fft).
_SPLIT_BF16_OWNERS = set()


def note_precision(owner, precision: str):
    (_SPLIT_BF16_OWNERS.discard if precision == "fp32" else _SPLIT_BF16_OWNERS.add)(id(owner))


def split_bf16_active() -> bool:
    return bool(_SPLIT_BF16_OWNERS)


class _CausalConv2d(Conv2d):
    """model.py:274-290: left-pads time by kernel_width-1 (parameter holder here)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True):
        self._left_pad = kernel_size[1] - 1
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=0, bias=bias)


class _no_parameter_init:
    """Context: Conv / BatchNorm modules constructed inside leave their parameters as allocated and ZERO them instead of drawing
    the Kaiming / unit initialisation (``reset_parameters`` is what costs: a third of Unmix() for weights a checkpoint replaces
    a moment later).  Not re-entrant; construction is single-threaded."""

    def __enter__(self):
        from torch.nn.modules.batchnorm import _NormBase
        from torch.nn.modules.conv import _ConvNd
        self._saved = (_ConvNd.reset_parameters, _NormBase.reset_parameters)

        def zero(mod):
            with torch.no_grad():
                for t in list(mod._parameters.values()) + list(mod._buffers.values()):
                    if t is not None:
                        t.zero_()
        _ConvNd.reset_parameters = zero
        _NormBase.reset_parameters = zero
        return self

    def __exit__(self, *exc):
        from torch.nn.modules.batchnorm import _NormBase
        from torch.nn.modules.conv import _ConvNd
        _ConvNd.reset_parameters, _NormBase.reset_parameters = self._saved
        return False


class _SlicedUnmixCDAE(nn.Module):
    """model.py:86-211: the four target CDAEs of one time-frequency block."""

    def __init__(self, slicq_sample_input, hidden_size_1: int = 50, hidden_size_2: int = 51,
                 freq_filter_small: int = 1, freq_filter_medium: int = 3, freq_filter_large: int = 5,
                 freq_thresh_small: int = 10, freq_thresh_medium: int = 20, time_filter_2: int = 4,
                 realtime: bool = False, input_mean=None, input_scale=None, weights_follow: bool = False):
        super().__init__()
        if (hidden_size_1, hidden_size_2, time_filter_2) != (50, 51, 4) or \
                (freq_filter_small, freq_filter_medium, freq_filter_large,
                 freq_thresh_small, freq_thresh_medium) != (1, 3, 5, 10, 20):
            raise ValueError("the HIP kernels are built for the reference's default CDAE geometry "
                             "(hidden 50/51, time filter 4, frequency filters 1/3/5 at thresholds 10/20)")
        nb_samples, nb_channels, nb_f_bins, nb_slices, nb_t_bins = slicq_sample_input.shape
        if nb_channels != 2:
            raise ValueError("the hot path is stereo (nb_channels == 2)")
        kf = freq_filter(nb_f_bins)
        window, hop = nb_t_bins, nb_t_bins // 2
        first = _CausalConv2d if realtime else Conv2d

        def stack(make):
            return Sequential(
                make(first, nb_channels, hidden_size_1, (kf, window), stride=(1, hop), bias=False),
                make(BatchNorm2d, hidden_size_1), ReLU(),
                make(Conv2d, hidden_size_1, hidden_size_2, (kf, time_filter_2), bias=False),
                make(BatchNorm2d, hidden_size_2), ReLU(),
                make(ConvTranspose2d, hidden_size_2, hidden_size_1, (kf, time_filter_2), bias=False),
                make(BatchNorm2d, hidden_size_1), ReLU(),
                make(ConvTranspose2d, hidden_size_1, nb_channels, (kf, window), stride=(1, hop), bias=True),
                Sigmoid())
        if weights_follow:
            # a checkpoint is loaded right after construction (build_models(state=...)): no Kaiming initialisation of 15 M
            # weights that are about to be overwritten and no deep copies (0.7 of the 0.9 s of Unmix() -- what a user of
            # Separator.load waits for, tools/cold_start.py); parameters and BatchNorm statistics start as ZEROS, strict
            # loading guarantees that every one of them is replaced
            with _no_parameter_init():
                self.cdaes = nn.ModuleList([stack(lambda cls, *a, **k: cls(*a, **k)) for _ in range(4)])
        else:
            cdae = stack(lambda cls, *a, **k: cls(*a, **k))
            self.cdaes = nn.ModuleList([cdae] + [copy.deepcopy(cdae) for _ in range(3)])
        self.mask = True
        self.realtime = realtime          # read per call: True -> mix-phase, False -> Wiener-EM (model.py:264)
        self.causal = realtime            # fixed at construction (model.py:125-128)
        self.nb_f_bins, self.nb_t_bins = nb_f_bins, nb_t_bins
        mean = torch.from_numpy(-input_mean).float() if input_mean is not None else torch.zeros(nb_f_bins)
        scale = torch.from_numpy(1.0 / input_scale).float() if input_scale is not None else torch.ones(nb_f_bins)
        self.input_mean = Parameter(mean)
        self.input_scale = Parameter(scale)

    def freeze(self):
        for p in self.parameters():
            p.grad = None
        self.eval()

    # -- the reference's per-block call, sliced_umx[i](Xblock, abs(Xblock)) (model.py:76-80, 213-271) ------
    def _block_model(self, device: torch.device):
        """A one-block xsq_model of this block's parameters, rebuilt when any of them changed."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        sd = self.state_dict()
        key = tuple((v.data_ptr(), v._version) for v in sd.values())
        cached = self.__dict__.setdefault("_block_handles", {}).get(idx)
        if cached is not None and cached[0] == key:
            return cached[1]
        if cached is not None:
            _lib.lib.xsq_model_destroy(cached[1])
        params = torch.cat([v.detach().to("cpu", torch.float32).reshape(-1) for k, v in sd.items()
                            if not k.endswith("num_batches_tracked")]).numpy()
        F_ = np.asarray([self.nb_f_bins], dtype=np.int32)
        T_ = np.asarray([self.nb_t_bins], dtype=np.int32)
        out = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib.xsq_model_create(C.byref(out), 1, F_.ctypes.data, T_.ctypes.data,
                                                 1 if self.causal else 0, params.ctypes.data, params.size),
                       "xsq_model_create")
        self._block_handles[idx] = (key, out)
        return out

    def __del__(self):
        try:
            for _, h in self.__dict__.get("_block_handles", {}).values():
                _lib.lib.xsq_model_destroy(h)
        except Exception:
            pass

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_block_handles", None)
        st.pop("_block_ws", None)
        return st

    def forward(self, xcomplex: Tensor, x: Tensor):
        """(B, 2, F, S, T, 2), (B, 2, F, S, T) -> (estimates (4, B, 2, F, S, T, 2), masks (4, B, 2, F, S, T)),
        model.py:213-271.  ``x`` is ``abs_of_real_complex(xcomplex)`` in every caller of the reference
        (model.py:74-76); the kernel forms the magnitude from ``xcomplex`` itself, ``x`` only has its shape
        checked and is never modified (the reference whitens it in place).  One launch set for the four targets
        of this block; ``Unmix.forward`` runs all 70 blocks in one."""
        from .phase import wiener_em_arena
        if self.training:
            raise _lib.XsqError("the HIP CDAE is the inference path (BatchNorm folded); call .eval()/.freeze()")
        F_, T_ = self.nb_f_bins, self.nb_t_bins
        if xcomplex.dim() != 6 or tuple(xcomplex.shape[1:3]) != (2, F_) or tuple(xcomplex.shape[4:]) != (T_, 2):
            raise ValueError(f"expected xcomplex (B, 2, {F_}, S, {T_}, 2); got {tuple(xcomplex.shape)}")
        if tuple(x.shape) != tuple(xcomplex.shape[:-1]):
            raise ValueError(f"x must be the magnitude of xcomplex, shape {tuple(xcomplex.shape[:-1])}; got {tuple(x.shape)}")
        if xcomplex.device.type != "cuda":
            raise _lib.XsqError(f"the CDAE runs on a ROCm device only (got '{xcomplex.device}'); there is no CPU fallback")
        if not self.mask:
            raise _lib.XsqError("the HIP CDAE applies the multiplicative skip connection (mask=True, the reference's default)")
        B = xcomplex.shape[0]
        S = xcomplex.shape[3]
        dev = xcomplex.device
        X = xcomplex.contiguous().float().view(-1)
        h = self._block_model(dev)
        table = BlockTable([(F_, T_)])
        with torch.cuda.device(dev):
            Y = torch.empty(4, B, 2, F_, S, T_, 2, dtype=torch.float32, device=dev)
            masks = torch.empty(4, B, 2, F_, S, T_, dtype=torch.float32, device=dev)
            nbytes = _lib.lib.xsq_cdae_workspace(h, B, S)
            if nbytes == 0:
                raise _lib.XsqError(f"xsq_cdae_workspace(B={B}, S={S}) failed: need at least 3 slices")
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            wss = self.__dict__.setdefault("_block_ws", {})
            ws = wss.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = wss[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib.xsq_cdae_forward(h, X.data_ptr(), B, S, Y.data_ptr(), masks.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "xsq_cdae_forward")
            if not self.realtime:
                wiener_em_arena(table, X, Y.view(-1), B, S)
        return Y, masks


class Unmix(nn.Module):
    """model.py:29-82."""

    def __init__(self, jagged_slicq_sample_input, realtime: bool = False, lstm: bool = False,
                 input_means=None, input_scales=None, weights_follow: bool = False):
        super().__init__()
        if lstm:
            raise ValueError("the LSTM variant is not on the accelerated path (SURVEY.md 2, row 4b)")
        self.sliced_umx = nn.ModuleList()
        shapes = []
        for i, C_block in enumerate(jagged_slicq_sample_input):
            self.sliced_umx.append(_SlicedUnmixCDAE(
                C_block, realtime=realtime,
                input_mean=input_means[i] if input_means else None,
                input_scale=input_scales[i] if input_scales else None, weights_follow=weights_follow))
            shapes.append((C_block.shape[2], C_block.shape[4]))
        self.table = BlockTable(shapes)
        self._F = np.asarray([s[0] for s in shapes], dtype=np.int32)
        self._T = np.asarray([s[1] for s in shapes], dtype=np.int32)
        self._handles = {}      # device index -> (version, handle)
        self._ws = {}
        self.precision = os.environ.get("XSQ_CDAE_PRECISION", "fp32")
        if self.precision not in _PRECISIONS:
            raise ValueError(f"XSQ_CDAE_PRECISION={self.precision!r} not in {sorted(_PRECISIONS)}")
        note_precision(self, self.precision)

    def freeze(self):
        for p in self.parameters():
            p.grad = None
        self.eval()

    # -- parameter packing -------------------------------------------------------
    # The folded device copy is rebuilt whenever the parameters can have changed through
    # the module API (load_state_dict, .to()/.float()/..., .train()); after editing
    # parameter tensors in place by hand, call refresh().
    def refresh(self):
        self._stamp = getattr(self, "_stamp", 0) + 1

    def _version(self) -> int:
        return getattr(self, "_stamp", 0)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """nn.Module.load_state_dict; a strict load of a well-formed checkpoint (same keys, same shapes) takes a direct
        path -- one ``copy_`` per tensor instead of a walk over 3,500 modules (0.1 s of Separator.load); anything else goes
        through the module walk and fails (or reports) exactly as torch does."""
        self.refresh()
        if strict and not assign:
            own = self.__dict__.get("_own_tensors")
            if own is None:
                own = self.__dict__["_own_tensors"] = {**dict(self.named_parameters()), **dict(self.named_buffers())}
            if len(own) == len(state_dict) and all(k in own and own[k].shape == v.shape for k, v in state_dict.items()):
                with torch.no_grad():
                    for k, v in state_dict.items():
                        own[k].copy_(v)
                from torch.nn.modules.module import _IncompatibleKeys
                return _IncompatibleKeys([], [])
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_own_tensors", None)            # (.to(device) may replace parameter objects)
        self.refresh()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        if bool(mode) != bool(self.training):     # an idempotent .eval() must not force a 60 MB repack + handle rebuild
            self.refresh()
        return super().train(mode)

    def __deepcopy__(self, memo):
        """Copies share nothing with the original on the device side: the ctypes handles and workspaces are
        dropped (a copied raw handle would be freed twice) and rebuilt on the copy's first forward."""
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k in ("_handles", "_ws") else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_handles"], st["_ws"] = {}, {}
        return st

    def packed_parameters(self) -> np.ndarray:
        """fp32 tensors of the state_dict in key order, num_batches_tracked left out
        (the host buffer xsq_model_create expects)."""
        # (state_dict order = registration order: parameters, then buffers, per module; built without the state_dict's
        #  OrderedDict + hooks and without 5,740 .to() calls when everything already is fp32 on one device: 0.46 -> 0.05 s)
        parts = [v.detach().reshape(-1) for k, v in self.state_dict(keep_vars=True).items() if not k.endswith("num_batches_tracked")]
        if all(p.dtype == torch.float32 for p in parts) and len({p.device for p in parts}) == 1:
            return torch.cat(parts).cpu().numpy()
        return torch.cat([p.to("cpu", torch.float32) for p in parts]).numpy()

    def _model(self, device: torch.device):
        if device.type != "cuda":
            raise _lib.XsqError(f"Unmix runs on a ROCm device only (got '{device}'); there is no CPU fallback")
        if self.training:
            raise _lib.XsqError("the HIP CDAE is the inference path (BatchNorm folded); call .eval()/.freeze()")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ver = self._version()
        cached = self._handles.get(idx)
        if cached is not None and cached[0] == ver:
            return cached[1]
        if cached is not None:
            _lib.lib.xsq_model_destroy(cached[1])
        params = self.packed_parameters()
        causal = {blk.causal for blk in self.sliced_umx}
        if len(causal) != 1:
            raise _lib.XsqError("all blocks must share the same first-layer type")
        out = C.c_void_p()
        with torch.cuda.device(idx):
            _lib.check(_lib.lib.xsq_model_create(
                C.byref(out), len(self.table), self._F.ctypes.data, self._T.ctypes.data,
                1 if causal.pop() else 0, params.ctypes.data, params.size), "xsq_model_create")
            # inside the device guard: bf16 modes allocate and launch on the CURRENT device
            _lib.check(_lib.lib.xsq_model_set_precision(out, _PRECISIONS[self.precision]), "xsq_model_set_precision")
            _lib.check(_lib.lib.xsq_model_set_winograd(out, int(getattr(self, "winograd", 7))), "xsq_model_set_winograd")
        self._handles[idx] = (ver, out)
        return out

    def set_precision(self, precision: str):
        """Arithmetic of the convolution contractions: "fp32" (exact, v_mfma_f32_32x32x2_f32) or "bf16x3"
        (fp32 operands split into hi + lo bf16, three bf16 MFMAs per product, fp32 accumulation)."""
        if precision not in _PRECISIONS:
            raise ValueError(f"precision {precision!r} not in {sorted(_PRECISIONS)}")
        self.precision = precision
        note_precision(self, precision)
        for idx, (_ver, h) in self._handles.items():
            with torch.cuda.device(idx):       # the split-weight pool is allocated / converted on the model's device
                _lib.check(_lib.lib.xsq_model_set_precision(h, _PRECISIONS[precision]), "xsq_model_set_precision")

    def set_winograd(self, on):
        """Fast-convolution forms of the fp32 inference layers.  True (default) = all of them, False = the direct kernels, an int =
        a bit mask: 1 = layers 2 / 3 of long rows as Winograd F(2, 4) along the four time taps (csrc/cdae_wino.h), 2 / 4 = layer 1 /
        layer 4 as F(2, 2) along the hop (csrc/cdae_l1f.h, csrc/cdae_l4f.h), 8 = layers 2 / 3 of rows >= 253 as F(4, 4) (csrc/cdae_wino4.h: an A/B
        arm, measured slower; needs a model created with XSQ_WINO4=1).  Same result to fp32 rounding (~2e-7 of a layer's output)."""
        self.winograd = (7 if on else 0) if isinstance(on, bool) else int(on)
        for _ver, h in self._handles.values():
            _lib.check(_lib.lib.xsq_model_set_winograd(h, self.winograd), "xsq_model_set_winograd")

    def __del__(self):
        try:
            _SPLIT_BF16_OWNERS.discard(id(self))
            for _, h in self._handles.values():
                _lib.lib.xsq_model_destroy(h)
        except Exception:
            pass

    def _workspace(self, device, nbytes):
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(device).cuda_stream)      # per stream: see SliCQEngine.workspace
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws[key] = None
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    # -- forward -----------------------------------------------------------------------
    def whitening_target(self, device: torch.device, B: int, S: int):
        """(workspace, mean pointer, scale pointer, split flag): where the forward transform may write the whitened
        magnitude of a (B, 2, ...) input with S slices (``SliCQEngine.forward(x, whiten=...)``) so that the next
        ``masks_arena`` / ``forward`` call on the same stream runs with ``xin_ready=True`` and skips its magnitude pass."""
        h = self._model(device)
        with torch.cuda.device(device):
            nbytes = _lib.lib.xsq_cdae_workspace(h, B, S)
            if nbytes == 0:
                raise _lib.XsqError(f"xsq_cdae_workspace(B={B}, S={S}) failed: need at least 3 slices")
            ws = self._workspace(device, nbytes)
        mean, scale, split = C.c_void_p(), C.c_void_p(), C.c_int()
        _lib.check(_lib.lib.xsq_model_whitening(h, C.byref(mean), C.byref(scale), C.byref(split)), "xsq_model_whitening")
        return ws, mean.value, scale.value, split.value

    def masks_arena(self, Xcomplex: List[Tensor], xin_ready: bool = False):
        """Mix-phase models only: (masks, X, B, S) with ``masks`` the real arena (4, B, 2, ...) of sigmoid
        masks and ``X`` the mix arena -- the estimate Y = masks * X is left to the consumer
        (``SliCQEngine.backward_masked`` forms it while it loads; Separator.forward uses this)."""
        X, lead, S = self.table.as_arena(list(Xcomplex))
        if len(lead) != 2 or lead[1] != 2:
            raise ValueError(f"expected blocks of shape (nb_samples, 2, F, S, T, 2); got lead dims {lead}")
        if not all(bool(blk.realtime) for blk in self.sliced_umx):
            raise _lib.XsqError("masks_arena: the Wiener-EM post-filter needs the estimates; use forward()")
        B = lead[0]
        dev = X.device
        h = self._model(dev)
        with torch.cuda.device(dev):
            masks = torch.empty(self.table.numel(8 * B, S, complex_=False), dtype=torch.float32, device=dev)
            nbytes = _lib.lib.xsq_cdae_workspace(h, B, S)
            if nbytes == 0:
                raise _lib.XsqError(f"xsq_cdae_workspace(B={B}, S={S}) failed: need at least 3 slices")
            ws = self._workspace(dev, nbytes)
            _lib.check(_lib.lib.xsq_cdae_forward_xin(h, X.data_ptr(), B, S, None, masks.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr(), int(bool(xin_ready))),
                       "xsq_cdae_forward")
        return masks, X, B, S

    def forward(self, Xcomplex: List[Tensor], return_masks=False, wiener_batch_group: int = 0, xin_ready: bool = False):
        """list over blocks of (B, 2, F_b, S, T_b, 2) -> list of (4, B, 2, F_b, S, T_b, 2)
        [+ masks (4, B, 2, F_b, S, T_b)].  model.py:69-82.  ``wiener_batch_group`` (extension):
        runs of that many batch items share the Wiener window maximum (0 = the whole batch, the
        reference's behaviour); Separator uses it to stack independent chunks along the batch."""
        from .phase import wiener_em_arena, wiener_em_masked_arena
        X, lead, S = self.table.as_arena(list(Xcomplex))
        if len(lead) != 2 or lead[1] != 2:
            raise ValueError(f"expected blocks of shape (nb_samples, 2, F, S, T, 2); got lead dims {lead}")
        B = lead[0]
        dev = X.device
        h = self._model(dev)
        modes = {bool(blk.realtime) for blk in self.sliced_umx}
        if len(modes) != 1:
            raise _lib.XsqError("mixed per-block post-filters (some mix-phase, some Wiener) are not supported")
        phasemix = modes.pop()
        # Wiener-EM from the masks (default): the last layer stores the real masks only and both EM passes form the
        # initial estimate mask * X while they load -- same bits, a third less traffic.  ``wiener_masked = False`` /
        # XSQ_WIENER_MASKED=0 restores the two-step form (layer 4 writes mask * X, the EM refines it in place).
        masked = (not phasemix and all(int(t) % 2 == 0 for t in self._T) and
                  bool(getattr(self, "wiener_masked", os.environ.get("XSQ_WIENER_MASKED", "1") != "0")))
        with torch.cuda.device(dev):
            Y = torch.empty(self.table.numel(8 * B, S), dtype=torch.float32, device=dev)
            masks = torch.empty(self.table.numel(8 * B, S, complex_=False), dtype=torch.float32,
                                device=dev) if (return_masks or masked) else None
            nbytes = _lib.lib.xsq_cdae_workspace(h, B, S)
            if nbytes == 0:
                raise _lib.XsqError(f"xsq_cdae_workspace(B={B}, S={S}) failed: need at least 3 slices")
            ws = self._workspace(dev, nbytes)
            _lib.check(_lib.lib.xsq_cdae_forward_xin(
                h, X.data_ptr(), B, S, None if masked else Y.data_ptr(), masks.data_ptr() if masks is not None else None,
                ws.data_ptr(), ws.numel(), _lib.stream_ptr(), int(bool(xin_ready))), "xsq_cdae_forward")
            if masked:
                wiener_em_masked_arena(self.table, X, masks, Y, B, S, batch_group=wiener_batch_group)
            elif not phasemix:
                wiener_em_arena(self.table, X, Y, B, S, batch_group=wiener_batch_group)
        Ylist = self.table.views(Y, (4, B, 2), S)
        if return_masks:
            return Ylist, self.table.views(masks, (4, B, 2), S, complex_=False)
        return Ylist
