"""Drop-in mirror of /root/reference/xumx_slicq_v2/separator.py (Separator,
load_target_models) for the ROCm backend.

Same class/function names, arguments, attribute names, output layout
(targets first: (4, nb_samples, 2, N)) and exceptions as
the reference's torch path.  The ONNX runtimes and the GitHub download are
outside the accelerated path.
"""
from __future__ import annotations

import os

import json
import sys
from pathlib import Path
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from .model import Unmix
from .transforms import ComplexNorm, NSGTBase, make_filterbanks
from .weights import seeded_state_dict

# the reference's list (separator.py:29) plus the backend this package provides
_SUPPORTED_RUNTIMES = ["torch-cpu", "torch-cuda", "onnx-cpu", "onnx-cuda", "hip-rocm"]
_ACCELERATED = "hip-rocm"


class Separator(nn.Module):
    sources = ["bass", "vocals", "other", "drums"]      # separator.py:48
    accepts_item_lists = True                           # demix_into takes a list of per-item tensors (sharding.ShardedDemixer)

    @classmethod
    def load(cls, chunk_size: int = 2621440, model_path: Optional[str] = None,
             runtime_backend: Optional[str] = _ACCELERATED, warmup: int = 0, realtime: bool = False,
             device: Union[str, torch.device] = "cuda"):
        """separator.py:50-93."""
        if runtime_backend not in _SUPPORTED_RUNTIMES:
            raise ValueError(f"requested runtime backend {runtime_backend} not in {_SUPPORTED_RUNTIMES}")
        xumx_model, encoder, sample_rate = load_target_models(
            model_path, runtime_backend=runtime_backend, realtime=realtime, device=device)
        separator = cls(xumx_model=xumx_model, encoder=encoder, sample_rate=sample_rate,
                        runtime_backend=runtime_backend, chunk_size=chunk_size, device=device).to(device)
        separator.freeze()
        for _ in range(warmup):
            waveform = torch.rand((1, 2, int(100 * sample_rate)), dtype=torch.float32, device=device)
            separator.forward(waveform)
        return separator

    def __init__(self, xumx_model: Unmix = None, encoder: Tuple = None, runtime_backend: str = _ACCELERATED,
                 sample_rate: float = 44100.0, chunk_size: Optional[int] = 2621440, device: str = "cuda",
                 quiet: bool = False):
        super().__init__()
        if runtime_backend != _ACCELERATED:
            raise ValueError(f"this package provides the '{_ACCELERATED}' backend only (got {runtime_backend}); "
                             "use the reference for torch-*/onnx-*")
        self.device = device
        self.nb_channels = 2
        self.register_buffer("sample_rate", torch.as_tensor(sample_rate))
        self.chunk_size = chunk_size if chunk_size is not None else sys.maxsize
        self.xumx_model = xumx_model
        self.runtime_backend = runtime_backend
        self.nsgt, self.insgt, self.cnorm = encoder
        self.quiet = quiet
        self.sources = Separator.sources

    def freeze(self):
        for p in self.parameters():
            p.grad = None
        self.xumx_model.freeze()
        self.eval()

    def _graph_key(self, audio_big: Tensor):
        """Everything a captured forward depends on besides the input values: shape and device, the chunking
        switches, the packed-parameter version of the model (captured kernels hold raw pointers into the
        xsq_model handle), the arithmetic mode and the post-filter / fusion switches read per call."""
        m = self.xumx_model
        return (tuple(audio_big.shape), audio_big.device.index, self.chunk_size, getattr(m, "precision", "fp32"),
                bool(getattr(self, "overlap_tail", True)), bool(getattr(self, "batch_chunks", True)),
                int(getattr(self, "max_stack", 8)), int(getattr(self, "pass_streams", 1)),
                m._version(), tuple(bool(b.realtime) for b in m.sliced_umx), self._fused(),
                bool(getattr(self, "fuse_whiten", os.environ.get("XSQ_FUSE_WHITEN", "1") != "0")),
                bool(getattr(m, "wiener_masked", os.environ.get("XSQ_WIENER_MASKED", "1") != "0")), self._packed_fft(),
                bool(getattr(self, "native", os.environ.get("XSQ_NATIVE_FORWARD", "1") != "0")), int(getattr(self, "max_item_slices", 0)),
                int(getattr(m, "winograd", 7)))

    def drop_graphs(self):
        """Forget every captured forward (they hold raw pointers into the model handle and the workspaces)."""
        self.__dict__.pop("_graphs", None)

    def forward_graphed(self, audio_big: Tensor) -> Tensor:
        """``forward`` replayed from a HIP graph captured per input shape (launch-bound shapes: the
        2.2 s tail chunk or streaming-sized inputs spend more time between launches than inside them).
        The result tensor is owned by the graph and overwritten by the next call of the same shape.
        Graphs captured against an older parameter version / post-filter setting are dropped before the
        model handle they point into is rebuilt (``Unmix._model`` frees it)."""
        key = self._graph_key(audio_big)
        cache = self.__dict__.setdefault("_graphs", {})
        ver = key[8:]
        for k in [k for k in cache if k[1] == key[1] and k[8:] != ver]:
            del cache[k]                              # stale: would replay against freed device memory
        entry = cache.get(key)
        wiener = self._native_mode(audio_big)
        if wiener is not None and (audio_big.dtype != torch.float32 or not audio_big.is_contiguous()):
            audio_big = audio_big.contiguous().float()
        if entry is None:
            # native path: the captured kernels read the input through a DEVICE pointer slot (xsq_separator_forward_indirect),
            # so a replay follows the caller's tensor -- no copy of the input into a static buffer (85 MB for a 240 s track).
            # Module-API schedule (an A/B switch off its default): the captured launches hold the input's address, so the
            # input is copied into a static tensor as before.
            slot = torch.zeros(1, dtype=torch.int64, device=audio_big.device) if wiener is not None else None
            static_in = None if slot is not None else audio_big.clone()
            if slot is not None:
                slot.fill_(audio_big.data_ptr())
            call = (lambda: self._forward_native(audio_big, wiener, slot=slot)) if slot is not None else (lambda: self.forward(static_in))
            side = torch.cuda.Stream(device=audio_big.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up on a side stream: fills the shape-keyed caches
                for _ in range(2):
                    call()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = call()
            # the captured kernels hold raw pointers into the grow-only workspaces: keep those tensors
            # alive with the graph even if a later, larger call replaces them in their owners
            from . import phase
            keep = (list(self.insgt.nsgt.nsgt._ws.values()) + list(self.nsgt.nsgt.nsgt._ws.values())
                    + list(self.xumx_model._ws.values()) + list(phase._WS.values())
                    + list(self.__dict__.get("_nws", {}).values()))
            entry = cache[key] = [graph, static_in, static_out, keep, slot, audio_big.data_ptr()]
        graph, static_in, static_out, _keep, slot, last_ptr = entry
        if slot is None:
            static_in.copy_(audio_big)
        elif audio_big.data_ptr() != last_ptr:
            slot.fill_(audio_big.data_ptr())       # 8 bytes, one small launch on the replay's stream, in front of the replay
            entry[5] = audio_big.data_ptr()
        graph.replay()
        return static_out

    def _packed_fft(self) -> bool:
        """Packed-fp32 butterflies in the slice FFTs: a DIAGNOSTIC build option (csrc/Makefile PACKED_FFT=1), not part of
        the product library -- measured, they change nothing (the transforms are bound by their HBM phases, DESIGN.md
        section 4) and next to v_mfma_f32_16x16x32_bf16 waves of another stream the packed transform returned wrong values
        (tools/probe/pk_mfma_hazard.hip).  ``packed_fft = True`` / XSQ_PACKED_FFT=1 asks the library for them; the product
        build refuses with ``XsqError``.  A diagnostic build honours it only while no model or trainer of this process
        contracts on split-bf16 MFMAs."""
        want = bool(getattr(self, "packed_fft", os.environ.get("XSQ_PACKED_FFT", "0") != "0"))
        eng = self.nsgt.nsgt.nsgt
        if not want and not eng._packed_fft:
            return False
        from .model import split_bf16_active
        on = want and not split_bf16_active()
        eng.set_packed_fft(on)
        self.insgt.nsgt.nsgt.set_packed_fft(on)
        return on

    def _fused(self) -> bool:
        """Mix-phase models: the CDAE writes the masks only and the inverse transform forms mask * X while it
        loads (same products, same order: bitwise equal to decoding the materialised estimates)."""
        return bool(getattr(self, "fuse_phasemix", os.environ.get("XSQ_FUSE_PHASEMIX", "1") != "0")
                    and all(bool(b.realtime) for b in self.xumx_model.sliced_umx))

    def _encode(self, audio: Tensor):
        """Forward transform of (B, 2, n) audio -> (coefficient list, xin_ready).  By default the analysis kernels also
        write the CDAE's whitened magnitude straight into the model's workspace (xsq_slicqt_forward_xin; same values as
        the separate magnitude pass, which is then skipped); ``fuse_whiten = False`` / XSQ_FUSE_WHITEN=0 restores it."""
        if not getattr(self, "fuse_whiten", os.environ.get("XSQ_FUSE_WHITEN", "1") != "0") or audio.dim() != 3 or audio.shape[1] != 2 \
                or audio.device.type != "cuda":
            return self.nsgt(audio), False
        eng = self.nsgt.nsgt.nsgt
        S = eng.plan.num_slices(audio.shape[-1])
        ws, mean, scale, split = self.xumx_model.whitening_target(audio.device, audio.shape[0], S)
        arena, lead, S = eng.forward(audio, whiten=(ws.data_ptr(), mean, scale, split))
        return eng.table.views(arena, lead, S), True

    def _decode_into(self, out: Tensor, Xc, length: int, offsets: Tensor, group: int = 0, xin_ready: bool = False):
        """CDAE + post-filter + inverse transform of the coefficient list ``Xc`` (batch B); packed channel
        (target, b, c) is written to out.view(-1)[offsets[target, b, c] : +length]."""
        eng = self.insgt.nsgt.nsgt
        offs = offsets.reshape(-1).contiguous()
        if self._fused():
            masks, X, B, S = self.xumx_model.masks_arena(Xc, xin_ready=xin_ready)
            eng.backward_masked(masks, X, 8 * B, 2 * B, S, length, out, offs)
            return
        Ylist = self.xumx_model(Xc, wiener_batch_group=group, xin_ready=xin_ready)
        arena, lead, S = eng.table.as_arena(list(Ylist))
        eng.backward(arena, offs.numel(), S, length, out=out, row_offsets=offs)

    def _input_rows(self, items):
        """Input row table (device int64, element offsets from items[0]) of a list of equally shaped (nb_samples, 2, n) fp32
        tensors on one device, or None when the list has to be packed: xsq_demix_pass then reads every work item where it
        lies -- chunk views of different tracks -- instead of through a torch.cat copy (84 MB per stacked pass of four
        full chunks).  Cached by the tensors' addresses and strides."""
        t0 = items[0]
        if len(items) < 2 or self._native_mode(t0) is None:
            return None
        for t in items:
            if (t.dim() != 3 or t.shape != t0.shape or t.dtype != torch.float32 or t.device != t0.device or t.stride(-1) != 1
                    or (t.data_ptr() - t0.data_ptr()) % 4):
                return None
        # the table has 2 * nb rows per item: nb (and the device) belong to the key, not only the addresses and strides
        key = (t0.device.index, int(t0.shape[0])) + tuple((t.data_ptr(), t.stride(0), t.stride(1)) for t in items)
        cache = self.__dict__.setdefault("_xrows", {})
        hit = cache.get(key)
        if hit is None:
            base = t0.data_ptr()
            rows = [(t.data_ptr() - base) // 4 + b * t.stride(0) + c * t.stride(1)
                    for t in items for b in range(t.shape[0]) for c in range(2)]
            if len(cache) > 4096:
                # kernels in flight on either stream may still read an evicted table (the allocator orders reuse on the
                # allocating stream only): wait for the device before the tables go.  Once per 4096 new item lists.
                torch.cuda.synchronize(t0.device)
                cache.clear()
            hit = cache[key] = torch.tensor(rows, dtype=torch.int64, device=t0.device)
        return hit

    @torch.no_grad()
    def demix_into(self, audio, out: Tensor, row_offsets: Tensor, group: int = 1):
        """One pass over ``audio`` (k * group, 2, n): k independent work items of n <= chunk_size samples each
        (chunks of any tracks -- no state crosses the reference's chunk loop, separator.py:153-229), ``group``
        = nb_samples of an item; or a LIST of k tensors (group, 2, n), read in place.  The stems of packed channel
        (target, item*group + b, c) land at out.view(-1)[row_offsets[target, item*group + b, c] : + n] -- the hard concat
        of separator.py:231 by placement.  The Wiener window maximum keeps its per-item scope (quirk A13) through
        ``group``.  Building block of ``sharding.ShardedDemixer``; ``forward`` is the single-track case."""
        x_rows, keep = None, None
        if isinstance(audio, (list, tuple)):
            keep = list(audio)
            x_rows = self._input_rows(keep)
            if x_rows is None:
                audio = keep[0] if len(keep) == 1 else torch.cat(keep, dim=0)
            else:
                audio = keep[0]
        if audio.dim() != 3 or audio.shape[1] != 2:
            raise ValueError(f"audio must be (items * nb_samples, 2, n); got {tuple(audio.shape)}")
        B = audio.shape[0] * (len(keep) if x_rows is not None else 1)
        n = audio.shape[-1]
        self._packed_fft()
        if n > self.chunk_size:
            raise ValueError(f"work items are at most chunk_size = {self.chunk_size} samples (got {n})")
        if tuple(row_offsets.shape) != (4, B, 2) or row_offsets.dtype != torch.int64:
            raise ValueError(f"row_offsets must be int64 (4, {B}, 2); got {tuple(row_offsets.shape)}")
        min_samples = int(self.nsgt.nsgt.sllen / 2) + 1
        wiener = self._native_mode(audio)
        if wiener is not None and out.is_contiguous() and out.dtype == torch.float32 and B % max(1, group) == 0:
            # one C call per pass (xsq_demix_pass): the zero padding of a short item is a slice count, not a copy
            from . import _lib
            if x_rows is None and (audio.dtype != torch.float32 or not audio.is_contiguous()):
                audio = audio.contiguous().float()
            dev = audio.device
            offs = row_offsets if row_offsets.is_contiguous() else row_offsets.contiguous()
            with torch.cuda.device(dev):
                model = self.xumx_model._model(dev)
                d = self.nsgt.nsgt.nsgt.demixer(dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                n_pad = max(n, min_samples)
                key = ("pass", B, n_pad, wiener, self.xumx_model._version(), dev.index)
                nbytes = self.__dict__.setdefault("_nsizes", {}).get(key)
                if nbytes is None:
                    nbytes = _lib.lib.xsq_demix_pass_workspace(d, model, B, n_pad, wiener)
                    if nbytes == 0:
                        raise _lib.XsqError("xsq_demix_pass_workspace: " + _lib.last_error())
                    self._nsizes[key] = nbytes
                ws = self._native_ws(dev, stream, "pass", nbytes)
                _lib.check(_lib.lib.xsq_demix_pass(d, model, audio.data_ptr(), x_rows.data_ptr() if x_rows is not None else None,
                                                   B, n, n_pad, int(group), wiener, out.data_ptr(), offs.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), stream), "xsq_demix_pass")
            return
        if x_rows is not None:
            audio = torch.cat(keep, dim=0)
        if n < min_samples:
            audio = torch.cat([audio, torch.zeros((*audio.shape[:-1], min_samples - n), device=audio.device,
                                                  dtype=audio.dtype)], dim=-1)
        Xc, ready = self._encode(audio)
        self._decode_into(out, Xc, n, row_offsets, group=group, xin_ready=ready)

    # -- the native whole-call path ---------------------------------------------------------------------------
    def _native_mode(self, audio: Tensor):
        """None when this call has to take the Python chunk loop (an A/B switch is off its default, a mixed or
        non-stereo model), else the post-filter of the native call: 0 mix-phase, 1 Wiener-EM."""
        if not getattr(self, "native", os.environ.get("XSQ_NATIVE_FORWARD", "1") != "0"):
            return None
        if audio.dim() != 3 or audio.shape[1] != 2 or audio.device.type != "cuda" or audio.shape[-1] < 1:
            return None
        if not (getattr(self, "batch_chunks", True) and int(getattr(self, "pass_streams", 1)) == 1
                and getattr(self, "fuse_whiten", os.environ.get("XSQ_FUSE_WHITEN", "1") != "0")
                and getattr(self, "fuse_phasemix", os.environ.get("XSQ_FUSE_PHASEMIX", "1") != "0")
                and getattr(self.xumx_model, "wiener_masked", os.environ.get("XSQ_WIENER_MASKED", "1") != "0")):
            return None
        modes = {bool(b.realtime) for b in self.xumx_model.sliced_umx}
        if len(modes) != 1:
            return None
        return 0 if modes.pop() else 1

    def _native_ws(self, dev: torch.device, stream_ptr: int, which: str, nbytes: int) -> Tensor:
        """Grow-only workspace of the native path per (device, stream, main | tail).  PyTorch owns the memory."""
        key = (dev.index, stream_ptr, which)
        pool = self.__dict__.setdefault("_nws", {})
        ws = pool.get(key)
        if ws is None or ws.numel() < nbytes:
            pool[key] = None
            ws = pool[key] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        return ws

    def _tail_stream(self, dev: torch.device):
        pool = self.__dict__.setdefault("_side_streams", {}).setdefault(dev.index, [])
        if not pool:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[0]

    def _forward_native(self, audio_big: Tensor, wiener: int, slot: Optional[Tensor] = None) -> Tensor:
        """``forward`` as ONE C call (xsq_separator_forward, csrc/demix.hip): the stacked full chunks on the caller's
        stream, the tail chunk beside them on a side stream, the input read and the stems written in place through
        row-offset tables cached per call shape.  Per call the host allocates the result and makes one ctypes call."""
        from . import _lib
        import ctypes as C
        if audio_big.dtype != torch.float32 or not audio_big.is_contiguous():
            audio_big = audio_big.contiguous().float()
        nb, N, dev = audio_big.shape[0], audio_big.shape[-1], audio_big.device
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        eng = self.nsgt.nsgt.nsgt
        self._packed_fft()
        with torch.cuda.device(dev):
            model = self.xumx_model._model(dev)
            d = eng.demixer(dev)
            main = torch.cuda.current_stream(dev)
            overlap = bool(getattr(self, "overlap_tail", True))
            side = self._tail_stream(dev) if overlap else main
            cs = int(min(self.chunk_size, 1 << 62))
            max_stack = int(getattr(self, "max_stack", 8))
            key = (nb, N, cs, max_stack, wiener, self.xumx_model._version(), dev.index, int(getattr(self, "max_item_slices", 0)))
            # the cap is sticky state of the native demixer and part of its plan key: set it on EVERY call, not only on a
            # cache miss here -- after 0 -> 20 -> 0 a cached shape would otherwise run under the stale cap
            _lib.check(_lib.lib.xsq_demixer_set_max_rows(d, int(getattr(self, "max_item_slices", 0))), "xsq_demixer_set_max_rows")
            sizes = self.__dict__.setdefault("_nsizes", {}).get(key)
            if sizes is None:
                mb, tb = C.c_size_t(), C.c_size_t()
                _lib.check(_lib.lib.xsq_separator_workspace(d, model, nb, N, cs, max_stack, wiener, C.byref(mb), C.byref(tb)),
                           "xsq_separator_workspace")
                sizes = self._nsizes[key] = (mb.value, tb.value)
            two = overlap and sizes[1] > 0
            ws = self._native_ws(dev, main.cuda_stream, "main", sizes[0] if two else max(sizes))
            wt = self._native_ws(dev, main.cuda_stream, "tail", sizes[1]) if two else None
            out = torch.empty(4, nb, 2, N, dtype=torch.float32, device=dev)
            fn, src = ((_lib.lib.xsq_separator_forward, audio_big.data_ptr()) if slot is None
                       else (_lib.lib.xsq_separator_forward_indirect, slot.data_ptr()))
            _lib.check(fn(d, model, src, nb, N, cs, max_stack, wiener, 1 if two else 0, out.data_ptr(),
                          ws.data_ptr(), ws.numel(), wt.data_ptr() if two else None, wt.numel() if two else 0,
                          main.cuda_stream, side.cuda_stream if two else main.cuda_stream), "xsq_separator_forward")
        return out

    @torch.no_grad()
    def forward(self, audio_big: Tensor) -> Tensor:
        """(nb_samples, 2, N) fp32 on a ROCm device -> (4, nb_samples, 2, N).  separator.py:133-232:
        chunks of chunk_size samples, short chunks zero-padded to sllen/2+1, hard concat.

        The reference walks the chunks one by one; no state crosses the loop, so here all FULL
        chunks of the call are stacked along the batch axis and run as one pass (the Wiener
        window maximum stays per chunk via ``wiener_batch_group``) -- same numbers, a fifth of the
        launches.  By default the whole call is issued by native code (``_forward_native``); the Python
        loop below is the same schedule spelled through the module API and serves the A/B switches
        (``native = False``, ``batch_chunks = False`` for the literal loop, ``fuse_* = False``, ...)."""
        wiener = self._native_mode(audio_big)
        if wiener is not None:
            return self._forward_native(audio_big, wiener)
        nb, N, cs = audio_big.shape[0], audio_big.shape[-1], self.chunk_size
        min_samples = int(self.nsgt.nsgt.sllen / 2) + 1
        dev = audio_big.device
        self._packed_fft()
        # every chunk's stems are written straight into their place of the result (the hard concat
        # of separator.py:231 without a copy): packed channel (target, [chunk,] b, c) -> out row
        out = torch.empty(4, nb, 2, N, dtype=torch.float32, device=dev)
        rows = (torch.arange(4, device=dev).view(4, 1, 1, 1) * nb + torch.arange(nb, device=dev).view(1, 1, nb, 1)) * 2 \
            + torch.arange(2, device=dev).view(1, 1, 1, 2)                       # (4, 1, nb, 2) row of `out`

        def decode(audio, length, offsets, group=0):
            Xc, ready = self._encode(audio)
            self._decode_into(out, Xc, length, offsets, group, xin_ready=ready)

        full = N // cs if getattr(self, "batch_chunks", True) else 0
        start0 = 0
        # stack at most `max_stack` (chunk, sample) pairs per pass: bounds the workspaces (~1.8 GB per
        # stacked full chunk) and keeps BC*S inside one launch for any track length
        per_pass = max(1, getattr(self, "max_stack", 8) // nb)
        stacked = []                                                                 # (first sample, chunks)
        while full - start0 // cs >= 2 and per_pass >= 2:
            k = min(per_pass, full - start0 // cs)
            stacked.append((start0, k))
            start0 += k * cs

        def rest(first):
            for start in range(first, N, cs):
                audio = audio_big[..., start:min(start + cs, N)]
                n_samples = audio.shape[-1]
                if n_samples < min_samples:
                    audio = torch.cat([audio, torch.zeros((*audio.shape[:-1], min_samples - n_samples),
                                                          device=dev, dtype=audio.dtype)], dim=-1)
                decode(audio, n_samples, rows * N + start)

        if not stacked:
            rest(0)
            return out
        # The remaining chunks (the short tail of the track) are launch-bound -- a dozen kernels of a few
        # hundred workgroups each -- and independent of the stacked passes: they go out FIRST, on a side stream
        # with its own workspaces, and fill in beside the big launches instead of running after them.  The
        # stacked passes themselves are dealt to `pass_streams` streams (default 1: all on the caller's).
        main = torch.cuda.current_stream(dev)
        tail_first = stacked[-1][0] + stacked[-1][1] * cs
        pool = self.__dict__.setdefault("_side_streams", {}).setdefault(dev.index, [])
        nstreams = max(1, min(int(getattr(self, "pass_streams", 1)), len(stacked)))
        overlap_tail = tail_first < N and getattr(self, "overlap_tail", True)
        while len(pool) < nstreams:                       # pool[0] = tail, pool[1:] = extra pass streams
            pool.append(torch.cuda.Stream(device=dev))
        used = []
        if overlap_tail:
            pool[0].wait_stream(main)
            used.append(pool[0])
            with torch.cuda.stream(pool[0]):
                rest(tail_first)
        for i, (s0, k) in enumerate(stacked):
            st = main if i % nstreams == 0 else pool[i % nstreams]
            if st is not main and st not in used:
                st.wait_stream(main)
                used.append(st)
            with torch.cuda.stream(st):
                a = audio_big[..., s0:s0 + k * cs].reshape(nb, 2, k, cs).permute(2, 0, 1, 3).reshape(k * nb, 2, cs)
                offs = rows * N + s0 + torch.arange(k, device=dev).view(1, k, 1, 1) * cs   # (4, k, nb, 2)
                decode(a, cs, offs, group=nb)                                    # batch = (chunk, b)
        for st in used:
            main.wait_stream(st)
        if not overlap_tail:
            rest(tail_first)
        return out

    @staticmethod
    def to_dict(estimates: Tensor, aggregate_dict: Optional[dict] = None) -> dict:
        """separator.py:234-259."""
        estimates_dict = {target: estimates[k] for k, target in enumerate(Separator.sources)}
        if aggregate_dict is not None:
            new_estimates = {}
            for key in aggregate_dict:
                new_estimates[key] = torch.tensor(0.0)
                for target in aggregate_dict[key]:
                    new_estimates[key] = new_estimates[key] + estimates_dict[target]
            estimates_dict = new_estimates
        return estimates_dict


def build_models(fscale: str = "bark", fbins: int = 262, fmin: float = 32.9, sample_rate: float = 44100.0,
                 seq_dur: float = 2.0, realtime: bool = False, device="cuda", state=None,
                 strict: bool = True):
    """Plan + filterbanks + Unmix from explicit config (what load_target_models does after
    reading the JSON, separator.py:321-356)."""
    nsgt_base = NSGTBase(fscale, fbins, fmin, fs=sample_rate, device=device)
    jagged_slicq, _ = nsgt_base.predict_input_size(1, 2, seq_dur)
    cnorm = ComplexNorm()
    nsgt, insgt = make_filterbanks(nsgt_base, sample_rate)
    xumx_model = Unmix(cnorm(jagged_slicq), realtime=realtime, weights_follow=state is not None and strict)
    if state is not None:
        # the reference loads with strict=False and silently drops mismatches (quirk A7); be strict
        xumx_model.load_state_dict(state, strict=strict)
    xumx_model.freeze()
    return xumx_model, (nsgt, insgt, cnorm), sample_rate


def load_target_models(model_path: str, runtime_backend: str = _ACCELERATED, realtime: bool = False,
                       device="cuda"):
    """separator.py:262-387 for the accelerated backend: <model_path>/xumx_slicq_v2.json
    (args.sample_rate, fscale, fbins, fmin, seq_dur, realtime) + xumx_slicq_v2.pth."""
    if runtime_backend != _ACCELERATED:
        raise ValueError(f"unsupported runtime backend: {runtime_backend}")
    if model_path is None:
        raise ValueError("model_path is required: there is no network to download the pretrained model from")
    model_path = Path(model_path).expanduser()
    json_path = Path(model_path, "xumx_slicq_v2.json")
    assert model_path.exists() and json_path.exists()
    with open(json_path, "r") as stream:
        results = json.load(stream)
    pth = Path(model_path, "xumx_slicq_v2.pth")
    if pth.stat().st_size < 4096:
        raise RuntimeError(f"{pth} is a Git-LFS pointer, not a checkpoint; fetch the real weights or use "
                           "seeded_separator() for synthetic ones")
    state = torch.load(pth, map_location="cpu")
    args = results["args"]
    return build_models(args["fscale"], args["fbins"], args["fmin"], args["sample_rate"], args["seq_dur"],
                        realtime=args["realtime"], device=device, state=state)


def seeded_separator(realtime: bool = False, wiener: Optional[bool] = None, seed: int = 1234,
                     device="cuda", chunk_size: int = 2621440, **cfg) -> Separator:
    """Separator with the seeded synthetic weights (no checkpoint exists offline).
    ``wiener`` overrides the post-filter: None follows the reference (offline -> Wiener-EM,
    realtime -> mix-phase); False on the offline stack is BASELINE config 2."""
    xumx_model, encoder, sr = build_models(realtime=realtime, device=device, **cfg)
    xumx_model.load_state_dict(seeded_state_dict(xumx_model.table.shapes, seed=seed), strict=True)
    if wiener is not None:
        for blk in xumx_model.sliced_umx:      # the flag the reference reads at model.py:264
            blk.realtime = not wiener
    sep = Separator(xumx_model=xumx_model, encoder=encoder, sample_rate=sr, chunk_size=chunk_size,
                    device=device, quiet=True).to(device)
    sep.freeze()
    return sep
