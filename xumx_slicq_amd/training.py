"""Mirror of the training half of /root/reference/xumx_slicq_v2/training.py:34-112 (``loop`` with
``train=True``) on the HIP library: one call = forward with BatchNorm on batch statistics, ComplexMSE +
MaskSum loss, backward of every trainable tensor, AdamW update (training.py:391-393 defaults).

The reference hands the step to torch autograd + ``torch.optim.AdamW``; here the whole step is
``xsq_train_step`` (csrc/train.hip) over flat parameter / gradient / moment pools that keep the
reference's state_dict order, so a checkpoint of the reference loads, trains and saves unchanged.
The SDR term (auraloss, off by default: ``--sdr-mcoef -1``), the LSTM variant, the LR scheduler /
early stopping / tensorboard logging around the loop are outside the hot path."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .model import Unmix


class PendingLoss:
    """Loss of a step issued with ``wait=False``: ``result()`` (or indexing / iteration, like the tuple ``step``
    returns) blocks until THAT step has finished on the device.  Valid for the trainer's last four steps."""

    def __init__(self, trainer: "Trainer", ticket: int):
        self._trainer, self._ticket, self._value = trainer, int(ticket), None

    def result(self) -> Tuple[float, float, float]:
        if self._value is None:
            out = (C.c_double * 2)()
            _lib.check(_lib.lib.xsq_train_loss(self._trainer._h, self._ticket, out), "xsq_train_loss")
            mse, mask = float(out[0]), float(out[1])
            self._value = (mse + mask, mse, mask)
        return self._value

    def __getitem__(self, i):
        return self.result()[i]

    def __iter__(self):
        return iter(self.result())


class Trainer:
    """Owns the device-side training state of one ``Unmix``.

    ``step(x, y_targets)`` = the body of ``for x, y in pbar`` (training.py:66-110) and returns
    ``(loss, mse_loss, mask_loss)`` as python floats; ``gradients()`` exposes what
    ``loss.backward()`` would have left in ``.grad``; ``state_dict()`` / ``sync_to(unmix)`` hand
    the trained tensors back in the reference's layout."""

    def __init__(self, unmix: Unmix, encoder, lr: float = 1e-3, weight_decay: float = 1e-5,
                 device: str | torch.device = "cuda", precision: str = "fp32"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.XsqError("training runs on a ROCm device only; there is no CPU fallback")
        self.nsgt, self.insgt, self.cnorm = encoder
        self.lr, self.weight_decay = float(lr), float(weight_decay)
        self.table = unmix.table
        self._F, self._T = unmix._F, unmix._T
        causal = {blk.causal for blk in unmix.sliced_umx}
        wiener = {not blk.realtime for blk in unmix.sliced_umx}
        if len(causal) != 1 or len(wiener) != 1:
            raise _lib.XsqError("all blocks must share the same first-layer type and post-filter")
        self.causal, self.wiener = causal.pop(), wiener.pop()
        self._spec = [(k, tuple(v.shape)) for k, v in unmix.state_dict().items() if not k.endswith("num_batches_tracked")]
        params = unmix.packed_parameters()
        self.nparams = int(params.size)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.xsq_train_create(C.byref(self._h), len(self.table), self._F.ctypes.data,
                                                 self._T.ctypes.data, 1 if self.causal else 0,
                                                 params.ctypes.data, params.size), "xsq_train_create")
        if precision not in ("fp32", "bf16x6", "bf16"):
            raise ValueError(f"precision {precision!r}: the training step offers 'fp32', 'bf16x6' and 'bf16'")
        self.precision = precision
        from .model import note_precision
        note_precision(self, precision)          # bf16x6: no packed-fp32 slice FFT in this process meanwhile
        # "bf16": the reference's autocast arithmetic for the convolutions (training.py:473-476): operands rounded to bf16, fp32 accumulate
        _lib.check(_lib.lib.xsq_train_set_precision(self._h, {"fp32": 0, "bf16": 1, "bf16x6": 2}[precision]), "xsq_train_set_precision")
        self._ws = None
        self.steps = 0

    def __del__(self):
        try:
            from .model import note_precision
            note_precision(self, "fp32")
            if self._h:
                _lib.lib.xsq_train_destroy(self._h)
        except Exception:
            pass

    # -- one step ------------------------------------------------------------------------
    def step_arena(self, X: Tensor, Yt: Tensor, B: int, S: int, apply_update: bool = True, wait: bool = True):
        """X: mix arena (2B channels), Yt: target arena (8B channels), both flat fp32 on the device.
        ``wait=False``: returns a ``PendingLoss`` without waiting for the device (the next step can be issued before
        this one's loss is looked at; the reference's ``loss.item()`` every step, training.py:110, is ``wait=True``)."""
        with torch.cuda.device(self.device):
            need = self.__dict__.setdefault("_need", {}).get((B, S))
            if need is None:
                need = _lib.lib.xsq_train_workspace(self._h, B, S, 1 if self.wiener else 0)
                if need == 0:
                    raise _lib.XsqError("xsq_train_workspace: bad shape")
                self._need[(B, S)] = need
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            out = (C.c_double * 2)() if wait else None
            _lib.check(_lib.lib.xsq_train_step(self._h, X.data_ptr(), Yt.data_ptr(), B, S, 1 if self.wiener else 0,
                                               self.lr, self.weight_decay, 1 if apply_update else 0, out,
                                               self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()),
                       "xsq_train_step")
            if not wait:     # the arenas must outlive the kernels that read them: hand them to the pending object
                pending = PendingLoss(self, int(_lib.lib.xsq_train_ticket(self._h)))
                pending._keep = (X, Yt)
        if apply_update:
            self.steps += 1
        if not wait:
            return pending
        mse, mask = float(out[0]), float(out[1])
        return mse + mask, mse, mask

    @torch.no_grad()
    def step(self, x: Tensor, y_targets: Tensor, apply_update: bool = True, wait: bool = True):
        """x (B, 2, N) mix, y_targets (4, B, 2, N): training.py:66-108.  Returns (loss, mse, mask) -- or, with
        ``wait=False``, a ``PendingLoss`` that resolves to the same tuple."""
        x = x.to(self.device, torch.float32)
        y_targets = y_targets.to(self.device, torch.float32)
        # the two transforms are independent: the targets' (4x the rows) goes to a side stream beside the mix's
        main = torch.cuda.current_stream(self.device)
        side = self.__dict__.setdefault("_side", None) or torch.cuda.Stream(device=self.device)
        self._side = side
        # (straight to the arenas: the module API's list of 70 block views per transform -- built, checked back into an arena
        #  and record_stream'ed one by one -- cost ~0.3 ms of host time per step during which the device sat idle,
        #  profiles/r07h_timeline_train.txt)
        eng = self.nsgt.nsgt.nsgt
        side.wait_stream(main)
        with torch.cuda.stream(side):
            Ytg, lead_t, S_t = eng.forward(y_targets)
        X, lead, S = eng.forward(x)
        main.wait_stream(side)
        Ytg.record_stream(main)
        if len(lead) != 2 or lead[1] != 2 or lead_t != (4, lead[0], 2) or S != S_t:
            raise ValueError(f"expected x (B, 2, N) and y_targets (4, B, 2, N); got arenas {lead} / {lead_t}")
        return self.step_arena(X, Ytg, lead[0], S, apply_update, wait)

    # -- state ---------------------------------------------------------------------------
    def _read(self, what: int) -> "OrderedDict[str, Tensor]":
        buf = np.empty(self.nparams, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.xsq_train_read(self._h, what, buf.ctypes.data), "xsq_train_read")
        out, o = OrderedDict(), 0
        for k, shp in self._spec:
            n = int(np.prod(shp)) if shp else 1
            out[k] = torch.from_numpy(buf[o:o + n].reshape(shp).copy())
            o += n
        return out

    def state_dict(self) -> "OrderedDict[str, Tensor]":
        """Parameters and BatchNorm running statistics, reference key layout (num_batches_tracked left out)."""
        return self._read(0)

    def gradients(self) -> Dict[str, Tensor]:
        """Gradients of the last step for every trainable tensor (the running statistics have none)."""
        return OrderedDict((k, v) for k, v in self._read(1).items()
                           if not k.endswith(("running_mean", "running_var")))

    def _write(self, what: int, tensors: Dict[str, Tensor]):
        buf = np.empty(self.nparams, dtype=np.float32)
        o = 0
        for k, shp in self._spec:
            n = int(np.prod(shp)) if shp else 1
            v = tensors[k]
            if tuple(v.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(v.shape)} does not match {tuple(shp)}")
            buf[o:o + n] = v.detach().to("cpu", torch.float32).reshape(-1).numpy()
            o += n
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.xsq_train_write(self._h, what, buf.ctypes.data), "xsq_train_write")

    def _trainable(self):
        """Keys of ``unmix.parameters()`` in the order torch.optim.AdamW numbers them (training.py:391-393): the
        state_dict order without the BatchNorm running statistics (buffers)."""
        return [k for k, _ in self._spec if not k.endswith(("running_mean", "running_var"))]

    def optimizer_state_dict(self) -> dict:
        """``torch.optim.AdamW.state_dict()`` as the reference checkpoints it (training.py:419-430, :526): ``state``
        {parameter index: {step, exp_avg, exp_avg_sq}} over ``unmix.parameters()`` and ONE entry in ``param_groups``
        (lr, betas, eps, weight_decay, params) -- a reference ``.chkpnt``'s ``checkpoint["optimizer"]`` loads here and
        what is saved here loads into the reference's optimizer.  Before the first step ``state`` is empty, as torch's."""
        step = int(_lib.lib.xsq_train_step_count(self._h, -1))
        if step < 0:
            raise _lib.XsqError(f"xsq_train_step_count failed ({step}): {_lib.last_error()}")
        keys = self._trainable()
        state = {}
        if step > 0:
            m, v = self._read(2), self._read(3)
            state = {i: {"step": torch.tensor(float(step)), "exp_avg": m[k], "exp_avg_sq": v[k]} for i, k in enumerate(keys)}
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(keys)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, state: dict):
        """Resume from ``optimizer.state_dict()`` (torch AdamW layout, see above): restores both moments and the step
        counter, so bias correction continues where it stopped.  One parameter group, one common step count (what a
        single AdamW over ``unmix.parameters()`` produces); anything else is rejected."""
        groups = state["param_groups"]
        keys = self._trainable()
        if len(groups) != 1 or list(groups[0]["params"]) != list(range(len(keys))):
            raise ValueError(f"expected one param group over {len(keys)} parameters (torch.optim.AdamW(unmix.parameters()))")
        g = groups[0]
        if tuple(g.get("betas", (0.9, 0.999))) != (0.9, 0.999) or float(g.get("eps", 1e-8)) != 1e-8 or g.get("amsgrad", False):
            raise ValueError("the step implements AdamW with betas (0.9, 0.999), eps 1e-8, amsgrad off (training.py:391-393)")
        st = state["state"]
        shapes = dict(self._spec)
        if st:
            if sorted(st) != list(range(len(keys))):
                raise ValueError("optimizer state must cover every parameter (or none)")
            steps = {int(float(st[i]["step"])) for i in st}
            if len(steps) != 1:
                raise ValueError(f"parameters disagree on the step count: {sorted(steps)[:4]}")
            step = steps.pop()
        else:
            step = 0
        for what, name in ((2, "exp_avg"), (3, "exp_avg_sq")):
            full = {k: torch.zeros(shapes[k]) for k, _ in self._spec}          # running statistics carry no moments
            for i, k in enumerate(keys):
                if st:
                    full[k] = st[i][name]
            self._write(what, full)
        rc = int(_lib.lib.xsq_train_step_count(self._h, step))
        if rc < 0:
            raise _lib.XsqError(f"xsq_train_step_count failed ({rc}): {_lib.last_error()}")
        self.steps = step
        self._synced = self.steps
        self.lr = float(g.get("lr", self.lr))
        self.weight_decay = float(g.get("weight_decay", self.weight_decay))

    def load_state_dict(self, sd: Dict[str, Tensor]):
        """Parameters + running statistics from a (reference-layout) state_dict."""
        self._write(0, sd)

    def sync_to(self, unmix: Unmix) -> Unmix:
        """Hands the trained tensors to ``unmix``; ``num_batches_tracked`` advances by the steps taken SINCE
        THE LAST sync (calling this once per epoch must not count earlier epochs again)."""
        sd = unmix.state_dict()
        for k, v in self.state_dict().items():
            sd[k] = v
        fresh = self.steps - getattr(self, "_synced", 0)
        for k in sd:
            if k.endswith("num_batches_tracked"):
                sd[k] = sd[k] + fresh
        unmix.load_state_dict(sd, strict=True)
        self._synced = self.steps
        return unmix
