"""Audio front / back end of the reference CLI without torchaudio (absent offline):
``load_audio`` / ``load_info`` / ``preprocess_audio`` of
/root/reference/xumx_slicq_v2/data.py:35-156 and the float-PCM wav write of
inference.py:135-142, for RIFF/WAVE files (PCM 8/16/24/32-bit and IEEE float 32/64).
Host-side I/O: not part of the timed path (the reference times only ``separator(audio)``).
"""
from __future__ import annotations

import struct
import warnings
from typing import Optional, Tuple

import numpy as np
import torch


def _read_chunks(path):
    with open(path, "rb") as f:
        riff, _size, wave = struct.unpack("<4sI4s", f.read(12))
        if riff != b"RIFF" or wave != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        while True:
            head = f.read(8)
            if len(head) < 8:
                break
            cid, size = struct.unpack("<4sI", head)
            if cid == b"fmt ":
                fmt = f.read(size)
            elif cid == b"data":
                return fmt, f.tell(), size
            else:
                f.seek(size, 1)
            if size & 1:
                f.seek(1, 1)
    raise ValueError(f"{path}: no data chunk")


def _parse_fmt(fmt: bytes):
    tag, channels, rate, _br, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:            # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
        tag = struct.unpack("<H", fmt[24:26])[0]
    if tag not in (1, 3):
        raise ValueError(f"unsupported wav format tag {tag} (PCM and IEEE float only)")
    return tag, channels, rate, align, bits


def load_info(path: str) -> dict:
    """data.py:35-61."""
    fmt, _off, size = _read_chunks(path)
    _tag, channels, rate, align, _bits = _parse_fmt(fmt)
    samples = size // align
    return {"samplerate": rate, "samples": samples, "channels": channels, "duration": samples / rate}


def load_audio(path: str, start: float = 0.0, dur: Optional[float] = None,
               info: Optional[dict] = None) -> Tuple[torch.Tensor, int]:
    """data.py:64-95: (channels, samples) float32 in [-1, 1) and the sample rate."""
    fmt, off, size = _read_chunks(path)
    tag, channels, rate, align, bits = _parse_fmt(fmt)
    total = size // align
    first, count = 0, total
    if dur is not None:
        first = min(int(start * rate), total)
        count = min(int(dur * rate), total - first)
    with open(path, "rb") as f:
        f.seek(off + first * align)
        raw = f.read(count * align)
    if tag == 3:
        a = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif bits == 8:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif bits == 16:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif bits == 32:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported PCM width {bits}")
    return torch.from_numpy(a.reshape(-1, channels).T.copy()), rate


def load_audio_into(path: str, take) -> Tuple[torch.Tensor, int]:
    """``load_audio`` + the channel rules of ``preprocess_audio`` for the pipelined CLI: the file's samples are converted
    STRAIGHT into a caller-supplied (pinned) float32 buffer as (2, samples) -- one strided pass per channel, no interleaved
    float copy, no transpose copy, no second copy into the staging buffer.  ``take(numel)`` returns a 1-D float32 tensor of at
    least that many elements.  Returns ((2, samples) view of it, rate).  Same values as load_audio -> preprocess_audio."""
    fmt, off, size = _read_chunks(path)
    tag, channels, rate, align, bits = _parse_fmt(fmt)
    n = size // align
    with open(path, "rb") as f:
        f.seek(off)
        raw = f.read(n * align)
    if (tag == 1 and bits == 24) or channels > 2:      # 24-bit PCM, or the reference's shape rules for > 2 channels: the general decoder
        sig, rate = load_audio(path)
        out = take(2 * n)[:2 * n].view(2, n)
        out.copy_(preprocess_audio(sig)[0])
        return out, rate
    dt, scale, bias = {(3, 32): ("<f4", None, 0.0), (3, 64): ("<f8", None, 0.0), (1, 8): (np.uint8, 1.0 / 128.0, -128.0),
                       (1, 16): ("<i2", 1.0 / 32768.0, 0.0), (1, 32): ("<i4", 1.0 / 2147483648.0, 0.0)}[(tag, bits)]
    out = take(2 * n)[:2 * n].view(2, n)
    dst = out.numpy()
    frames = np.frombuffer(raw, dtype=dt).reshape(n, channels)
    # one converting, de-interleaving pass per channel (a strided read, a contiguous float32 write), then the scale in place --
    # in numpy: single-threaded inside the reader thread that called (a torch copy_ fans out over every core of the host and
    # measured 2.4x SLOWER on the 100-core GPU box than here)
    for c in range(2):
        dst[c] = frames[:, min(c, channels - 1)]             # mono is duplicated
    if bias:
        np.add(dst, np.float32(bias), out=dst)
    if scale is not None:
        np.multiply(dst, np.float32(scale), out=dst)
    return out, rate


def save_wav_float(path: str, audio: torch.Tensor, rate: int) -> None:
    """(channels, samples) -> IEEE float32 wav (torchaudio.save(..., encoding="PCM_F"), inference.py:135-142)."""
    a = audio.detach().to("cpu", torch.float32).numpy()
    if a.ndim == 1:
        a = a[None]
    channels, n = a.shape
    data = np.ascontiguousarray(a.T).astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, channels, int(rate), int(rate) * channels * 4, channels * 4, 32)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sI4s", b"RIFF", 4 + 8 + len(fmt) + 8 + len(data), b"WAVE"))
        f.write(struct.pack("<4sI", b"fmt ", len(fmt)) + fmt)
        f.write(struct.pack("<4sI", b"data", len(data)) + data)


def save_wav_float_interleaved(path: str, frames: torch.Tensor, rate: int) -> None:
    """(samples, channels) float32 ALREADY in the payload's order (the pipelined CLI interleaves on the GPU): header +
    one write of the tensor's memory, no host-side transpose or copy."""
    a = frames.detach()
    assert a.device.type == "cpu" and a.dtype == torch.float32 and a.dim() == 2 and a.is_contiguous()
    n, channels = a.shape
    nbytes = n * channels * 4
    fmt = struct.pack("<HHIIHH", 3, channels, int(rate), int(rate) * channels * 4, channels * 4, 32)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sI4s", b"RIFF", 4 + 8 + len(fmt) + 8 + nbytes, b"WAVE"))
        f.write(struct.pack("<4sI", b"fmt ", len(fmt)) + fmt)
        f.write(struct.pack("<4sI", b"data", nbytes))
        f.write(memoryview(a.numpy()).cast("B"))


def preprocess_audio(audio: torch.Tensor, rate: Optional[float] = None,
                     model_rate: Optional[float] = None) -> torch.Tensor:
    """data.py:98-156: any of (T,), (C,T), (T,C), (B,C,T) -> (nb_samples, 2, T).
    Mono is duplicated, more than two channels are cut to the first two (the reference
    means to warn there but never imports ``warnings``)."""
    if audio.dim() == 1:
        audio = audio[None, None, ...]
    elif audio.dim() == 2:
        audio = audio[None, ...] if min(audio.shape) <= 2 else audio[:, None, ...]
    if audio.shape[1] > audio.shape[2]:
        audio = audio.transpose(1, 2)
    if audio.shape[1] > 2:
        warnings.warn("Channel count > 2!. Only the first two channels will be processed!")
        audio = audio[:, :2, :]
    if audio.shape[1] == 1:
        audio = torch.repeat_interleave(audio, 2, dim=1)
    if rate is not None and model_rate is not None and float(rate) != float(model_rate):
        raise ValueError(f"input is {rate} Hz, the model {float(model_rate)} Hz: resample first "
                         "(the reference uses torchaudio's sinc resampler, not available offline)")
    return audio
