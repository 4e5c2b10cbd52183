"""Counterpart of /root/reference/xumx_slicq_v2/inference.py for the ROCm backend:
``separate`` (inference.py:14-33, same timing convention: wall time of the
``separator(audio)`` call only) and a small CLI (``python -m xumx_slicq_amd``)."""
from __future__ import annotations

import argparse
import time
from pathlib import Path

import torch

from . import audio as xaudio
from .separator import Separator, seeded_separator


def separate(audio, separator, rate=None, device=None):
    """inference.py:14-33: returns ({target: (nb_samples, 2, T)}, seconds)."""
    if rate is None:
        raise Exception("rate` must be provided.")
    if device:
        audio = audio.to(device)
    audio = xaudio.preprocess_audio(audio, rate, separator.sample_rate)
    torch.cuda.synchronize(audio.device)
    start_time = time.time()
    estimates = separator(audio)
    torch.cuda.synchronize(audio.device)
    time_delta = time.time() - start_time
    return separator.to_dict(estimates), time_delta


def inference_main(argv=None):
    p = argparse.ArgumentParser(description="xumx-sliCQ-V2 inference on MI355X (hip-rocm backend)")
    p.add_argument("--input-dir", type=str, default="/input")
    p.add_argument("--output-dir", type=str, default="/output")
    p.add_argument("--ext", type=str, default=".wav")
    p.add_argument("--model-path", type=str, default=None,
                   help="directory with xumx_slicq_v2.json/.pth; omit for seeded synthetic weights")
    p.add_argument("--realtime", action="store_true")
    p.add_argument("--warmup", type=int, default=0)
    p.add_argument("--device", type=str, default="cuda")
    args = p.parse_args(argv)
    if args.model_path:
        separator = Separator.load(model_path=args.model_path, runtime_backend="hip-rocm",
                                   warmup=args.warmup, realtime=args.realtime, device=args.device)
    else:
        separator = seeded_separator(realtime=args.realtime, device=args.device)
    out_dir = Path(args.output_dir)
    tot, n = 0.0, 0
    for wav in sorted(Path(args.input_dir).glob(f"*{args.ext}")):
        sig, rate = xaudio.load_audio(str(wav))
        estimates, dt = separate(sig, separator, rate=rate, device=args.device)
        tot, n = tot + dt, n + 1
        target_dir = out_dir / wav.stem
        target_dir.mkdir(parents=True, exist_ok=True)
        for target, est in estimates.items():
            xaudio.save_wav_float(str(target_dir / f"{target}.wav"), est[0], rate)
        print(f"{wav.name}: {sig.shape[-1] / rate:.1f} s demixed in {dt * 1e3:.1f} ms")
    if n:
        print(f"xumx-sliCQ-V2 inference time: {tot / n:.4f} s/track over {n} track(s)")


if __name__ == "__main__":
    inference_main()
