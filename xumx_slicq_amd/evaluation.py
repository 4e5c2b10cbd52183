"""Mirror of /root/reference/xumx_slicq_v2/evaluation.py:16-44 (``separate_and_evaluate``): track audio ->
``preprocess_audio`` -> ``separator(audio)`` -> ``Separator.to_dict`` -> per-target scores.

The reference scores with ``museval.eval_mus_track`` on a ``musdb.MultiTrack``; neither package (nor MUSDB18-HQ,
nor trained weights) exists offline.  This mirror takes any track object with the three attributes the reference
reads (``audio`` (T, C) array, ``rate``, ``targets`` {name: object with ``.audio`` (T, C)}), runs the same
separation call, and scores with museval when it is importable; otherwise it reports the global
signal-to-distortion ratio 10 log10(|s|^2 / |s - s_hat|^2) per target (the "new SDR" of the Music Demixing
Challenge -- NOT BSSEval v4; the result says which one it is).  The separation half is the accelerated hot path;
the scoring half is CPU bookkeeping.
"""
from __future__ import annotations

from typing import Dict, Union

import numpy as np
import torch

from .audio import preprocess_audio
from .separator import Separator


def global_sdr(reference: np.ndarray, estimate: np.ndarray, eps: float = 1e-10) -> float:
    """10 log10(sum s^2 / sum (s - s_hat)^2) over all samples and channels of one target."""
    ref = np.asarray(reference, dtype=np.float64)
    est = np.asarray(estimate, dtype=np.float64)
    n = min(ref.shape[0], est.shape[0])
    ref, est = ref[:n], est[:n]
    return float(10.0 * np.log10((np.sum(ref * ref) + eps) / (np.sum((ref - est) ** 2) + eps)))


def separate_and_evaluate(separator: Separator, track, device: Union[str, torch.device] = "cuda") -> Dict:
    """evaluation.py:16-44.  Returns {"metric": "museval-bsseval-v4" | "global-sdr", "scores": ...,
    "estimates": {target: (T, C) float32 array}}."""
    audio = torch.as_tensor(np.asarray(track.audio), dtype=torch.float32, device=device)
    audio = preprocess_audio(audio, track.rate, float(separator.sample_rate))
    estimates = separator.to_dict(separator(audio))
    estimates = {k: v[0].detach().cpu().numpy().T for k, v in estimates.items()}      # (T, C), as museval wants them
    try:
        import museval
    except ImportError:                 # only the import: an error INSIDE museval must not turn into the other metric
        museval = None
    if museval is not None:
        return {"metric": "museval-bsseval-v4", "scores": museval.eval_mus_track(track, estimates), "estimates": estimates}
    scores = {name: global_sdr(np.asarray(track.targets[name].audio), est)
              for name, est in estimates.items() if name in getattr(track, "targets", {})}
    return {"metric": "global-sdr", "scores": scores, "estimates": estimates}
