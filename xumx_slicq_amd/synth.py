"""Seeded synthetic stereo audio used by the bench, the smoke run and the tests.

There is no dataset and no network: every run uses the same rule
("Synthetic inputs"): NumPy PCG64 uniform noise at half scale
plus three fixed sinusoids with different phases per channel, fp32.
"""
from __future__ import annotations

import numpy as np
import torch

SAMPLE_RATE = 44100.0
INPUT_SEED = 20260101


def synth_audio(n: int, seed: int = INPUT_SEED, nb_samples: int = 1,
                nb_channels: int = 2) -> torch.Tensor:
    """(nb_samples, nb_channels, n) fp32 in roughly [-0.8, 0.8]."""
    rng = np.random.default_rng(seed)
    x = 0.5 * rng.uniform(-1.0, 1.0, (nb_samples, nb_channels, n))
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    for i, f in enumerate((110.0, 1760.0, 9000.0)):
        for c in range(nb_channels):
            x[:, c, :] += 0.1 * np.sin(2.0 * np.pi * f * t + 0.7 * c + 1.3 * i)
    return torch.from_numpy(x.astype(np.float32))


def synth_audio_device(n: int, seed: int, device, nb_samples: int = 1, nb_channels: int = 2) -> torch.Tensor:
    """The same recipe generated ON the device (torch's device generator seeded with ``seed``): for workloads
    whose inputs would take minutes to draw with NumPy (bench.py's 50-track set, 1.3e9 samples).  Values are
    reproducible per (seed, device type), not equal to ``synth_audio``'s."""
    g = torch.Generator(device=device).manual_seed(int(seed))
    x = torch.rand((nb_samples, nb_channels, n), generator=g, device=device, dtype=torch.float32) - 0.5
    t = torch.arange(n, device=device, dtype=torch.float64) / SAMPLE_RATE
    for i, f in enumerate((110.0, 1760.0, 9000.0)):
        for c in range(nb_channels):
            x[:, c, :] += (0.1 * torch.sin(2.0 * np.pi * f * t + 0.7 * c + 1.3 * i)).float()
    return x
