"""Host-side sliCQT plan: scale, window lengths, windows, centre bins, duals.

One-off, host-side arithmetic.  The integer tables hang
on fp32 rounding (round-half-even ties, a ceil() 8e-6 away from an integer),
so every step runs in the precision the reference uses: fp32 torch ops for the
band table and the analysis windows, fp64 for the dual windows
(/root/reference/xumx_slicq_v2/nsgt/{fscale,nsgfwin_sl,util}.py).  The result is
checked against reference-generated integers in tests/golden/plan.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
import torch

TWO_PI = 2.0 * math.pi


def _bark_frequencies(fmin: float, fmax: float, bins: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Band centres and Q factors (nsgt/fscale.py:56-89; Q by the reference's
    central difference with step 1e-8 in doubles, fscale.py:15-23)."""
    lo, hi = 6.0 * math.asinh(fmin / 600.0), 6.0 * math.asinh(fmax / 600.0)
    step = (hi - lo) / (bins - 1)
    hz = lambda b: 600.0 * math.sinh((b * step + lo) / 6.0)
    eps = 1.0e-8
    f = [hz(b) for b in range(bins)]
    q = [hz(b) * eps / (hz(b + eps) - hz(b - eps)) for b in range(bins)]
    return torch.tensor(f, dtype=torch.float32), torch.tensor(q, dtype=torch.float32)


def _mel_frequencies(fmin: float, fmax: float, bins: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """nsgt/fscale.py:131-161 (MelScale; Q from the generic central difference)."""
    to_mel = lambda f: math.log10(f / 700.0 + 1.0) * 2595.0
    lo, hi = to_mel(fmin), to_mel(fmax)
    step = (hi - lo) / (bins - 1)
    hz = lambda b: (math.pow(10.0, (b * step + lo) / 2595.0) - 1.0) * 700.0
    eps = 1.0e-8
    f = [hz(b) for b in range(bins)]
    q = [hz(b) * eps / (hz(b + eps) - hz(b - eps)) for b in range(bins)]
    return torch.tensor(f, dtype=torch.float32), torch.tensor(q, dtype=torch.float32)


_SCALES = {"bark": _bark_frequencies, "mel": _mel_frequencies}


def _cosine_sum_window(n: int) -> torch.Tensor:
    """Modified Blackman-Harris of length n with its peak rotated to index 0
    (nsgt/util.py:14-46).  The phase step is an fp32 quotient, as in the reference."""
    period = torch.tensor(float(2 * (n // 2)), dtype=torch.float32)
    k = torch.arange(n)
    w = torch.full((n,), 0.35872)
    for coef, mult in ((-0.48832, 1), (0.14128, 2), (-0.01168, 3)):
        w = w + coef * torch.cos(k * (mult * TWO_PI / period))
    return torch.roll(w, n // 2)


def _hann0(n: int) -> torch.Tensor:
    """fp64 Hann with the peak at index 0 (nsgt/util.py:5-11)."""
    return 0.5 * (torch.cos(torch.arange(n, dtype=torch.float64) * (TWO_PI / n)) + 1.0)


@dataclass
class SliCQPlan:
    """Tables of one sliCQT configuration (bands DC..Nyquist of the real transform)."""
    scale: str
    fbins: int
    fmin: float
    fmax: float
    fs: float
    L: int                     # slice length ("sllen")
    tr: int                    # transition length ("trlen")
    Lg: np.ndarray             # (nbands,) int32 band lengths
    c: np.ndarray              # (nbands,) int32 centre bins
    g: np.ndarray              # (sum Lg,) fp32 analysis windows, each peak-at-0
    gd: np.ndarray             # (sum Lg,) fp64 dual windows
    tw: np.ndarray             # (L,) fp32 slice window
    blocks: List[Tuple[int, int, int]]   # (first_band, F_b, T_b)

    @property
    def nbands(self) -> int:
        return len(self.Lg)

    @property
    def hop(self) -> int:       # half hop between slices ("hhop"); slices advance by 2*hop
        return self.L // 4

    @property
    def ncoefs(self) -> int:    # NSGT_sliced.ncoefs, nsgt/slicq.py:133-137
        return int(self.Lg.max())

    @property
    def coefs_per_slice(self) -> int:
        return int(self.Lg.sum())

    def num_slices(self, n: int) -> int:
        nb = -(-n // self.hop)
        return (nb + 1) // 2 + 1

    def block_shapes(self) -> List[Tuple[int, int]]:
        return [(F, T) for (_, F, T) in self.blocks]


def build_plan(scale: str = "bark", fbins: int = 262, fmin: float = 32.9, fmax: float = 22050.0,
               fs: float = 44100.0, min_win: int = 16) -> SliCQPlan:
    if scale not in _SCALES:
        raise ValueError(f"scale '{scale}' is not on the accelerated path (have: {sorted(_SCALES)})")
    f, q = _SCALES[scale](fmin, fmax, fbins)

    # slice / transition length, nsgt/fscale.py:40-53
    L = int(torch.ceil(torch.max(q * 8.0 * fs / f)))
    L += -L % 4
    tr = L // 4
    tr += -tr % 2

    # keep 0 < f < Nyquist, nsgt/nsgfwin_sl.py:21-30
    nyq = fs / 2.0
    keep = (f > 0) & (f < nyq)
    first = int(torch.argmax(keep.long()))
    last = first + int(keep[first:].long().sum())
    assert bool(keep[first:last].all()) and not bool(keep[last:].any()), "scale must be increasing"
    f, q = f[first:last], q[first:last]
    nb = len(f)                                    # "lbas"

    # centre positions in bins for [DC, f..., Nyquist, mirrored], fp32 (nsgfwin_sl.py:41-55)
    pos = torch.cat((torch.zeros(1), f, torch.tensor([nyq], dtype=torch.float32)))
    pos = torch.cat((pos, fs - torch.flip(pos, (0,))[1:-1]))
    pos *= float(L) / fs

    # band lengths, rounded to multiples of 4 (nsgfwin_sl.py:57-72,82)
    width = torch.zeros_like(pos)
    width[0] = 2 * pos[1]
    width[1] = pos[1] / q[0]
    inner = torch.tensor(list(range(2, nb)) + [nb + 1])
    width[inner] = pos[inner + 1] - pos[inner - 1]
    width[nb] = pos[nb] / q[nb - 1]
    width[nb + 2:] = torch.flip(width[1: nb + 1], (0,))
    width *= 0.25
    M = (torch.round(width).int() * 4).clamp_min(min_win).tolist()

    # analysis windows (nsgfwin_sl.py:84-103): Blackman-Harris, except that the DC and
    # Nyquist windows become flat-topped with Hann edges when they are wider than their
    # neighbour
    g = [_cosine_sum_window(m) for m in M]
    for kk in (1, nb + 2):
        wide, narrow = M[kk - 1], M[kk]
        if wide > narrow:
            plateau = torch.ones(wide, dtype=torch.float32)
            lo = wide // 2 - narrow // 2
            plateau[lo: lo + narrow] = _hann0(narrow).float()
            g[kk - 1] = plateau

    # centre bins, even (nsgfwin_sl.py:105; nsgt/util.py:72-100)
    centre = (torch.round(pos / 2.0).int() * 2).tolist()
    assert (-centre[-1]) % L + centre[-1] == L

    # dual windows: divide by the diagonal of the frame operator (nsgt/util.py:103-116)
    diag = torch.zeros(L, dtype=torch.float64)
    idx = []
    for gi, m, c in zip(g, M, centre):
        bins = (torch.arange(-(m // 2), m - m // 2) + c) % L
        idx.append(bins)
        diag.index_add_(0, bins, (torch.square(torch.fft.fftshift(gi)) * float(m)).double())
    gd = [gi.double() / torch.fft.ifftshift(diag[bins]) for gi, bins in zip(g, idx)]

    # slice window (nsgt/slicing.py:7-18)
    hop, half = L // 4, tr // 2
    edge = _hann0(2 * tr)
    tw = torch.zeros(L, dtype=torch.float32)
    tw[hop - half: hop + half] = edge[tr:].float()
    tw[hop + half: 3 * hop - half] = 1.0
    tw[3 * hop - half: 3 * hop + half] = edge[:tr].float()

    used = len(g) // 2 + 1                          # real transform: bands 0..Nyquist (slicq.py:123-131)
    Lg = np.asarray(M[:used], dtype=np.int32)
    c = np.asarray(centre[:used], dtype=np.int32)
    if np.any(Lg % 4) or np.any(c % 2):
        raise ValueError("plan violates Lg % 4 == 0 / even centre bins; the closed forms need both")
    blocks, j = [], 0
    while j < used:                                 # runs of equal length (nsgt/nsgtf.py:66-78)
        k = j
        while k + 1 < used and Lg[k + 1] == Lg[j]:
            k += 1
        blocks.append((j, k - j + 1, int(Lg[j])))
        j = k + 1
    return SliCQPlan(
        scale=scale, fbins=fbins, fmin=fmin, fmax=fmax, fs=fs, L=L, tr=tr, Lg=Lg, c=c,
        g=torch.cat(g[:used]).numpy().astype(np.float32),
        gd=torch.cat(gd[:used]).numpy().astype(np.float64),
        tw=tw.numpy(), blocks=blocks)
