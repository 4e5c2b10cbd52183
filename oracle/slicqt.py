"""Oracle: Bark-scale sliced Constant-Q transform (sliCQT) and its inverse.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU fp32 restatement of
``/root/reference/xumx_slicq_v2/nsgt/*.py`` + ``transforms.py``.

The plan follows the reference's arithmetic step by step, in the same
precisions (fp32 torch ops for M / rfbas / g, fp64 for the dual windows),
because the integer tables hang on fp32 rounding (hard part 6).
The transforms use the closed forms that SURVEY.md 8(a) F* / I2 verified
against the reference: the quarter rotation of ``slicing`` and the ``arrange``
rolls cancel because every centre bin is even and every band length is a
multiple of 4.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

PI = math.pi


# --------------------------------------------------------------------------
# plan
# --------------------------------------------------------------------------
def bark_scale(fmin: float, fmax: float, bins: int):
    """(f, q) of the Bark scale.  nsgt/fscale.py:56-89 (BarkScale), 25-38
    (Scale.__call__), 15-23 (Scale.Q: central difference, dbnd=1e-8, doubles)."""
    bmin = 6.0 * math.asinh(fmin / 600.0)
    bmax = 6.0 * math.asinh(fmax / 600.0)
    bbnd = (bmax - bmin) / (bins - 1)
    dbnd = 1.0e-8

    def F(b):
        return 600.0 * math.sinh((b * bbnd + bmin) / 6.0)

    f = torch.as_tensor([F(b) for b in range(bins)], dtype=torch.float32)
    q = torch.as_tensor(
        [F(b) * dbnd / (F(b + dbnd) - F(b - dbnd)) for b in range(bins)],
        dtype=torch.float32,
    )
    return f, q


def mel_scale(fmin: float, fmax: float, bins: int):
    """(f, q) of the Mel scale.  nsgt/fscale.py:131-161 (MelScale) with the generic
    central-difference Q (fscale.py:15-23)."""
    hz2mel = lambda f: math.log10(f / 700.0 + 1.0) * 2595.0
    mmin, mmax = hz2mel(fmin), hz2mel(fmax)
    mbnd = (mmax - mmin) / (bins - 1)
    dbnd = 1.0e-8

    def F(b):
        return (math.pow(10.0, (b * mbnd + mmin) / 2595.0) - 1.0) * 700.0

    f = torch.as_tensor([F(b) for b in range(bins)], dtype=torch.float32)
    q = torch.as_tensor([F(b) * dbnd / (F(b + dbnd) - F(b - dbnd)) for b in range(bins)], dtype=torch.float32)
    return f, q


def suggested_sllen_trlen(f: torch.Tensor, q: torch.Tensor, sr: float):
    """nsgt/fscale.py:40-53."""
    Ls = int(torch.ceil(max((q * 8.0 * sr) / f)))
    Ls = Ls + -Ls % 4
    tr = Ls // 4
    tr = tr + -tr % 2
    return Ls, tr


def _hann(l: int) -> torch.Tensor:
    """nsgt/util.py:5-11 (fp64, peak at index 0)."""
    r = torch.arange(l, dtype=torch.float64)
    return 0.5 * (torch.cos(r * (PI * 2.0 / l)) + 1.0)


def _blackharr(n: int) -> torch.Tensor:
    """nsgt/util.py:14-46 with mod=True, l=n: modified Blackman-Harris stored
    peak-at-index-0 (the two halves swapped)."""
    # the reference divides a python double by a 0-dim fp32 tensor, i.e. the
    # step 2*pi/nn is an fp32 quotient of fp32 operands; keep that rounding.
    nn = torch.tensor(float((n // 2) * 2), dtype=torch.float32)
    k = torch.arange(n)
    bh = (
        0.35872
        - 0.48832 * torch.cos(k * (2 * PI / nn))
        + 0.14128 * torch.cos(k * (4 * PI / nn))
        - 0.01168 * torch.cos(k * (6 * PI / nn))
    )
    return torch.hstack((bh[-(n // 2):], bh[: -(n // 2)]))


def tukey_slice_window(L: int, tr: int) -> torch.Tensor:
    """nsgt/slicing.py:7-18 (makewnd): 0 | rising Hann half | 1 | falling | 0."""
    h = L // 4
    htr = tr // 2
    w = _hann(2 * tr)
    tw = torch.empty(L, dtype=torch.float32)
    tw[: h - htr] = 0
    tw[h - htr: h + htr] = w[tr:]
    tw[h + htr: 3 * h - htr] = 1
    tw[3 * h - htr: 3 * h + htr] = w[:tr]
    tw[3 * h + htr:] = 0
    return tw


@dataclass
class Plan:
    fs: float
    L: int                      # slice length (sllen)
    tr: int                     # transition length (trlen)
    nbands: int                 # bands used on the real transform (DC..Nyquist)
    Lg: np.ndarray              # (nbands,) int   band lengths  (M == len(g))
    c: np.ndarray               # (nbands,) int   centre bins   (rfbas)
    g: List[np.ndarray]         # nbands x fp32   analysis windows, peak at 0
    gd: List[np.ndarray]        # nbands x fp64   dual windows, peak at 0
    tw: np.ndarray              # (L,) fp32       slice window
    blocks: list = field(default_factory=list)  # [(first_band, F_b, T_b)]

    @property
    def h(self):
        return self.L // 4

    def nslices(self, n: int) -> int:
        """Slices produced for an n-sample signal (nsgt/slicing.py:47-72)."""
        nb = -(-n // self.h)
        return (nb + 1) // 2 + 1


def make_plan(fscale="bark", fbins=262, fmin=32.9, fmax=22050.0, fs=44100.0,
              min_win=16) -> Plan:
    """transforms.py:21-71 (NSGTBase) -> nsgt/slicq.py:70-151 (NSGT_sliced
    with real=True, multichannel=True, Qvar=1, reducedform=0)."""
    if fscale == "bark":
        f, q = bark_scale(fmin, fmax, fbins)
    elif fscale == "mel":
        f, q = mel_scale(fmin, fmax, fbins)
    else:
        raise ValueError("oracle covers the Bark and Mel scales (SURVEY.md 2, row 2; 8(f) rank 3)")
    L, tr = suggested_sllen_trlen(f, q, fs)

    # ---- nsgt/nsgfwin_sl.py:8-111 -------------------------------------
    nf = fs / 2.0
    lim = int(torch.argmax((f >= nf).long()))
    if lim != 0:                                   # :27-30 drop f >= Nyquist
        f, q = f[:lim], q[:lim]
    lbas = len(f)
    frqs = torch.zeros(lbas + 2, dtype=torch.float32)
    frqs[1:-1] = f
    frqs[-1] = nf
    fbas = torch.cat((frqs, fs - torch.flip(frqs, (0,))[1:-1]))   # :46-53
    fbas *= float(L) / fs                                          # :55
    M = torch.zeros(fbas.shape, dtype=torch.float32)               # :57-72
    M[0] = 2 * fbas[1]
    M[1] = fbas[1] / q[0]
    for k in list(range(2, lbas)) + [lbas + 1]:
        M[k] = fbas[k + 1] - fbas[k - 1]
    M[lbas] = fbas[lbas] / q[lbas - 1]
    M[lbas + 2: 2 * (lbas + 1)] = torch.flip(M[1: lbas + 1], (0,))
    M *= 1 / 4.0
    M = torch.round(M).int()
    M *= 4
    M = torch.clip(M, min_win, torch.inf)                          # :82
    Mi = [int(m) for m in M]
    g = [_blackharr(m).to(torch.float32) for m in Mi]              # :84-85
    for kk in (1, lbas + 2):                                       # :89-103
        if Mi[kk - 1] > Mi[kk]:
            a, b = Mi[kk - 1], Mi[kk]
            gk = torch.ones(a, dtype=torch.float32)
            gk[a // 2 - b // 2: a // 2 + int(math.ceil(b / 2.0))] = _hann(b)
            g[kk - 1] = gk
    rfbas = torch.round(fbas / 2.0).int() * 2                      # :105

    # ---- nsgt/util.py:72-100 (calcwinrange) ---------------------------
    shift = torch.zeros(len(rfbas), dtype=rfbas.dtype)
    shift[1:] = rfbas[1:] - rfbas[:-1]
    shift[0] = -rfbas[-1] % L
    timepos = torch.cumsum(shift, 0)
    nn = int(timepos[-1])
    assert nn == L, (nn, L)
    timepos = timepos - shift[0]
    wins = []
    for gi, tp in zip(g, timepos):
        lg = len(gi)
        wins.append((torch.arange(-(lg // 2), lg - lg // 2) + int(tp)) % nn)

    # ---- nsgt/util.py:103-116 (nsdual), fp64 --------------------------
    d = torch.zeros(nn, dtype=torch.float64)
    for gi, mi, wi in zip(g, Mi, wins):
        xa = torch.square(torch.fft.fftshift(gi)) * float(mi)   # fp32, as the reference
        d[wi] += xa
    gd = [gi.to(torch.float64) / torch.fft.ifftshift(d[wi]) for gi, wi in zip(g, wins)]

    nb = len(g) // 2 + 1                 # slicq.py:123-131: sl = slice(0, len(g)//2+1)
    Lg = np.array(Mi[:nb], dtype=np.int64)
    c = np.array([int(t) for t in timepos[:nb]], dtype=np.int64)
    assert np.all(Lg % 4 == 0) and np.all(c % 2 == 0)
    # consecutive equal-Lg bands are bucketed into blocks, nsgt/nsgtf.py:66-78
    blocks, j = [], 0
    while j < nb:
        k = j
        while k + 1 < nb and Lg[k + 1] == Lg[j]:
            k += 1
        blocks.append((j, k - j + 1, int(Lg[j])))
        j = k + 1
    return Plan(
        fs=fs, L=L, tr=tr, nbands=nb, Lg=Lg, c=c,
        g=[gi.numpy().copy() for gi in g[:nb]],
        gd=[gi.numpy().copy() for gi in gd[:nb]],
        tw=tukey_slice_window(L, tr).numpy(),
        blocks=blocks,
    )


def _sq(Lg: int) -> np.ndarray:
    """signed frequency offset of window index q (peak-at-0 storage)."""
    q = np.arange(Lg)
    return np.where(q < (Lg + 1) // 2, q, q - Lg)


# --------------------------------------------------------------------------
# forward: NSGT_SL.forward
# --------------------------------------------------------------------------
def forward(plan: Plan, x: torch.Tensor) -> List[torch.Tensor]:
    """x (..., n) fp32 -> list over blocks of (..., F_b, S, T_b, 2) fp32.

    transforms.py:106-131 (NSGT_SL.forward) -> nsgt/slicq.py:182-196 ->
    slicing (nsgt/slicing.py:21-72) -> nsgtf_sl (nsgt/nsgtf.py:7-84) ->
    arrange (nsgt/slicq.py:13-33); closed form SURVEY.md 8(a) F*:
      coef[s,ch,j,:] = (-1)^(c_j/2) IFFT_Lg( g_j[q] U_s[(c_j+sq(q)) mod L] ),
      U_s = FFT_L( tw * xpad[(2s-2)h : (2s+2)h] ).
    """
    lead = x.shape[:-1]
    n = x.shape[-1]
    xb = x.reshape(-1, n).to(torch.float32)
    L, h = plan.L, plan.h
    S = plan.nslices(n)
    xpad = torch.zeros(xb.shape[0], (2 * S + 2) * h, dtype=torch.float32)
    xpad[:, 2 * h: 2 * h + n] = xb
    seg = xpad.unfold(-1, L, 2 * h)[:, :S]                 # (BC, S, L)
    U = torch.fft.fft(seg * torch.from_numpy(plan.tw))     # nsgtf.py:40
    out = []
    for (j0, F, T) in plan.blocks:
        sq = _sq(T)
        idx = torch.from_numpy((plan.c[j0:j0 + F, None] + sq[None, :]) % L)  # (F,T)
        gw = torch.from_numpy(np.stack(plan.g[j0:j0 + F]))                   # (F,T)
        t = U[:, :, idx] * gw                                                # nsgtf.py:55
        cb = torch.fft.ifft(t)                                               # :69,81
        sign = torch.from_numpy(np.where((plan.c[j0:j0 + F] // 2) % 2 == 0, 1.0, -1.0)
                                .astype(np.float32))
        cb = cb * sign[None, None, :, None]
        cb = cb.permute(0, 2, 1, 3).contiguous()                             # (BC,F,S,T)
        out.append(torch.view_as_real(cb).reshape(*lead, F, S, T, 2))
    return out


# --------------------------------------------------------------------------
# inverse: INSGT_SL.forward
# --------------------------------------------------------------------------
def inverse(plan: Plan, X_list: List[torch.Tensor], length: int) -> torch.Tensor:
    """list of (*lead, F_b, S, T_b, 2) -> (*lead, length).  Does NOT modify X_list.

    transforms.py:154-178 -> nsgt/slicq.py:198-230 -> nsigtf_sl
    (nsgt/nsigtf.py:5-106) -> unslicing (nsgt/unslicing.py:33-69); closed form
    SURVEY.md 8(a) I2:
      fr_s[c_j+sq(q)] += (-1)^(c_j/2) Lg gd_j[q] FFT_Lg(coef_s,j)[q]   (bins 0..L/2)
      seg_s = irfft_L(fr_s);  y[(2s-2)h + p] += seg_s[p].
    """
    L, h = plan.L, plan.h
    lead = X_list[0].shape[:-4]
    S = X_list[0].shape[-3]
    BC = int(np.prod(lead)) if len(lead) else 1
    fr = torch.zeros(BC, S, L // 2 + 1, dtype=torch.complex64)
    for (j0, F, T), Xb in zip(plan.blocks, X_list):
        cb = torch.view_as_complex(Xb.reshape(BC, F, S, T, 2).contiguous())
        fc = torch.fft.fft(cb)                                   # nsigtf.py:29-32
        sq = _sq(T)
        for f in range(F):
            j = j0 + f
            sign = 1.0 if (plan.c[j] // 2) % 2 == 0 else -1.0
            w = torch.from_numpy(plan.gd[j] * (T * sign)).to(torch.complex64)   # :91-92
            k = plan.c[j] + sq
            keep = (k >= 0) & (k <= L // 2)
            fr[:, :, torch.from_numpy(k[keep])] += (fc[:, f] * w)[:, :, torch.from_numpy(keep)]
    seg = torch.fft.irfft(fr, n=L)                               # nsigtf.py:99-103
    y = torch.zeros(BC, (2 * S + 2) * h, dtype=torch.float32)
    for s in range(S):                                           # unslicing.py:59-66
        y[:, 2 * s * h: 2 * s * h + L] += seg[:, s]
    y = y[:, 2 * h: 2 * h + length]                              # slicq.py:218-229
    return y.reshape(*lead, length)


def complex_norm(spec):
    """transforms.py:181-208 (ComplexNorm): magnitude of list or tensor."""
    if isinstance(spec, list):
        return [torch.abs(torch.view_as_complex(b)) for b in spec]
    return torch.abs(torch.view_as_complex(spec))
