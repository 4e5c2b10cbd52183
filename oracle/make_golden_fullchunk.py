"""Full-chunk reference fixture: one FULL 2,621,440-sample chunk (S = 292 slices; block 69 has 85,264 frames = 18 Wiener
windows of <= 5000 frames, /root/reference/xumx_slicq_v2/phase.py:43-59, each with its own window maximum,
norbert/__init__.py:257) plus a 98,240-sample tail chunk through the REFERENCE Separator (imported from /root/reference,
never copied) -- the sizes the committed fixtures did not reach (their largest: 262,144 samples, <= 2 windows per block).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_fullchunk        (development container only; ~1-2 min)

Input: chunk 0 and the tail of the bench's own 240 s track (synth_audio(10,584,000, seed 20260101): samples [0, 2,621,440)
and [10,485,760, 10,584,000) back to back, 2,719,680 samples), so that the full-size GPU test's output can be held against
the fixture directly -- chunks are independent (separator.py:153-229).  Offline conv stack, seeded weights 1234, with the
Wiener-EM post-filter and with mix-phase.  Writes tests/golden/stems_fullchunk.npz: the stems at stride 97 and
per-stem checksums (sum, sum of squares, max abs)."""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle.make_golden import WEIGHT_SEED, checksums
from xumx_slicq_amd.synth import synth_audio
from xumx_slicq_amd.weights import seeded_state_dict

OUT = os.path.join(ROOT, "tests", "golden")
CHUNK, TRACK, SEED, STRIDE = 2_621_440, 10_584_000, 20260101, 97


def fixture_input():
    """(1, 2, 2,719,680): chunk 0 + the tail chunk of the bench track."""
    x = synth_audio(TRACK, seed=SEED)
    return torch.cat([x[..., :CHUNK], x[..., 4 * CHUNK:]], dim=-1).contiguous()


def main():
    torch.set_num_threads(8)
    from xumx_slicq_v2.model import Unmix
    from xumx_slicq_v2.separator import Separator
    from xumx_slicq_v2.transforms import ComplexNorm, NSGTBase, make_filterbanks

    base = NSGTBase("bark", 262, 32.9, fs=44100.0, device="cpu")
    enc, dec = make_filterbanks(base, 44100.0)
    cnorm = ComplexNorm()
    with torch.no_grad():
        jag, _ = base.predict_input_size(1, 2, 2.0)
    sd = seeded_state_dict([(b.shape[2], b.shape[4]) for b in jag], seed=WEIGHT_SEED)
    x = fixture_input()
    n = x.shape[-1]
    assert n == CHUNK + 98_240
    d = dict(n=n, chunk_size=CHUNK, stride=STRIDE, seed=SEED, input_sums=checksums(x))
    for name, phasemix in (("offline_wiener", False), ("offline_phasemix", True)):
        m = Unmix(cnorm(jag), realtime=False)
        m.load_state_dict(sd, strict=True)
        m.freeze()
        for blk in m.sliced_umx:                 # the flag read at model.py:264
            blk.realtime = phasemix
        sep = Separator(xumx_model=m, encoder=(enc, dec, cnorm), runtime_backend="torch-cpu", chunk_size=CHUNK, quiet=True)
        sep.freeze()
        with torch.no_grad():
            est = sep(x.clone())
        assert est.shape == (4, 1, 2, n)
        d[f"{name}_sums"] = np.stack([checksums(est[t]) for t in range(4)])
        d[name] = est[..., ::STRIDE].contiguous().numpy()
        print(name, d[f"{name}_sums"])
        del est, sep, m
    path = os.path.join(OUT, "stems_fullchunk.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
