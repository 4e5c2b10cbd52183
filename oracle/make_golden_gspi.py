"""Real-audio fixture: the one real signal the reference ships (/root/reference/.github/gspi.wav: mono, 16-bit PCM,
262,144 samples at 44.1 kHz; used by xumx_slicq_v2/visualization.py:125 and demixui.py:31) through the reference's own
front end and Separator.  Development container only -- the reference is IMPORTED from /root/reference, never copied.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_gspi

Writes tests/golden/stems_gspi.npz: the decoded int16 samples (data held by the reference, not source text), the
float32 (1, 2, T) tensor the reference's preprocess_audio makes of them (mono -> duplicated stereo, data.py:123-146),
and the stems of the reference Separator (seeded weights 1234) for the realtime model and the offline model with
Wiener-EM -- at stride 13 plus per-target checksums -- with the default 2,621,440-sample chunk and with chunk_size =
100,000 (three chunks: the hard concat of separator.py:229-231 on real audio)."""
from __future__ import annotations

import os
import sys
import types
import wave

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle.make_golden import WEIGHT_SEED, checksums
from xumx_slicq_amd.weights import seeded_state_dict

OUT = os.path.join(ROOT, "tests", "golden")


class _Stub(types.ModuleType):            # data.py imports torchaudio / musdb at module level; preprocess_audio uses neither here
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Stub(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return None


def main():
    torch.set_num_threads(8)
    for name in ("torchaudio", "torchaudio.transforms", "musdb", "museval", "torchinfo"):
        sys.modules.setdefault(name, _Stub(name))
    from xumx_slicq_v2.data import preprocess_audio
    from xumx_slicq_v2.model import Unmix
    from xumx_slicq_v2.separator import Separator
    from xumx_slicq_v2.transforms import ComplexNorm, NSGTBase, make_filterbanks

    with wave.open(os.path.join(REF, ".github", "gspi.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 44100)
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    # torchaudio.load(normalize=True) of 16-bit PCM: int16 / 32768 as float32, shape (channels, frames) (data.py:64-95)
    sig = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None, :]
    audio = preprocess_audio(sig, 44100.0, 44100.0)                 # the call of inference.py:25
    assert audio.shape == (1, 2, pcm.size) and torch.equal(audio[0, 0], audio[0, 1])

    base = NSGTBase("bark", 262, 32.9, fs=44100.0, device="cpu")
    enc, dec = make_filterbanks(base, 44100.0)
    cnorm = ComplexNorm()
    with torch.no_grad():
        jag, _ = base.predict_input_size(1, 2, 2.0)
    sd = seeded_state_dict([(b.shape[2], b.shape[4]) for b in jag], seed=WEIGHT_SEED)

    def build(realtime_conv, phasemix):
        m = Unmix(cnorm(jag), realtime=realtime_conv)
        m.load_state_dict(sd, strict=True)
        m.freeze()
        for blk in m.sliced_umx:
            blk.realtime = phasemix
        return m

    d = dict(pcm=pcm, rate=44100, n=pcm.size, stride=13, audio_sums=checksums(audio))
    for name, m in (("realtime", build(True, True)), ("offline_wiener", build(False, False))):
        for cs in (2621440, 100000):
            sep = Separator(xumx_model=m, encoder=(enc, dec, cnorm), runtime_backend="torch-cpu", chunk_size=cs, quiet=True)
            sep.freeze()
            with torch.no_grad():
                est = sep(audio.clone())
            assert est.shape == (4, 1, 2, pcm.size)
            tag = f"{name}_cs{cs}"
            d[f"{tag}_sums"] = np.stack([checksums(est[t]) for t in range(4)])
            d[tag] = est[..., ::13].contiguous().numpy()
            print(tag, d[f"{tag}_sums"][:, 2])
    path = os.path.join(OUT, "stems_gspi.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
