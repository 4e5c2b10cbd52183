"""Audio front/back end: preprocess_audio shape rules
(data.py:98-156) and wav I/O round trips without torchaudio."""
import struct
import warnings

import numpy as np
import pytest
import torch

from xumx_slicq_amd import audio as A


def test_preprocess_audio_shape_rules():
    t = torch.arange(100, dtype=torch.float32)
    assert A.preprocess_audio(t).shape == (1, 2, 100)                      # 1-D mono -> duplicated
    assert torch.equal(A.preprocess_audio(t)[0, 0], A.preprocess_audio(t)[0, 1])
    assert A.preprocess_audio(torch.zeros(2, 100)).shape == (1, 2, 100)    # (C, T)
    assert A.preprocess_audio(torch.zeros(100, 2)).shape == (1, 2, 100)    # (T, C) -> swapped
    assert A.preprocess_audio(torch.zeros(3, 2, 100)).shape == (3, 2, 100)
    assert A.preprocess_audio(torch.zeros(5, 100)).shape == (5, 2, 100)    # all dims > 2: batch of mono
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = A.preprocess_audio(torch.arange(4 * 50, dtype=torch.float32).view(1, 4, 50))
        assert out.shape == (1, 2, 50) and len(w) == 1
        assert torch.equal(out[0, 1], torch.arange(50, 100, dtype=torch.float32))
    with pytest.raises(ValueError):
        A.preprocess_audio(t, rate=48000, model_rate=44100.0)
    assert A.preprocess_audio(t, rate=44100, model_rate=torch.as_tensor(44100.0)).shape == (1, 2, 100)


def test_float_wav_round_trip(tmp_path):
    x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (2, 1234)).astype(np.float32))
    p = str(tmp_path / "a.wav")
    A.save_wav_float(p, x, 44100)
    y, rate = A.load_audio(p)
    assert rate == 44100 and torch.equal(x, y)
    info = A.load_info(p)
    assert info["samples"] == 1234 and info["channels"] == 2 and info["samplerate"] == 44100
    seg, _ = A.load_audio(p, start=100 / 44100, dur=200 / 44100)
    assert torch.equal(seg, x[:, 100:300])


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
def test_pcm_wav_decoding(tmp_path, bits):
    rng = np.random.default_rng(bits)
    n, ch = 500, 2
    if bits == 8:
        v = rng.integers(0, 256, (n, ch)); raw = v.astype(np.uint8).tobytes(); want = (v - 128) / 128.0
    elif bits == 16:
        v = rng.integers(-32768, 32768, (n, ch)); raw = v.astype("<i2").tobytes(); want = v / 32768.0
    elif bits == 24:
        v = rng.integers(-(1 << 23), 1 << 23, (n, ch))
        u = (v & 0xFFFFFF).astype(np.uint32)
        raw = np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], -1).astype(np.uint8).tobytes()
        want = v / 8388608.0
    else:
        v = rng.integers(-(1 << 31), 1 << 31, (n, ch)); raw = v.astype("<i4").tobytes(); want = v / 2147483648.0
    fmt = struct.pack("<HHIIHH", 1, ch, 44100, 44100 * ch * bits // 8, ch * bits // 8, bits)
    p = tmp_path / "p.wav"
    junk = b"LISTxx"            # an odd-sized chunk before fmt must be skipped with its pad byte
    body = struct.pack("<4sI", b"JUNK", 5) + b"12345" + b"\0" + struct.pack("<4sI", b"fmt ", 16) + fmt + \
        struct.pack("<4sI", b"data", len(raw)) + raw
    p.write_bytes(struct.pack("<4sI4s", b"RIFF", 4 + len(body), b"WAVE") + body)
    y, rate = A.load_audio(str(p))
    assert y.shape == (ch, n) and rate == 44100
    assert np.allclose(y.numpy().T, want.astype(np.float32), atol=1e-7)


def test_evaluation_harness_scores_a_track_with_a_stand_in_separator():
    """evaluation.py:16-44: audio -> preprocess -> separator -> to_dict -> scores (global SDR when museval is absent)."""
    import types
    import numpy as np
    import torch
    from xumx_slicq_amd.evaluation import global_sdr, separate_and_evaluate
    from xumx_slicq_amd.separator import Separator
    rng = np.random.default_rng(3)
    T = 4000
    stems = {n: rng.standard_normal((T, 2)).astype(np.float32) * 0.1 for n in Separator.sources}
    track = types.SimpleNamespace(audio=sum(stems.values()), rate=44100,
                                  targets={n: types.SimpleNamespace(audio=a) for n, a in stems.items()})

    class Oracle:                      # returns the true stems, slightly off for one target
        sample_rate = torch.tensor(44100.0)
        to_dict = staticmethod(Separator.to_dict)

        def __call__(self, audio):
            assert audio.shape == (1, 2, T)
            out = torch.stack([torch.from_numpy(stems[n].T.copy())[None] for n in Separator.sources])
            out[1] = out[1] * 0.9
            return out

    res = separate_and_evaluate(Oracle(), track, device="cpu")
    assert res["metric"] in ("global-sdr", "museval-bsseval-v4")
    assert set(res["estimates"]) == set(Separator.sources) and res["estimates"]["bass"].shape == (T, 2)
    if res["metric"] == "global-sdr":
        assert res["scores"]["bass"] > 80.0                        # exact stem
        assert abs(res["scores"]["vocals"] - 20.0) < 1e-3          # 0.9 x: 10 log10(1 / 0.01)
    assert abs(global_sdr(stems["drums"], 0.5 * stems["drums"]) - 10 * np.log10(4.0)) < 1e-6


@pytest.mark.parametrize("bits,channels", [(16, 2), (16, 1), (8, 2), (32, 2), (24, 2), ("f32", 2), (16, 3)])
def test_load_audio_into_equals_load_then_preprocess(tmp_path, bits, channels):
    """The pipelined CLI's one-pass decoder (audio.load_audio_into: converting, de-interleaving copy straight into the pinned
    staging buffer) returns the bits of load_audio -> preprocess_audio (data.py:64-156) for every supported format."""
    import struct
    import warnings
    from xumx_slicq_amd import audio as A
    rng = np.random.default_rng(3)
    n = 12345
    x = rng.uniform(-1, 1, (channels, n)).astype(np.float32)
    path = str(tmp_path / "a.wav")
    if bits == "f32":
        A.save_wav_float(path, torch.from_numpy(x), 44100)
    else:
        if bits == 8:
            data = (np.round(x * 127) + 128).astype(np.uint8).T.tobytes()
        elif bits == 16:
            data = np.round(x * 32767).astype("<i2").T.copy().tobytes()
        elif bits == 32:
            data = np.round(x.astype(np.float64) * 2147483647).astype("<i4").T.copy().tobytes()
        else:
            v = np.round(x.astype(np.float64) * 8388607).astype(np.int32).T.copy().reshape(-1)
            data = b"".join(int(s).to_bytes(3, "little", signed=True) for s in v)
        fmt = struct.pack("<HHIIHH", 1, channels, 44100, 44100 * (bits // 8) * channels, (bits // 8) * channels, bits)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sI4s", b"RIFF", 4 + 8 + len(fmt) + 8 + len(data), b"WAVE"))
            f.write(struct.pack("<4sI", b"fmt ", len(fmt)) + fmt + struct.pack("<4sI", b"data", len(data)) + data)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sig, rate = A.load_audio(path)
        want = A.preprocess_audio(sig, rate, 44100)[0]
        got, rate2 = A.load_audio_into(path, lambda numel: torch.empty(numel + 7))
    assert rate2 == rate == 44100 and got.shape == want.shape == (2, n) and torch.equal(got, want)
