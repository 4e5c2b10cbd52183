"""Pin the CPU oracle against fixtures produced by the reference itself
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import model as omodel
from oracle import separator as osep
from oracle import slicqt as oslicqt
from xumx_slicq_amd.synth import synth_audio

KEEP = [0, 1, 2, 4, 33, 69]


def sums(t):
    a = t.double().flatten()
    return np.array([float(a.sum()), float((a * a).sum()), float(a.abs().max())])


def test_plan_matches_reference(oracle_plan):
    g = load_golden("plan.npz")
    p = oracle_plan
    assert (p.L, p.tr, p.nbands) == (int(g["L"]), int(g["tr"]), int(g["nbands"])) == (18060, 4516, 263)
    assert np.array_equal(p.Lg, g["Lg"])
    assert np.array_equal(p.c % p.L, g["c"])
    assert np.array_equal(np.array([(F, T) for (_, F, T) in p.blocks]), g["blocks"])
    assert len(p.blocks) == 70
    assert np.array_equal(np.concatenate(p.g), g["g"])          # bit exact
    assert np.array_equal(np.concatenate(p.gd), g["gd"])
    assert np.array_equal(p.tw, g["tw"])
    assert p.nslices(int(2.0 * 44100)) == int(g["seq_dur_slices"])
    for n, S in ((9031, 3), (70000, 9), (100000, 13), (441000, 50), (2621440, 292)):
        assert p.nslices(n) == S


@pytest.mark.parametrize("n", [9031, 70000])
def test_forward_inverse_match_reference(oracle_plan, n):
    g = load_golden(f"slicqt_{n}.npz")
    x = synth_audio(n, seed=20260101 + n)
    C = oslicqt.forward(oracle_plan, x)
    assert C[0].shape[3] == int(g["S"])
    for i, cb in enumerate(C):
        assert cb.is_contiguous()
        assert np.allclose(sums(cb), g["fwd_sums"][i], rtol=1e-4, atol=1e-3)
    blocks = range(70) if n == 9031 else KEEP
    for i in blocks:
        ref = torch.from_numpy(g[f"fwd_{i}"])
        assert C[i].shape == ref.shape
        assert float((C[i] - ref).abs().max()) < 2e-5          # values up to ~36
    rng = np.random.default_rng(n)
    P = [cb + torch.from_numpy((0.1 * rng.standard_normal(cb.shape)).astype(np.float32))
         for cb in [torch.from_numpy(g[f"fwd_{i}"]) if f"fwd_{i}" in g else C[i] for i in range(70)]]
    keep = [p.clone() for p in P]
    y = oslicqt.inverse(oracle_plan, P, n)
    assert all(torch.equal(a, b) for a, b in zip(P, keep)), "inverse must not clobber its input"
    ref = torch.from_numpy(g["inv"])
    assert y.shape == ref.shape == (1, 2, n)
    assert float((y - ref).abs().max()) < 5e-6
    # perfect reconstruction
    assert float((oslicqt.inverse(oracle_plan, C, n) - x).abs().max()) < 5e-6


def test_cdae_masks_match_reference(oracle_plan, seeded_sd):
    g = load_golden("cdae_masks_70000.npz")
    n = int(g["n"])
    X = oslicqt.forward(oracle_plan, synth_audio(n, seed=20260101 + n))
    for causal, tag in ((False, "offline"), (True, "causal")):
        for i in KEEP:
            m = omodel.cdae_masks(seeded_sd, i, omodel.abs_of_real_complex(X[i]), causal)
            ref = torch.from_numpy(g[f"mask_{tag}_{i}"])
            assert m.shape == ref.shape
            assert float((m - ref).abs().max()) < 2e-5
            assert 0.02 < float(ref.std()), "seeded weights should give non-trivial masks"


def test_wiener_matches_reference():
    g = load_golden("wiener.npz")
    rng = np.random.default_rng(5)
    mix = torch.from_numpy(rng.standard_normal((1, 2, 2, 26, 200, 2)).astype(np.float32))
    mag = torch.from_numpy(np.abs(rng.standard_normal((4, 1, 2, 2, 26, 200))).astype(np.float32))
    y = omodel.blockwise_wiener(mix, mag)
    ref = torch.from_numpy(g["out_5200"])
    assert y.shape == ref.shape == (4, 1, 2, 2, 26, 200, 2)
    assert float((y - ref).abs().max()) < 2e-5
    # the reference's own test case (tests/test_phase.py:6-12): shape + finite, plus values
    rng = np.random.default_rng(6)
    mix2 = torch.from_numpy(rng.standard_normal((1, 2, 14, 257, 37, 2)).astype(np.float32))
    mag2 = torch.from_numpy(rng.standard_normal((4, 1, 2, 14, 257, 37)).astype(np.float32))
    y2 = omodel.blockwise_wiener(mix2, mag2)
    assert y2.shape == (4, 1, 2, 14, 257, 37, 2) and bool(torch.all(torch.isfinite(y2)))
    sub = torch.from_numpy(g["out_testphase_sub"])
    err = (y2.flatten()[::97] - sub).abs()
    assert float(err.max()) < 1e-3 * max(1.0, float(sub.abs().max()))


@pytest.mark.parametrize("n", [9031, 100000])
@pytest.mark.parametrize("name,causal,wiener", [
    ("realtime", True, False), ("offline_phasemix", False, False), ("offline_wiener", False, True)])
def test_stems_match_reference(oracle_plan, seeded_sd, n, name, causal, wiener):
    g = load_golden(f"stems_{n}.npz")
    x = synth_audio(n, seed=20260101 + n)
    est = osep.separate(oracle_plan, seeded_sd, x, causal=causal, wiener=wiener,
                        chunk_size=int(g["chunk_size"]))
    assert est.shape == (4, 1, 2, n)
    ref = torch.from_numpy(g[name])
    got = est if n == 9031 else est[..., ::7]
    d = got - ref
    rms = float(d.pow(2).mean().sqrt())
    assert rms < 1e-5 and float(d.abs().max()) < 1e-4, (rms, float(d.abs().max()))
    assert float(ref.pow(2).mean().sqrt()) > 1e-2


@pytest.mark.parametrize("name,causal,wiener", [("realtime", True, False), ("offline_wiener", False, True)])
def test_real_audio_stems_match_reference(oracle_plan, seeded_sd, name, causal, wiener):
    """The one real signal the reference ships (.github/gspi.wav: mono, 16-bit, 262,144 samples) through the
    reference's own front end and Separator (oracle/make_golden_gspi.py): the oracle on the same decoded samples,
    one chunk and three chunks of 100,000."""
    from xumx_slicq_amd.audio import preprocess_audio
    g = load_golden("stems_gspi.npz")
    sig = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0)[None, :]
    audio = preprocess_audio(sig, 44100, 44100.0)
    assert audio.shape == (1, 2, int(g["n"])) and torch.equal(audio[0, 0], audio[0, 1])
    a = audio.double().flatten()
    assert np.allclose([float(a.sum()), float((a * a).sum()), float(a.abs().max())], g["audio_sums"], rtol=1e-12)
    for cs in (2621440, 100000):
        est = osep.separate(oracle_plan, seeded_sd, audio, causal=causal, wiener=wiener, chunk_size=cs)
        ref = torch.from_numpy(g[f"{name}_cs{cs}"])
        d = est[..., ::int(g["stride"])] - ref
        rms = float(d.pow(2).mean().sqrt())
        assert rms < 1e-5 and float(d.abs().max()) < 1e-4, (cs, rms, float(d.abs().max()))
        assert float(ref.pow(2).mean().sqrt()) > 1e-3


@pytest.mark.parametrize("name,wiener", [("offline_wiener", True), ("offline_phasemix", False)])
def test_full_chunk_stems_match_reference(oracle_plan, seeded_sd, name, wiener):
    """One FULL chunk (2,621,440 samples: S = 292, block 69 = 85,264 frames = 18 Wiener windows, phase.py:43-59, each with
    its own window maximum, norbert/__init__.py:257) + the 98,240-sample tail chunk through the REFERENCE Separator
    (oracle/make_golden_fullchunk.py -> stems_fullchunk.npz, stride 97 + checksums): the oracle at the size the GPU path is
    benchmarked at.  Input = chunk 0 and the tail of bench.py's 240 s track."""
    g = load_golden("stems_fullchunk.npz")
    cs, n, stride = int(g["chunk_size"]), int(g["n"]), int(g["stride"])
    track = synth_audio(10_584_000, seed=int(g["seed"]))
    x = torch.cat([track[..., :cs], track[..., 4 * cs:]], dim=-1).contiguous()
    assert x.shape[-1] == n == cs + 98_240
    a = x.double().flatten()
    assert np.allclose([float(a.sum()), float((a * a).sum()), float(a.abs().max())], g["input_sums"], rtol=1e-12)
    est = osep.separate(oracle_plan, seeded_sd, x, causal=False, wiener=wiener, chunk_size=cs)
    ref = torch.from_numpy(g[name])
    d = (est[..., ::stride] - ref).double()
    rms, mx = float(d.pow(2).mean().sqrt()), float(d.abs().max())
    print(f"full chunk {name}: oracle vs reference rms {rms:.2e} max {mx:.2e}")
    assert rms < 1e-6 and mx < 1e-5, (rms, mx)          # measured 5e-8 / 5e-7
    sums = np.stack([[float(e.double().sum()), float((e.double() ** 2).sum()), float(e.abs().max())] for e in est])
    assert np.allclose(sums[:, 1], g[f"{name}_sums"][:, 1], rtol=1e-5) and np.allclose(sums[:, 2], g[f"{name}_sums"][:, 2], rtol=1e-4)
    assert float(ref.pow(2).mean().sqrt()) > 1e-2
