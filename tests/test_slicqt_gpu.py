"""GPU parity: HIP sliCQT / isliCQT (through the C ABI) against the CPU oracle
and the reference-generated fixtures.  Tolerances: coefficients reach ~36, so
2e-4 abs there is ~5e-6 relative; waveform round trip 1e-4 RMS / 1e-3 max-abs is
the bar BASELINE.json states, we hold far tighter."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from xumx_slicq_amd.synth import synth_audio

pytestmark = pytest.mark.gpu
KEEP = [0, 1, 2, 4, 33, 69]


@pytest.fixture(scope="module")
def fb():
    from xumx_slicq_amd.transforms import NSGTBase, make_filterbanks
    base = NSGTBase("bark", 262, 32.9, device="cuda")
    enc, dec = make_filterbanks(base)
    return base, enc, dec


@pytest.mark.parametrize("n", [9031, 70000])
def test_forward_matches_oracle_and_golden(fb, oracle_plan, n):
    from oracle import slicqt as O
    base, enc, dec = fb
    x = synth_audio(n, seed=20260101 + n)
    C = enc(x.cuda())
    Co = O.forward(oracle_plan, x)
    g = load_golden(f"slicqt_{n}.npz")
    assert len(C) == 70
    for i, (a, b) in enumerate(zip(C, Co)):
        assert a.shape == b.shape and a.is_contiguous() and a.dtype == torch.float32
        assert float((a.cpu() - b).abs().max()) < 2e-4, i
    for i in (range(70) if n == 9031 else KEEP):
        assert float((C[i].cpu() - torch.from_numpy(g[f"fwd_{i}"])).abs().max()) < 2e-4, i


@pytest.mark.parametrize("n", [9031, 70000])
def test_inverse_of_perturbed_coefficients_matches_golden(fb, oracle_plan, n):
    from oracle import slicqt as O
    base, enc, dec = fb
    g = load_golden(f"slicqt_{n}.npz")
    x = synth_audio(n, seed=20260101 + n)
    Cref = O.forward(oracle_plan, x)
    rng = np.random.default_rng(n)
    P = [cb + torch.from_numpy((0.1 * rng.standard_normal(cb.shape)).astype(np.float32))
         for cb in [torch.from_numpy(g[f"fwd_{i}"]) if f"fwd_{i}" in g else Cref[i] for i in range(70)]]
    Pd = [p.cuda() for p in P]          # separate allocations -> exercises the packing path
    keep = [p.clone() for p in Pd]
    y = dec(Pd, n)
    assert all(torch.equal(a, b) for a, b in zip(Pd, keep)), "decoder must not clobber its input"
    assert y.shape == (1, 2, n)
    d = y.cpu() - torch.from_numpy(g["inv"])
    assert float(d.abs().max()) < 2e-5 and float(d.pow(2).mean().sqrt()) < 5e-6
    yo = O.inverse(oracle_plan, P, n)
    assert float((y.cpu() - yo).abs().max()) < 2e-5


@pytest.mark.parametrize("n,B", [(9031, 1), (100000, 3), (441000, 1), (2621440, 1)])
def test_round_trip_is_perfect_reconstruction(fb, n, B):
    base, enc, dec = fb
    x = synth_audio(n, seed=7 + n, nb_samples=B).cuda()
    C = enc(x)
    assert C[0].shape[3] == base.plan.num_slices(n)
    y = dec(C, n)                         # views of one arena -> zero-copy path
    d = (y - x)
    assert y.shape == x.shape
    assert float(d.abs().max()) < 1e-5 and float(d.pow(2).mean().sqrt()) < 1e-6


def test_linearity_and_seven_dim_blocks(fb):
    base, enc, dec = fb
    n = 50000
    a = synth_audio(n, seed=1).cuda()
    b = synth_audio(n, seed=2).cuda()
    Ca, Cb, Cab = enc(a), enc(b), enc(2.0 * a - 0.5 * b)
    for i in (0, 1, 30, 69):
        assert float((Cab[i] - (2.0 * Ca[i] - 0.5 * Cb[i])).abs().max()) < 3e-4
    # 7-D input (targets, B, C, F, S, T, 2) as Unmix returns it
    Y = [torch.stack([ca, cb, 0.5 * ca, ca - cb]) for ca, cb in zip(Ca, Cb)]
    y = dec(Y, n)
    assert y.shape == (4, 1, 2, n)
    assert float((y[0] - a).abs().max()) < 1e-5 and float((y[3] - (a - b)).abs().max()) < 2e-5


def test_ragged_lengths(fb):
    base, enc, dec = fb
    for n in (9031, 9032, 13545, 13546, 18059, 18061, 27090, 27091):
        x = synth_audio(n, seed=n).cuda()
        assert float((dec(enc(x), n) - x).abs().max()) < 1e-5


def test_rocfft_and_lds_fft_backends_agree(fb, oracle_plan):
    """The Bark-262 plan runs on the hand-written LDS FFT; rocFFT stays as the generic backend."""
    from oracle import slicqt as O
    base, enc, dec = fb
    eng = base.nsgt
    n = 70000
    x = synth_audio(n, seed=5).cuda()
    try:
        eng.set_fft_backend(1)
        C_roc = [c.clone() for c in enc(x)]
        y_roc = dec(C_roc, n)
    finally:
        eng.set_fft_backend(0)
    C_lds = enc(x)
    y_lds = dec(C_roc, n)
    Co = O.forward(oracle_plan, x.cpu())
    for i in range(70):
        assert float((C_lds[i] - C_roc[i]).abs().max()) < 2e-4, i
        assert float((C_lds[i].cpu() - Co[i]).abs().max()) < 2e-4, i
    assert float((y_lds - y_roc).abs().max()) < 5e-6
    assert float((y_lds - x).abs().max()) < 1e-5


def test_radix4_band_kernel_matches_dense_gemm(fb, oracle_plan):
    """Bands with Lg >= 24 (XSQ_D4_MIN_LG_DEFAULT) run on the radix-4 DFT kernel by default; the dense grouped GEMM is the
    reference implementation of the same sums."""
    from oracle import slicqt as O
    base, enc, dec = fb
    eng = base.nsgt
    n = 100000
    x = synth_audio(n, seed=9, nb_samples=2).cuda()
    try:
        eng.set_band_radix4(False)
        C_dense = [c.clone() for c in enc(x)]
        y_dense = dec(C_dense, n)
    finally:
        eng.set_band_radix4(True)
    C_r4 = enc(x)
    y_r4 = dec(C_dense, n)
    Co = O.forward(oracle_plan, x.cpu())
    for i in range(70):
        assert float((C_r4[i] - C_dense[i]).abs().max()) < 1e-4, i
        assert float((C_r4[i].cpu() - Co[i]).abs().max()) < 2e-4, i
    assert float((y_r4 - y_dense).abs().max()) < 5e-6
    assert float((y_r4 - x).abs().max()) < 1e-5
    # rocFFT backend + radix-4 (arena-layout output of the synthesis kernel)
    try:
        eng.set_fft_backend(1)
        y_roc = dec(C_dense, n)
    finally:
        eng.set_fft_backend(0)
    assert float((y_roc - y_dense).abs().max()) < 5e-6


@pytest.mark.parametrize("n,lead", [(70000, (1, 2)), (9031, (4, 1, 2)), (30000, (2, 2))])
def test_short_bands_in_kernel_equal_the_dense_gemm_path(fb, n, lead):
    """Inverse transform A/B: bands with Lg < 48 synthesised inside the slice-FFT kernel (radix-4 stage + 4..15-point
    codelets; an opt-in experiment) against the dense DFT-matrix GEMM with its workspace round trip (default);
    even/odd slice counts, odd and even output row offsets (the fused overlap-add takes 8-byte stores only on
    aligned rows)."""
    base, enc, dec = fb
    eng = base.nsgt
    x = synth_audio(n, seed=77 + n, nb_samples=int(np.prod(lead)) // 2).cuda().view(*lead, n)
    rng = torch.Generator(device="cuda").manual_seed(n)
    P = [c + 0.1 * torch.randn(c.shape, generator=rng, device="cuda") for c in enc(x)]
    try:
        eng.set_short_inline(False)
        ref = dec(P, n).clone()
        eng.set_short_inline(True)
        got = dec(P, n).clone()
        # odd length and an odd row offset: the 4-byte path of the fused overlap-add
        ref_o = dec(P, n - 1).clone()
        arena, ld, S = eng.table.as_arena(list(P))
        BC = int(np.prod(ld))
        out = torch.full((BC * (n + 3) + 1,), float("nan"), device="cuda")
        offs = (torch.arange(BC, device="cuda") * (n + 3) + 1).to(torch.int64)
        eng.backward(arena, BC, S, n - 1, out=out, row_offsets=offs)
    finally:
        eng.set_short_inline(False)
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (*lead, n)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-6 * max(1.0, scale), float((got - ref).abs().max())
    placed = torch.stack([out[1 + r * (n + 3): 1 + r * (n + 3) + n - 1] for r in range(BC)]).view(*lead, n - 1)
    assert torch.equal(placed, ref_o)
    # nothing outside the rows was touched
    assert bool(torch.isnan(out[0])) and bool(torch.isnan(out[1 + n - 1: 1 + n + 3]).all())


def test_packed_butterflies_are_a_diagnostic_build_only(fb):
    """The packed-fp32 slice FFT kernels (k_slice_rfft<512, true> / k_slice_irfft<512, true>, slice_fft.h) are not part of
    the product library: xsq_plan_set_packed_fft(1) is refused with a message.  A diagnostic build (make PACKED_FFT=1,
    selected through XSQ_LIB) accepts it, and there the packed kernels must be bitwise the scalar ones."""
    from xumx_slicq_amd import _lib
    base, enc, dec = fb
    eng = base.nsgt
    try:
        eng.set_packed_fft(True)
    except _lib.XsqError as e:
        assert "built without the packed-fp32" in str(e) and not eng._packed_fft
        return
    try:
        for n in (9031, 70001, 300_000):
            x = synth_audio(n, seed=7 + n).cuda()
            eng.set_packed_fft(False)
            C0 = [c.clone() for c in enc(x)]
            y0 = dec(C0, n).clone()
            eng.set_packed_fft(True)
            C1 = enc(x)
            assert all(torch.equal(a, b) for a, b in zip(C0, C1)), n
            y1 = dec(C1, n)
            assert torch.equal(y0, y1), (n, float((y0 - y1).abs().max()))
    finally:
        eng.set_packed_fft(False)


def test_complex_product_band_kernel_arm_agrees_with_the_default(fb, tmp_path):
    """XSQ_D4_SYM=0 keeps the complex-product radix-4 kernel (csrc/band_dft4.h) as an A/B arm beside the default pair-contracted
    form (csrc/band_dft4s.h, a quarter of the matrix work).  The switch is read once per process: the arm runs in a child."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    base, enc, dec = fb
    n = 60000
    x = synth_audio(n, seed=31, nb_samples=1).cuda()
    C = enc(x)
    y = dec([c.clone() for c in C], n)
    out = tmp_path / "arm.pt"
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "from xumx_slicq_amd.transforms import NSGTBase, make_filterbanks\n"
            "from xumx_slicq_amd.synth import synth_audio\n"
            "base = NSGTBase('bark', 262, 32.9, device='cuda'); enc, dec = make_filterbanks(base)\n"
            "x = synth_audio(%d, seed=31, nb_samples=1).cuda(); C = enc(x); y = dec([c.clone() for c in C], %d)\n"
            "torch.save({'C': [c.cpu() for c in C], 'y': y.cpu()}, %r)\n" % (ROOT, n, n, str(out)))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, XSQ_D4_SYM="0"), timeout=600, capture_output=True)
    arm = torch.load(out)
    for i in range(70):
        assert float((C[i].cpu() - arm["C"][i]).abs().max()) < 1e-4, i
    assert float((y.cpu() - arm["y"]).abs().max()) < 5e-6
