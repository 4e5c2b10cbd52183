"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/xumx_slicq_hip.h declares (no compute without a GPU), and the
host-side plan matches the reference-generated fixture."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "xumx_slicq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(xsq_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from xumx_slicq_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/xumx_slicq_hip.h but not exported"
        assert n in _lib.EXPORTED, f"{n} has no ctypes signature in _lib.py"
    assert _lib.lib.xsq_abi_version() == 2


def test_null_arguments_are_rejected_not_crashed():
    from xumx_slicq_amd import _lib
    rc = _lib.lib.xsq_plan_create(None, 0, 0, 0, None, None, None, None, None)
    assert rc < 0 and "null" in _lib.last_error()
    assert _lib.lib.xsq_plan_num_slices(None, 100) < 0


def test_product_plan_matches_reference_fixture():
    from xumx_slicq_amd.plan import build_plan
    g = load_golden("plan.npz")
    p = build_plan("bark", 262, 32.9)
    assert (p.L, p.tr, p.nbands, len(p.blocks)) == (18060, 4516, 263, 70)
    assert np.array_equal(p.Lg, g["Lg"]) and np.array_equal(p.c % p.L, g["c"])
    assert np.array_equal(np.array(p.block_shapes()), g["blocks"])
    assert np.array_equal(p.g, g["g"]) and np.array_equal(p.gd, g["gd"]) and np.array_equal(p.tw, g["tw"])
    assert p.coefs_per_slice == 18640 and p.ncoefs == 292
    for n, S in ((9031, 3), (70000, 9), (100000, 13), (441000, 50), (2621440, 292)):
        assert p.num_slices(n) == S


def test_unknown_scale_raises():
    from xumx_slicq_amd.plan import build_plan
    with pytest.raises(ValueError):
        build_plan("cqlog", 100, 30.0)


def test_cpu_tensor_is_refused_loudly():
    import torch
    from xumx_slicq_amd import _lib
    from xumx_slicq_amd.transforms import NSGTBase, make_filterbanks
    base = NSGTBase("bark", 262, 32.9, device="cpu")
    enc, dec = make_filterbanks(base)
    with pytest.raises(_lib.XsqError):
        enc(torch.zeros(1, 2, 9031))
    with pytest.raises(ValueError):
        make_filterbanks(base, 48000.0)


def test_no_kernel_of_the_product_library_contains_packed_fp32_instructions(tmp_path):
    """Every device file is built with packed-fp32 ops OFF (csrc/Makefile NOPK): next to v_mfma_f32_16x16x32_bf16 waves of
    another stream a packed-fp32 slice FFT returned wrong values.  The hand-packed transform kernels
    exist only in a diagnostic build (PACKED_FFT=1); the product flags must leave slicqt.hip -- the file that holds the
    transforms -- without a single v_pk_{fma,add,mul}_f32."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "xumx_slicq_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "NOPK" in mk and "-packed-fp32-ops" in mk and "ifeq ($(PACKED_FFT),1)" in mk
    asm = str(tmp_path / "slicqt.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                    "-S", "--cuda-device-only", os.path.join(csrc, "slicqt.hip"), "-o", asm], check=True, capture_output=True, timeout=900)
    text = open(asm).read()
    assert "k_slice_rfftILi512ELb0" in text and "k_slice_rfftILi512ELb1" not in text and "k_slice_irfftILi512ELb1" not in text
    assert not re.search(r"v_pk_(fma|add|mul)_f32", text)


def _schedule(nb, N, cs, max_stack=8, wiener=0, cap=0, L=18060, sumFT=18640):
    from xumx_slicq_amd import _lib
    buf = np.zeros((4096, 8), dtype=np.int64)
    n = _lib.lib.xsq_separator_schedule(L, sumFT, nb, N, cs, max_stack, wiener, cap, buf.ctypes.data, len(buf))
    assert 0 < n <= len(buf), (n, _lib.last_error())
    return buf[:n]


@pytest.mark.parametrize("wiener", [0, 1])
def test_native_forward_schedule_covers_the_reference_chunk_loop(wiener):
    """The schedule xsq_separator_forward follows (pure host arithmetic, csrc/demix.hip: build_schedule) against the
    reference's chunk loop (separator.py:147-158): every (chunk, sample) pair exactly once with the loop's own lengths,
    stacked passes only over full chunks, never more item-slices per pass than the cap, a batch too large for one pass
    split over sample ranges whose passes form one set -- with a shared window-maximum table under Wiener-EM -- and the
    short chunks marked for the tail stream only when something is stacked and no set is split."""
    h = 18060 // 4
    S = lambda n: ((max(n, 9031) + h - 1) // h + 1) // 2 + 1
    cases = [(1, 10_584_000, 2_621_440, 8, 0), (1, 2_621_440, 2_621_440, 8, 0), (1, 777, 2_621_440, 8, 0), (2, 60000 * 3 + 12345, 60000, 8, 0),
             (5, 150_000, 60000, 8, 20), (32, 2_621_440 * 3 + 100_000, 2_621_440, 8, 0), (3, 30000 * 9 + 777, 30000, 4, 0),
             (40, 300_000, 100_000, 64, 0), (7, 1_000_003, 250_000, 3, 50)]
    for nb, N, cs, max_stack, cap in cases:
        P = _schedule(nb, N, cs, max_stack, wiener, cap)
        lim = min(cap, 7168) if cap else 7168
        seen = {}
        for i, (start, n, k, b0, nbb, tail, set_first, ext) in enumerate(P.tolist()):
            assert k >= 1 and nbb >= 1 and b0 + nbb <= nb and start % cs == 0
            assert k * nbb * S(n) <= max(lim, S(n)), (nb, N, cs, i)          # (a single item may exceed a tiny test cap)
            if k > 1:
                assert n == cs and tail == 0 and k * nbb <= max(max_stack, nbb)
            for j in range(k):
                want = min(cs, N - (start + j * cs))
                assert want == n
                for b in range(b0, b0 + nbb):
                    assert (start + j * cs, b) not in seen
                    seen[(start + j * cs, b)] = i
            first = P[set_first]
            assert (first[0], first[2]) == (start, k)                           # a set = the sample ranges of the same chunks
            assert ext == (1 if (wiener and sum(1 for q in P.tolist() if q[6] == set_first) > 1) else 0)
        assert sorted(seen) == [(s, b) for s in range(0, N, cs) for b in range(nb)]
        any_split = any(q[7] for q in P.tolist())
        any_stacked = any(q[2] > 1 for q in P.tolist())
        for start, n, k, b0, nbb, tail, set_first, ext in P.tolist():
            if k == 1:
                assert tail == (1 if (any_stacked and not any_split) else 0)
    # the bench track: one stacked pass of four chunks and the tail beside it
    P = _schedule(1, 10_584_000, 2_621_440, 8, wiener).tolist()
    assert [(p[0], p[1], p[2], p[5]) for p in P] == [(0, 2_621_440, 4, 0), (4 * 2_621_440, 98_240, 1, 1)]
    # nb = 32 at full chunks: 24 + 8 samples per chunk (7168 // 292), no stacking
    P = _schedule(32, 2_621_440 * 2, 2_621_440, 8, wiener).tolist()
    assert [(p[2], p[3], p[4]) for p in P] == [(1, 0, 24), (1, 24, 8), (1, 0, 24), (1, 24, 8)]
