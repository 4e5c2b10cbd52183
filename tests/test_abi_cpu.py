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
    assert _lib.lib.xsq_abi_version() == 1


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


def test_only_the_hand_packed_fft_kernels_contain_packed_fp32_instructions(tmp_path):
    """csrc/Makefile builds slicqt.hip with packed-fp32 ops ENABLED (the assembler needs the feature for the hand-placed
    v_pk_* butterflies of k_slice_rfft<512, true> / k_slice_irfft<512, true>) and the SLP vectoriser OFF, so that the
    compiler packs nothing by itself: every other kernel of the file must stay free of v_pk_{fma,add,mul}_f32 -- those
    kernels run beside split-bf16 MFMAs, where packed ops returned wrong values."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "xumx_slicq_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "-fno-slp-vectorize" in mk and "SLICQT_FLAGS" in mk
    asm = str(tmp_path / "slicqt.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-fno-slp-vectorize",
                    "-S", "--cuda-device-only", os.path.join(csrc, "slicqt.hip"), "-o", asm], check=True, capture_output=True, timeout=900)
    counts, cur = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        elif cur and re.search(r"v_pk_(fma|add|mul)_f32", line):
            counts[cur] = counts.get(cur, 0) + 1
    assert counts, "the hand-packed kernels were not generated"
    for name, n in counts.items():
        assert ("k_slice_rfftILi512ELb1" in name or "k_slice_irfftILi512ELb1" in name) and n > 1000, (name, n)
    # the other device files keep the feature off altogether
    assert "NOPK" in mk and "-packed-fp32-ops" in mk
