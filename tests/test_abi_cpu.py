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
