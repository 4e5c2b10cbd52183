"""Regression guard for the packed-fp32 / MFMA co-residency finding (csrc/Makefile NOPK): the
standalone probe tools/probe/pk_mfma_hazard.hip, built with the PRODUCT flags (no packed-fp32 ops), must return the
bits of the transform run alone next to every MFMA aggressor.  The packed build of the same probe is run as well and
its outcome printed (on MI355X / ROCm 7.2 it differs in 20 of 20 runs next to v_mfma_f32_16x16x32_bf16); it is
reported, not asserted: a toolchain or firmware that fixes it should not fail the suite."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    cmd = [HIPCC, "-O3", "-w", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "xumx_slicq_amd", "csrc"),
           *extra, os.path.join(ROOT, "tools", "probe", "pk_mfma_hazard.hip"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return exe


def _run(exe):
    out = subprocess.run([exe], check=True, capture_output=True, timeout=600, text=True).stdout
    return [(m.group(1).strip(), int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"^(.*?):\s+(\d+) / (\d+) ", out, flags=re.M)], out


def test_product_flags_are_exact_next_to_every_mfma_aggressor(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this box")
    rows, out = _run(_build(tmp_path, "pk_off", ["-DPROBE_NO_PK", *NOPK]))
    assert len(rows) >= 8, out
    assert all(bad == 0 for _, bad, _ in rows), out
    try:        # the packed build: reported only
        rows_on, out_on = _run(_build(tmp_path, "pk_on", []))
        print("\npacked-fp32 build of the probe:\n" + out_on)
    except Exception as e:      # noqa: BLE001
        print("packed build of the probe did not run:", e)


def test_hand_packed_transform_is_exact_next_to_the_fp32_aggressors(tmp_path):
    """The hand-placed v_pk_* butterflies of k_slice_rfft<512, true> (slice_fft.h) are launched only beside fp32 work.
    Probe built like csrc/slicqt.o (packed ops on, SLP vectoriser off): the scalar kernel is the control and must be exact
    beside EVERY aggressor; the packed kernel must be exact beside the aggressors of the fp32 path -- v_mfma_f32_32x32x2_f32,
    v_mfma_f32_16x16x4_f32, plain VALU, nothing -- and bitwise equal to the scalar kernel.  What it does beside the split-bf16
    MFMAs is printed, not asserted (the Separator never pairs them: test_packed_slice_fft_follows_the_contraction_mode)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this box")
    rows, out = _run(_build(tmp_path, "pk_hand", ["-fno-slp-vectorize", "-DPROBE_HAND_PK"]))
    print("\n" + out)
    assert "bitwise equal" in out and "DIFFERENT BITS" not in out, out
    per_victim = 6
    assert len(rows) >= 2 * per_victim, out
    scalar, packed = rows[:per_victim], rows[per_victim:2 * per_victim]
    assert all(bad == 0 for _, bad, _ in scalar), out
    fp32_side = ("v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "plain VALU (no MFMA)", "none")
    for name, bad, _ in packed:
        if name.replace("aggressor ", "").strip() in fp32_side:
            assert bad == 0, out


def test_buffer_range_check_covers_the_scalar_offset(tmp_path):
    """common.h's buffer loads / stores (layer-1 / layer-4 operands, the wide-store epilogues, band_dft4.h) switch rows and
    columns off by an out-of-range offset and put tile displacements into the SCALAR offset: that is only right if the
    hardware checks voffset + soffset against num_records as one sum, returns 0 for loads past it and drops stores there.
    tools/probe/buf_range.hip asserts exactly that on this device (incl. soffset alone past the range)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "buf_range")
    subprocess.run([HIPCC, "-O2", "-w", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "probe", "buf_range.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, timeout=120, text=True)
    print("\n" + r.stdout)
    assert r.returncode == 0 and "ALL AS RELIED ON" in r.stdout and "UNEXPECTED" not in r.stdout, r.stdout
