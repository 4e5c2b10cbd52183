"""Vector-instruction budget of the radix-4 band kernel, read off the gfx950 assembly (hipcc cross-compiles without a GPU).

On this part an fp32 MFMA and the other waves' vector instructions take turns on a SIMD ("Vector
issue"): the kernel's K-step costs its MFMA cycles PLUS ~4 cycles per other vector instruction, so those are a budget.
tools/valu_mfma.py counts them per MFMA loop; this test holds the counts the buffer-addressed kernel reached (108 -> 55 per
K-step of the masked synthesis) and that the instantiations stay free of scratch and inside three workgroups per CU."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
SRC = """#include "band_dft4.h"
using namespace xsq;
template __global__ void xsq::band_dft4_full_kernel<true, 10, false>(Band4Args, const Tile4Dev*, int);
template __global__ void xsq::band_dft4_full_kernel<false, 10, true>(Band4Args, const Tile4Dev*, int);
template __global__ void xsq::band_dft4_full_kernel<false, 10, false>(Band4Args, const Tile4Dev*, int);
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_band_kernel_vector_instruction_budget(tmp_path):
    src, asm = tmp_path / "d4only.hip", tmp_path / "d4only.s"
    src.write_text(SRC)
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "xumx_slicq_amd", "csrc"), "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    str(src), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    # registers / scratch from the kernel descriptors' metadata
    meta = re.findall(r"\.name:\s+(\S+)\s+\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, flags=re.S)
    assert len(meta) == 3, meta
    for name, scratch, vgprs in meta:
        assert int(scratch) == 0, (name, scratch)
        assert int(vgprs) <= 168, (name, vgprs)          # three waves per SIMD (amdgpu_waves_per_eu(3, 3))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_mfma.py"), str(asm)], check=True,
                         capture_output=True, text=True, timeout=120).stdout
    rows = {}
    cur = None
    for line in out.splitlines():
        if line.startswith("void "):
            cur = line
        m = re.search(r"MFMA\s+(\d+) cycles \(\s*(\d+)\), other vector\s+(\d+)", line)
        if m and cur:
            rows.setdefault(cur, []).append(tuple(int(x) for x in m.groups()))
    inv_masked = [v for k, v in rows.items() if "<false, 10, true>" in k][0]
    inv_plain = [v for k, v in rows.items() if "<false, 10, false>" in k][0]
    fwd = [v for k, v in rows.items() if "<true, 10, false>" in k][0]
    # the K loop: 80 MFMAs (ten 16-column blocks x 8) = 2560 cycles; other vector instructions per K-step
    for loops, budget in ((inv_masked, 64), (inv_plain, 56), (fwd, 140)):      # (fwd: sum over its two exclusive load paths)
        k_loop = min(loops, key=lambda r: r[2])
        assert k_loop[0] == 2560 and k_loop[1] == 80, loops
        assert k_loop[2] <= budget, (loops, budget)
    assert "v_pk_" not in text          # no packed-fp32 ops (Makefile NOPK; -fno-slp-vectorize stands in for it here)


SRC_S = """#include "band_dft4s.h"
using namespace xsq;
template __global__ void xsq::band_dft4s_kernel<true, false>(Band4Args, const Tile4Dev*, int);
template __global__ void xsq::band_dft4s_kernel<false, true>(Band4Args, const Tile4Dev*, int);
template __global__ void xsq::band_dft4s_kernel<false, false>(Band4Args, const Tile4Dev*, int);
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_pair_contracted_band_kernel_budget(tmp_path):
    """band_dft4s.h (the default band kernel: m-point DFTs contracted over input pairs): no scratch, inside three workgroups
    per CU (168 registers, 42.1 KB of LDS), 48 MFMAs = 1,536 cycles per K-step of the three-block form, and the synthesis
    K-step within its vector-instruction budget."""
    src, asm = tmp_path / "d4s.hip", tmp_path / "d4s.s"
    src.write_text(SRC_S)
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "xumx_slicq_amd", "csrc"), "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                    "-S", "--cuda-device-only", str(src), "-o", str(asm)], check=True, capture_output=True, timeout=600)      # (the library's flags: csrc/Makefile NOPK)
    text = asm.read_text()
    meta = re.findall(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+)\s+\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, flags=re.S)
    assert len(meta) == 3, meta
    for lds, name, scratch, vgprs in meta:
        assert int(scratch) == 0 and int(vgprs) <= 168 and 3 * int(lds) <= 160 * 1024, (name, scratch, vgprs, lds)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_mfma.py"), str(asm)], check=True,
                         capture_output=True, text=True, timeout=120).stdout
    rows, cur = {}, None
    for line in out.splitlines():
        if line.startswith("void "):
            cur = line
        m = re.search(r"MFMA\s+(\d+) cycles \(\s*(\d+)\), other vector\s+(\d+)", line)
        if m and cur:
            rows.setdefault(cur, []).append(tuple(int(x) for x in m.groups()))
    for sig, budget in (("<false, true>", 120), ("<false, false>", 105)):
        loops = [v for k, v in rows.items() if "band_dft4s_kernel" + sig in k][0]
        k_loop = min(loops, key=lambda r: r[2])
        assert k_loop[0] == 1536 and k_loop[1] == 48 and k_loop[2] <= budget, (sig, loops)
    assert "v_pk_" not in text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_committed_instruction_budget_is_the_one_of_the_sources(tmp_path):
    """profiles/isa_budget.json (what bench.py's `roofline_issue` is computed from) against a fresh run of tools/isa_budget.py
    over the current sources: the MFMA cycles and other-vector-instruction counts of every kernel's loops, exactly -- a kernel
    edited without regenerating the budget fails here.  Also holds the Winograd kernels' per-tap budget (195 MFMAs = 6,240
    cycles) and that they carry no scratch beyond a few spilled words outside the chunk loop."""
    import json
    out = tmp_path / "budget.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_budget.py"), str(out)], check=True, capture_output=True, timeout=1500)
    new = json.load(open(out))["kernels"]
    old = json.load(open(os.path.join(ROOT, "profiles", "isa_budget.json")))["kernels"]
    assert sorted(new) == sorted(old)
    for k in new:
        a = [(l["mfma_cycles"], l["nmfma"], l["valu"]) for l in new[k]["loops"]]
        b = [(l["mfma_cycles"], l["nmfma"], l["valu"]) for l in old[k]["loops"]]
        assert a == b and new[k]["outside"] == old[k]["outside"], (k, a, b)
    for k in ("cdae_wino<L2>", "cdae_wino<L3>"):
        (lp,) = new[k]["loops"]
        assert lp["mfma_cycles"] == 6240 and lp["nmfma"] == 195 and lp["valu"] < 420, lp
