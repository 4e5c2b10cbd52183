#!/usr/bin/env python3
"""Per-kernel vector-issue budget of the fp32 MFMA kernels, read off the gfx950 assembly (no GPU needed: hipcc cross-compiles).

On gfx950 an fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 cycles, 16x16x4_f32: 32) and the other vector instructions of the
SIMD's waves take turns on one issue port ("Vector issue": measured, a loop's time is the SUM of its MFMA cycles
and ~4 cycles per other vector instruction, not their maximum).  So the fastest a kernel can run is
    sum over tiles and waves of (MFMA cycles + 4 x other vector instructions)  /  (1024 SIMDs x clock)
-- its ISSUE BOUND.  This tool writes what bench.py needs to evaluate that bound per launch (`roofline_issue`): for every
MFMA kernel of the product library the MFMA cycles and other vector instructions (VALU, DPP, v_accvgpr moves; not LDS / memory
/ scalar) of (a) each loop that holds MFMAs, per trip, and (b) everything outside those loops (prologue + epilogue).

    python3 tools/isa_budget.py [out.json]   -> profiles/isa_budget.json   (committed; tests/test_isa_budget_cpu.py holds it to the sources)
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xumx_slicq_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
PASSES = {"32x32x2": 16, "16x16x4": 8, "32x32x1": 16, "16x16x1": 8, "4x4x1": 2,
          "32x32x16": 8, "16x16x32": 4, "32x32x8": 8, "16x16x16": 4, "32x32x4": 16}
# kernels of the fp32 inference path (event name of bench.py -> demangled-name filter)
WANT = {
    "cdae_wino<L2>": r"cdae_wino_kernel<false>", "cdae_wino<L3>": r"cdae_wino_kernel<true>",
    "cdae_l1f": r"cdae_l1f_kernel",
    "cdae_slab<L2>": r"cdae_slab_kernel<false, 3, true>", "cdae_slab<L3>": r"cdae_slab_kernel<true, 3, true>",
    "gemm<CdaeL1Op>": r"grouped_gemm_kernel<xsq::CdaeL1Op, 1, 1>", "gemm<CdaeL2Op>": r"grouped_gemm_kernel<xsq::CdaeL2Op, 1, 1>",
    "gemm<CdaeL3Op>": r"grouped_gemm_kernel<xsq::CdaeL3Op, 1, 1>", "gemm<CdaeL4Op>": r"grouped_gemm_kernel<xsq::CdaeL4Op, 1, 2>",
    "band_dft4<forward>": r"band_dft4_full_kernel<true, 10, false>", "band_dft4<inverse,masked>": r"band_dft4_full_kernel<false, 10, true>",
    "band_dft4<inverse>": r"band_dft4_full_kernel<false, 10, false>",
    "band_dft4s<forward>": r"band_dft4s_kernel<true, false>", "band_dft4s<inverse,masked>": r"band_dft4s_kernel<false, true>",
    "band_dft4s<inverse>": r"band_dft4s_kernel<false, false>",
    "gemm<BandFwdOp>": r"grouped_gemm_kernel<xsq::BandFwdOp, 1, 0>", "gemm<BandInvOp>": r"grouped_gemm_kernel<xsq::BandInvOp, 1, 0>",
}
VECTOR = re.compile(r"^\s*v_")
NOT_ISSUE = re.compile(r"^\s*v_(mfma|nop)")


def mfma_cycles(op):
    m = re.search(r"(\d+x\d+x\d+)", op)
    return 4 * PASSES.get(m.group(1), 8) if m else 32


def compile_asm(src, out):
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", "-S", "--cuda-device-only"]
    subprocess.run([HIPCC] + flags + [os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, timeout=1800)


def parse(path):
    """{mangled: {"outside": {...}, "loops": [{"header", "depth", "mfma_cycles", "nmfma", "valu"}]}}; an instruction belongs to
    the innermost loop its basic block is in."""
    funcs, fn, cur, d = {}, None, None, None
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn, cur = m.group(1), None
            d = funcs[fn] = {"outside": {"mfma_cycles": 0, "nmfma": 0, "valu": 0}, "loops": {}}
            continue
        if fn is None:
            continue
        if ".Lfunc_end" in l:
            fn = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)$", l)
        if m:
            note = m.group(2)
            cur = None
            mm = re.search(r"Loop Header: Depth=(\d+)", note)
            if mm:
                cur = (m.group(1).replace(".L", ""), int(mm.group(1)))
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if mm and not cur:
                cur = (mm.group(1), int(mm.group(2)))
            if cur:
                d["loops"].setdefault(cur[0], {"header": cur[0], "depth": cur[1], "mfma_cycles": 0, "nmfma": 0, "valu": 0})
            continue
        if l.startswith(".L") and ":" in l:          # a plain block label outside any loop
            cur = None
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        tgt = d["loops"][cur[0]] if cur else d["outside"]
        if s.startswith("v_mfma"):
            tgt["mfma_cycles"] += mfma_cycles(s.split()[0])
            tgt["nmfma"] += 1
        elif VECTOR.match(s) and not NOT_ISSUE.match(s):
            tgt["valu"] += 1
    return funcs


L4F_PROBE = r"""
// tools/isa_budget.py: the four column-width bodies of cdae_l4f_kernel as kernels of their own (the product kernel holds all
// four behind a uniform switch: a static count over it cannot tell which one runs)
#include "cdae_l4f.h"
namespace xsq {
template <int NCB>
__global__ __launch_bounds__(256, XSQ_L4F_WAVES_PER_EU) void l4f_body_probe(CdaeArgs a, const L4fTileDev* __restrict__ tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) float Bs[L4_BROWS * L4_BLD];
    const L4fTileDev t = tiles[xcd_remap(blockIdx.x, ntiles)];
    asm volatile("" :: "s"(t.Q0), "s"(t.kf), "s"(t.F), "s"(t.run), "s"(t.in_off), "s"(t.out_off), "s"(t.bias_off), "s"(t.u_off),
                 "s"(t.hop), "s"(t.n0), "s"(t.P), "s"(t.x_off));
    cdae_l4f_body<NCB, false>(a, t, Bs);
}
template __global__ void l4f_body_probe<1>(CdaeArgs, const L4fTileDev*, int);
template __global__ void l4f_body_probe<2>(CdaeArgs, const L4fTileDev*, int);
template __global__ void l4f_body_probe<3>(CdaeArgs, const L4fTileDev*, int);
template __global__ void l4f_body_probe<4>(CdaeArgs, const L4fTileDev*, int);
}
"""


def l4f_bodies():
    """{"cdae_l4f<NCB>": budget} from a probe translation unit (see L4F_PROBE)."""
    src = os.path.join(CSRC, "_isa_budget_l4f_probe.hip")
    open(src, "w").write(L4F_PROBE)
    try:
        compile_asm("_isa_budget_l4f_probe.hip", "/tmp/isa_budget_l4f.s")
    finally:
        os.remove(src)
    funcs = parse("/tmp/isa_budget_l4f.s")
    names = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, dem in zip(funcs, names):
        m = re.search(r"l4f_body_probe<(\d)>", dem)
        if not m:
            continue
        d = funcs[mangled]
        loops = [v for v in d["loops"].values() if v["nmfma"]]
        extra = sum(v["valu"] for v in d["loops"].values() if not v["nmfma"])
        out["cdae_l4f<%s>" % m.group(1)] = {"symbol": "cdae_l4f_body<%s> (probe kernel of tools/isa_budget.py)" % m.group(1),
                                            "outside": dict(d["outside"], valu=d["outside"]["valu"] + extra),
                                            "loops": sorted(loops, key=lambda v: v["mfma_cycles"])}
    return out


def main():
    out = {"what": "MFMA cycles and other vector instructions (4 issue cycles each) per loop trip and outside the MFMA loops, per wave; "
                   "tools/isa_budget.py from the gfx950 assembly of csrc/cdae.hip and csrc/slicqt.hip", "kernels": {}}
    for src in ("cdae.hip", "slicqt.hip"):
        asm = "/tmp/isa_budget_%s.s" % src.split(".")[0]
        compile_asm(src, asm)
        funcs = parse(asm)
        names = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.split("\n")
        for mangled, dem in zip(funcs, names):
            for key, pat in WANT.items():
                if re.search(re.escape(pat), dem):
                    d = funcs[mangled]
                    loops = [v for v in d["loops"].values() if v["nmfma"]]
                    # vector instructions of loops without MFMAs stay unattributed to a trip count: count them once, as outside
                    extra = sum(v["valu"] for v in d["loops"].values() if not v["nmfma"])
                    outside = dict(d["outside"], valu=d["outside"]["valu"] + extra)
                    out["kernels"][key] = {"symbol": dem.split("(")[0], "outside": outside, "loops": sorted(loops, key=lambda v: v["mfma_cycles"])}
    out["kernels"].update(l4f_bodies())
    missing = [k for k in list(WANT) + ["cdae_l4f<%d>" % i for i in (1, 2, 3, 4)] if k not in out["kernels"]]
    if missing:
        sys.exit("isa_budget: kernels not found in the assembly: %s" % missing)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_budget.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-28s outside: MFMA %5d cyc, vector %4d | " % (k, v["outside"]["mfma_cycles"], v["outside"]["valu"]) +
              "; ".join("loop %s: MFMA %5d cyc (%3d), vector %4d" % (l["header"], l["mfma_cycles"], l["nmfma"], l["valu"]) for l in v["loops"]))
    print(path)


if __name__ == "__main__":
    main()
