#!/usr/bin/env python3
"""Vector-issue budget of the loops that hold MFMAs, from the device assembly (no GPU needed).

On gfx950 the fp32 MFMAs (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) run at exactly the packed-fp32 rate of the SIMD's vector
ALU and, measured ("Vector issue"), do not overlap with other waves' vector instructions: a loop's
time on a SIMD is the SUM of its MFMA cycles and 4 cycles per other vector instruction (a wave64 op on 16 lanes), not
their maximum.  This lists, per kernel and per loop that contains MFMAs, the MFMA cycles, the other vector instructions
(VALU, v_accvgpr_*, DPP), LDS and memory instructions of one trip through the loop's blocks (all blocks of the loop
summed: an upper bound where the loop has exclusive branches) and the share of the vector port the MFMAs can reach.

usage: tools/valu_mfma.py file.s [name-filter]     (hipcc -S --cuda-device-only ... -o file.s)"""
import re
import subprocess
import sys

PASSES = {"32x32x2": 16, "16x16x4": 8, "32x32x1": 16, "16x16x1": 8, "4x4x1": 2,       # fp32
          "32x32x16": 8, "16x16x32": 4, "32x32x8": 8, "16x16x16": 4, "32x32x4": 16}   # bf16 / f16 (gfx950); passes of 4 cycles


def mfma_cycles(op):
    m = re.search(r"(\d+x\d+x\d+)", op)
    return 4 * PASSES.get(m.group(1), 8) if m else 32


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    fn, loops, cur = None, {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn, loops, cur = m.group(1), {}, None
            continue
        if fn is None:
            continue
        if ".Lfunc_end" in l:
            if want in fn:
                rows = [(h, d) for h, d in loops.items() if d.get("mfma", 0)]
                if rows:
                    name = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()
                    print(name[:150])
                    for h, d in rows:
                        v = 4 * d.get("valu", 0)
                        print("   loop %-10s depth %d: MFMA %5d cycles (%3d), other vector %4d (%5d cycles), LDS %3d, memory %3d, scalar %3d"
                              "  -> MFMA share of the vector port <= %.2f" % (h, d["depth"], d["mfma"], d["nmfma"], d.get("valu", 0), v,
                                                                               d.get("lds", 0), d.get("vmem", 0), d.get("salu", 0), d["mfma"] / (d["mfma"] + v)))
            fn = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", l)
        if m:
            lab, note = m.groups()
            cur = None
            mm = re.search(r"Loop Header: Depth=(\d+)", note)
            if mm:
                cur = (lab.replace(".L", ""), int(mm.group(1)))
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if mm and not cur:
                cur = (mm.group(1), int(mm.group(2)))
            if "Inner Loop Header" in note:
                mm = re.search(r"Depth=(\d+)", note)
                cur = (lab.replace(".L", ""), int(mm.group(1)))
            if cur:
                loops.setdefault(cur[0], {"depth": cur[1], "nmfma": 0})
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = None
            continue
        t = l.strip().split()
        if not t or cur is None or t[0].startswith((";", ".")):
            continue
        op, d = t[0], loops[cur[0]]
        if op.startswith("v_mfma"):
            d["mfma"] = d.get("mfma", 0) + mfma_cycles(op)
            d["nmfma"] += 1
        elif op.startswith("v_"):
            d["valu"] = d.get("valu", 0) + 1
        elif op.startswith("ds_"):
            d["lds"] = d.get("lds", 0) + 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            d["vmem"] = d.get("vmem", 0) + 1
        elif op.startswith("s_"):
            d["salu"] = d.get("salu", 0) + 1


main()
