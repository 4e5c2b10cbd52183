#!/usr/bin/env python3
"""Scan the device assembly of the library for memory operations that serialise an epilogue: per kernel, the
number of global stores, and the loads / `s_waitcnt vmcnt(0)` that follow the FIRST store.  On gfx950 loads and
stores share one in-order counter, so a load issued between two stores makes the next wait cover the store as well
("Reading the instruction stream").

usage: tools/scan_isa.py [file.hip ...]      (default: every .hip of xumx_slicq_amd/csrc; needs hipcc, no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xumx_slicq_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S", "--cuda-device-only"]


def scan(asm):
    rows, cur = [], None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = [m.group(1), 0, 0, 0, False]
            rows.append(cur)
            continue
        if cur is None:
            continue
        if "s_endpgm" in line:
            cur = None
        elif "global_store" in line or "buffer_store" in line:
            cur[1] += 1
            cur[4] = True
        elif cur[4] and ("global_load" in line or "buffer_load" in line):
            cur[2] += 1
        elif cur[4] and "s_waitcnt vmcnt(0)" in line:
            cur[3] += 1
    return rows


def main():
    files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, f, "-o", tmp.name], check=True, cwd=CSRC,
                           stderr=subprocess.DEVNULL)
            for name, stores, loads, waits, _ in scan(tmp.name):
                if stores:
                    pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    print(f"{stores:4d} stores  {loads:4d} loads after the first  {waits:3d} vmcnt(0) after the first  {pretty[:120]}")


if __name__ == "__main__":
    main()
