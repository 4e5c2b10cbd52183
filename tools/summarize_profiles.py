#!/usr/bin/env python3
"""Condense a tools/collect_profiles.sh output directory (gpurun_out/prof_<round>) into the
small summaries committed under profiles/: per-kernel rocprofv3 --stats table, SQ counters,
and HBM traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for
gfx950 wide coalesced reads; WRITE_SIZE as is; both in bytes per launch)."""
import glob
import json
import os
import sys

import pandas as pd

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
pick = max if (len(sys.argv) < 5 or sys.argv[4] == "newest") else min   # several runs may share the directory
os.makedirs(dst, exist_ok=True)
_glob = glob.glob


def newest(pattern):
    return [pick(_glob(pattern), key=os.path.getmtime)]


glob.glob = newest


def short(n):
    import re
    n = n.replace("void xsq::band_dft4_kernel<false>", "band_dft4<inverse>").replace("void xsq::band_dft4_kernel<true>", "band_dft4<forward>")    # profiles before r03e
    n = re.sub(r"void xsq::band_dft4s_kernel<false(?:, (?:true|false))?>", "band_dft4s<inverse>", n)       # pair-contracted form (band_dft4s.h)
    n = re.sub(r"void xsq::band_dft4s_kernel<true(?:, (?:true|false))?>", "band_dft4s<forward>", n)
    n = re.sub(r"void xsq::band_dft4_full_kernel<false(?:, \d+)?(?:, (?:true|false))?>", "band_dft4<inverse>", n)
    n = re.sub(r"void xsq::band_dft4_full_kernel<true(?:, \d+)?(?:, (?:true|false))?>", "band_dft4<forward>", n)
    n = re.sub(r"void xsq::k_slice_(i?rfft)<\d+(?:, (?:true|false))*>", r"k_slice_\1", n)
    m = re.match(r"void xsq::cdae_slab_kernel<(true|false), (\d)(?:, (?:true|false))?>", n)
    if m:
        return {"0": "slab", "3": "slab", "1": "slab_bf3", "2": "slab_bf6"}[m.group(2)] + ("<CdaeL3>" if m.group(1) == "true" else "<CdaeL2>")
    m = re.match(r"void xsq::cdae_wino_kernel<(true|false)>", n)
    if m:
        return "wino" + ("<CdaeL3>" if m.group(1) == "true" else "<CdaeL2>")
    if "cdae_l1f_kernel" in n:                 # layers 1 / 4 as F(2, 2) along the hop (cdae_l1f.h, cdae_l4f.h: round 6)
        return "l1f<CdaeL1>"
    if "cdae_l4f_kernel" in n:
        return "l4f<CdaeL4>"
    n = re.sub(r"void xsq::grouped_gemm_bf6_kernel<xsq::(\w+)>", r"gemm_bf6<\1>", n)
    n = re.sub(r"void xsq::grouped_gemm_bf3_kernel<xsq::(\w+), \d+, \d+>", r"gemm_bf3<\1>", n)
    n = re.sub(r"void xsq::grouped_gemm_kernel<xsq::(\w+)(?:, \d+)*>", r"gemm<\1>", n)
    n = n.replace("void xsq::grouped_gemm_kernel<xsq::", "gemm<").replace("xsq::", "")
    return n.split("(")[0][:48]


ours = "gemm|slab|wino|l1f|l4f|band_dft4|k_|fft|bluestein|c2r|r2c"
st = pd.read_csv(glob.glob(f"{src}/trace/*/*kernel_stats.csv")[0])
st["Kernel"] = st["Name"].map(short)
st = st[["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"]]
st.to_csv(f"{dst}/{tag}_kernel_stats.csv", index=False)

out = {}
sq = pd.read_csv(glob.glob(f"{src}/pmc_sq/*/*counter_collection.csv")[0])
sq["Kernel"] = sq["Kernel_Name"].map(short)
p = sq[sq.Kernel.str.contains(ours)].pivot_table(index="Kernel", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
meta = sq.groupby("Kernel")[["VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Scratch_Size"]].max()
p = p.join(meta)
p["wait_any_frac"] = p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"]
p["wait_inst_frac"] = p["SQ_WAIT_INST_ANY"] / p["SQ_WAVE_CYCLES"]
p.round(4).to_csv(f"{dst}/{tag}_pmc_sq.csv")

rows = []
for name, corr in (("fetch", 2.0), ("write", 1.0)):
    d = pd.read_csv(glob.glob(f"{src}/pmc_{name}/*/*counter_collection.csv")[0])
    d["Kernel"] = d["Kernel_Name"].map(short)
    d = d[d.Kernel.str.contains(ours)]
    g = d.groupby("Kernel")["Counter_Value"].agg(["mean", "max", "count"])
    g.columns = [f"{name}_KB_mean_raw", f"{name}_KB_max_raw", "launches"]
    g[f"{name}_MB_largest_launch"] = g[f"{name}_KB_max_raw"] * corr / 1024.0
    rows.append(g)
t = rows[0].join(rows[1], lsuffix="", rsuffix="_w")
t["steps_profiled"] = int(os.environ.get("XSQ_PROFILED_STEPS", "2"))     # collect_profiles.sh: --steps 1 --warmup 1
t.round(3).to_csv(f"{dst}/{tag}_hbm_traffic.csv")
for f in ("trace_bench.json", "pmc_sq_bench.json"):
    try:
        line = open(f"{src}/{f}").read().strip().splitlines()[-1]
        json.loads(line)
        open(f"{dst}/{tag}_{f}", "w").write(line + "\n")
    except Exception as e:  # noqa: BLE001
        print("skip", f, e)
print(st.head(14).to_string())
print(t.round(1).to_string())
