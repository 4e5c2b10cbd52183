#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace run of bench.py: per kernel start / end relative to
the step's first launch, the idle gaps between consecutive kernels of the main stream, and what the tail stream's kernels
overlap.  usage: tools/timeline.py <dir with *_kernel_trace.csv> [steps_from_end]  -> prints a table + totals."""
import glob
import sys

import pandas as pd

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
src = sys.argv[1]
f = max(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getsize(p))
d = pd.read_csv(f)
d = d.sort_values("Start_Timestamp").reset_index(drop=True)
name_col = "Kernel_Name"


def short(n):
    import re
    n = re.sub(r"void xsq::band_dft4_full_kernel<false.*", "band_synthesis_dft4", n)
    n = re.sub(r"void xsq::band_dft4_full_kernel<true.*", "band_analysis_dft4", n)
    n = re.sub(r"void xsq::k_slice_(i?rfft).*", r"slice_\1", n)
    m = re.match(r"void xsq::cdae_slab_kernel<(true|false).*", n)
    if m:
        return "cdae_l3_slab" if m.group(1) == "true" else "cdae_l2_slab"
    n = re.sub(r"void xsq::grouped_gemm_kernel<xsq::(\w+).*", r"gemm<\1>", n)
    return n.split("(")[0][:40]


d["k"] = d[name_col].map(short)
train = len(sys.argv) > 3 and sys.argv[3] == "train"
if train:       # a training step ends with its AdamW launch
    ours = d[~d.k.str.contains("rocclr|fill|copy", case=False)].reset_index(drop=True)
    steps = [i + 1 for i in ours.index[ours.k.str.contains("k_adamw")].tolist()]
else:
    ours = d[d.k.str.contains("gemm|slab|band_|slice_")].reset_index(drop=True)
    # a step starts at the tail pass's slice_rfft (issued first): two slice_rfft per step (tail, stacked)
    starts = ours.index[ours.k == "slice_rfft"].tolist()
    steps = [starts[i] for i in range(0, len(starts), 2)]
n_back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i0, i1 = steps[-n_back - 1], steps[-n_back]
s = ours.iloc[i0:i1].copy()
t0 = s.Start_Timestamp.min()
s["start_us"] = (s.Start_Timestamp - t0) / 1e3
s["end_us"] = (s.End_Timestamp - t0) / 1e3
s["dur_us"] = s.end_us - s.start_us
qcol = "Queue_Id" if "Queue_Id" in s.columns else None
print(s[["k", "start_us", "end_us", "dur_us"] + ([qcol] if qcol else [])].to_string(index=False))
# union of busy intervals vs span
iv = sorted(zip(s.start_us, s.end_us))
busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
span = max(e for _, e in iv) - iv[0][0]
nxt = ours.iloc[i1].Start_Timestamp
print(f"step span {span:.1f} us, union of kernel intervals {busy:.1f} us, idle inside the span {span - busy:.1f} us, "
      f"sum of durations {s.dur_us.sum():.1f} us, next step's first launch at {(nxt - t0) / 1e3:.1f} us")
