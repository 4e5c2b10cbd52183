# One SQ counter pass over the headline bench (2 steps) and a per-kernel table: matrix-pipe busy, vector-ALU issue utilisation,
# share of wave cycles in s_waitcnt, LDS bank-conflict cycles per LDS-active... for the kernels matching FILTER (regex).
# usage (GPU box): tools/pmc_quick.sh OUTDIR [FILTER]     (XSQ_LIB / XSQ_CDAE_VARIANT etc. pass through)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}      # (resolved before the cd: the scripts run from /tmp)
cd /tmp && export TMPDIR=/tmp
O=$R/$1; F=${2:-wino|slab}
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmcq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc.err
python3 - "$O" "$F" <<'PY'
import glob, sys, re
import pandas as pd
f = max(glob.glob("/tmp/pmcq/*/*counter_collection.csv"), key=lambda p: __import__("os").path.getmtime(p))
d = pd.read_csv(f)
d = d[d.Kernel_Name.str.contains(sys.argv[2])]
d["K"] = d.Kernel_Name.str.replace("void xsq::", "").str.slice(0, 44)
# the largest dispatch of each kernel (the stacked pass)
big = d[d.Counter_Name == "SQ_WAVE_CYCLES"].sort_values("Counter_Value").groupby("K").tail(1)[["K", "Dispatch_Id"]]
d = d.merge(big, on=["K", "Dispatch_Id"])
p = d.pivot_table(index="K", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
p["mfma_busy"] = p.SQ_VALU_MFMA_BUSY_CYCLES / (32.0 * p.SQ_BUSY_CYCLES)
p["valu_issue"] = p.SQ_INSTS_VALU / (8.0 * p.SQ_BUSY_CYCLES)
p["wait_share"] = p.SQ_WAIT_INST_ANY / p.SQ_WAVE_CYCLES
p["lds_wait_share"] = p.SQ_WAIT_INST_LDS / p.SQ_WAVE_CYCLES
p["bank_conflict_per_lds_inst"] = p.SQ_LDS_BANK_CONFLICT / p.SQ_INSTS_LDS
p["waves_per_simd"] = p.SQ_WAVE_CYCLES / (4.0 * 8.0 * p.SQ_BUSY_CYCLES) * 4
out = p[["mfma_busy", "valu_issue", "wait_share", "lds_wait_share", "bank_conflict_per_lds_inst", "waves_per_simd", "SQ_BUSY_CYCLES"]].round(3)
print(out.to_string())
out.to_csv(sys.argv[1] + "/pmc_quick.csv")
PY
