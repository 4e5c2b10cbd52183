# rocprofv3 kernel statistics of the training step (tools/bench_train.py).  usage (GPU box): tools/prof_train.sh OUTFILE [bench_train args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}      # (resolved before the cd: the scripts run from /tmp)
cd /tmp && export TMPDIR=/tmp
O=$R/$1; shift
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -- python3 $R/tools/bench_train.py --steps 20 --warmup 5 --no-profile "$@" > /tmp/tr.log 2> /tmp/tr.err
tail -1 /tmp/tr.log | cut -c1-200
python3 - "$O" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/prof_tr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 25.0
with open(sys.argv[1], "w") as o:
    tot = 0.0
    for r in rows:
        ms = float(r["TotalDurationNs"]) / steps / 1e6
        tot += ms
        if r["Name"].startswith("__amd_rocclr"): continue      # one-time uploads of model / trainer construction
        name = re.sub(r"\(.*", "", r["Name"]).replace("void xsq::", "")[:90]
        line = "%-92s calls/step %5.1f  ms/step %.4f" % (name, float(r["Calls"]) / steps, ms)
        o.write(line + "\n")
    o.write("total kernel time, ms per step: %.3f\n" % tot)
print(open(sys.argv[1]).read()[:3600])
PY
