# A/B of diagnostic environment switches on the headline bench (one process per setting; differences below ~2 % are noise).
# usage: tools/ab_env.sh OUTDIR "NAME=ENV..." ...      e.g.  tools/ab_env.sh gpurun_out/ab "base=" "late=XSQ_CDAE_VARIANT=512"
O=$1; shift
mkdir -p $O
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $envs python bench.py --steps 20 --warmup 3 --no-variants --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels"]
    print(sys.argv[2], d["ms_per_step"], " ".join(f"{n}={v['ms_per_step']:.3f}" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
