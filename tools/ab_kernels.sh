# Diagnostic: A/B of build-flag variants of the library on the GPU box.  tools/ab_kernels.sh OUTFILE PATTERN "NAME=EXTRA FLAGS" ...
# builds each variant out of tree (make EXTRA=...), runs bench.py with it (XSQ_LIB) and prints ms per step + the kernels whose
# event names match PATTERN.  NAME=base with empty flags gives the product build on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; P=$2; shift; shift
: > $O
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C $R/xumx_slicq_amd/csrc -j16 OBJDIR=/tmp/ab_$name OUT=/tmp/libab_$name.so "EXTRA=$flags" 2>/dev/null >/dev/null
  XSQ_LIB=/tmp/libab_$name.so python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>/tmp/ab_err.txt > /tmp/ab_out.json || tail -3 /tmp/ab_err.txt
  python3 -c "
import json, re
d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]); k=d['kernels']
print('$name', '[$flags]', 'ms/step', d['ms_per_step'], {n: round(v['ms_per_step'],3) for n,v in k.items() if re.search('$P', n)})" | tee -a $O
done
