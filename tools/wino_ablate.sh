# Diagnostic: ablated builds of the Winograd kernels (csrc/cdae_wino.h, XSQ_WINO_ABL; results are wrong by construction, only
# the timings matter) and its occupancy variants.  Run on the GPU box:  tools/wino_ablate.sh OUTFILE "NAME=EXTRA FLAGS" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; shift
: > $O
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C $R/xumx_slicq_amd/csrc -j16 OBJDIR=/tmp/wn_$name OUT=/tmp/libwn_$name.so "EXTRA=$flags" 2>/dev/null >/dev/null
  XSQ_LIB=/tmp/libwn_$name.so python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants 2>/tmp/wn_err.txt > /tmp/wn_out.json || tail -3 /tmp/wn_err.txt
  python3 -c "
import json
d=json.loads(open('/tmp/wn_out.json').read().strip().splitlines()[-1]); k=d['kernels']
print('$name', '[$flags]', 'ms/step', d['ms_per_step'], {n: round(v['ms_per_step'],3) for n,v in k.items() if 'slab' in n})" | tee -a $O
done
