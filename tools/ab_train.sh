# Diagnostic: A/B of build-flag variants on the training step.  tools/ab_train.sh OUTFILE "NAME=EXTRA FLAGS" ...  (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; shift
: > $O
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C $R/xumx_slicq_amd/csrc -j16 OBJDIR=/tmp/abt_$name OUT=/tmp/libabt_$name.so "EXTRA=$flags" 2>/dev/null >/dev/null
  for p in fp32 bf16; do
    XSQ_LIB=/tmp/libabt_$name.so python3 $R/tools/bench_train.py --precision $p --steps 20 --warmup 5 --no-profile 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$name', '[$flags]', '$p', 'ms/step', round(d['ms_per_step'], 3), 'loss', d['losses'][-1])" | tee -a $O
  done
done
