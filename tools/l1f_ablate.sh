# Diagnostic: ablated / variant builds of the layer-1 F(2, 2) kernel (csrc/cdae_l1f.h, XSQ_L1F_ABL; ablated results are wrong by
# construction, only the timings matter).  Run on the GPU box:  [SHOW="kernel name filters"] tools/l1f_ablate.sh OUTFILE "NAME=EXTRA FLAGS" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; shift
: > $O
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C $R/xumx_slicq_amd/csrc -j16 OBJDIR=/tmp/lf_$name OUT=/tmp/liblf_$name.so "EXTRA=$flags" 2>/dev/null >/dev/null
  XSQ_LIB=/tmp/liblf_$name.so python3 $R/tools/kernel_times.py ${SHOW:-cdae_l1 cdae_l4} 2>/tmp/lf_err.txt | sed "s/^/$name [$flags] /" | tee -a $O || tail -3 /tmp/lf_err.txt
done
