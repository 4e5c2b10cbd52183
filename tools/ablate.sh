# Diagnostic: build ablated variants of the library and time the bench with each (results of the
# ablated builds are wrong by construction; only their timings matter).  Run on the GPU box.
# FLAG = the ablation macro (XSQ_ABLATE: gemm_tile.h / slice_fft.h, XSQ_SLAB_ABL: cdae_slab.h), VARIANTS = its values, SHOW = kernel-name filters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/xumx_slicq_amd/csrc
for v in ${VARIANTS:-0 1 2 4 6 8 14 15}; do
  # same flags as the product library (csrc/Makefile), plus the ablation switch; objects outside the tree
  make -s -j4 OBJDIR=/tmp/ab_$v OUT=/tmp/libab_$v.so EXTRA=-D${FLAG:-XSQ_ABLATE}=$v 2>/dev/null
  echo "== ablate=$v"
  XSQ_LIB=/tmp/libab_$v.so python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/ab_err.txt > /tmp/ab_out.json || tail -5 /tmp/ab_err.txt
  python3 -c "
import json
d=json.loads(open('/tmp/ab_out.json').read().strip().splitlines()[-1]); k=d['kernels']
print(' ms/step', d['ms_per_step'], {n: round(v['ms_per_step'],2) for n,v in k.items() if any(w in n for w in '${SHOW:-gemm fft dft4}'.split())})" || tail -5 /tmp/ab_err.txt
done
[ -n "$NO_PEAK" ] || /opt/rocm/bin/hipcc -O3 -w --offload-arch=gfx950 $R/tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && /tmp/mfma_peak
