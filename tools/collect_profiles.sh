cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${ROUND:-r02}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > $O/pmc_sq_bench.json 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_write.err
find $O -name "*.csv" | head -20; du -sh $O
# the split-bf16 variant of the same command (kernel trace only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bf16x3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --precision bf16x3 > $O/trace_bf16x3_bench.json 2> $O/trace_bf16x3.err
