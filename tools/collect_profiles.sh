R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}      # (resolved before the cd: the scripts run from /tmp)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_${ROUND:-r02}
mkdir -p $O
# kernel trace with the step counts the driver's bench run uses (--steps 20 --warmup 5): a 3 + 1 step run times its launches from
# cold and reads 10-15 % long (r06z: slab 1.154 ms against 1.008 with these flags and 0.984 by the events of an untraced run)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > $O/pmc_sq_bench.json 2> $O/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > /dev/null 2> $O/pmc_write.err
find $O -name "*.csv" | head -20; du -sh $O
# the split-bf16 variant of the same command (kernel trace only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bf16x3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --precision bf16x3 > $O/trace_bf16x3_bench.json 2> $O/trace_bf16x3.err
# BASELINE configs[2] (Wiener-EM on): the same four passes into a second directory, summarised as <round>_wiener_*
W=$R/gpurun_out/prof_${ROUND:-r02}_wiener
mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --wiener > $W/trace_bench.json 2> $W/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $W/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --wiener > $W/pmc_sq_bench.json 2> $W/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --wiener > /dev/null 2> $W/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --wiener > /dev/null 2> $W/pmc_write.err
# summaries are small: produce them on the box, so that only they travel back (the raw traces exceed the 64 MiB merge limit)
mkdir -p $R/gpurun_out/prof_${ROUND:-r02}_summary
python3 $R/tools/summarize_profiles.py $O $R/gpurun_out/prof_${ROUND:-r02}_summary ${ROUND:-r02} > $R/gpurun_out/prof_${ROUND:-r02}_summary/summary.txt 2>&1
python3 $R/tools/summarize_profiles.py $W $R/gpurun_out/prof_${ROUND:-r02}_summary ${ROUND:-r02}_wiener > $R/gpurun_out/prof_${ROUND:-r02}_summary/summary_wiener.txt 2>&1
rm -rf $O/trace $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/trace_bf16x3 $W/trace $W/pmc_sq $W/pmc_fetch $W/pmc_write
