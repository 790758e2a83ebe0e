#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the bench command, the FETCH_SIZE / WRITE_SIZE passes (separate, as
# MI355X_MICROARCH.md section HBM prescribes) and the two-queue timeline -> gpurun_out/prof/rNN_* (copy the summaries to profiles/).
# usage: bash tools/profile_round.sh r03
R=${1:-r03}
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/${R}_bench_under_rocprof.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc_fetch.log || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc_write.log || { tail -5 $OUT/pmc_write.log; exit 1; }
cd $ROOT
python3 tools/rocprof_summary.py $OUT/trace $OUT/pmc_fetch $OUT/pmc_write --out $OUT/${R} > /dev/null
python3 tools/timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) --out $OUT/${R}_timeline.md --anatomy $OUT/${R}_step_anatomy.txt > /dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${R}_rocprofv3_kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT; head -12 $OUT/${R}_kernel_stats.md
