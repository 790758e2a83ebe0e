#!/bin/bash
# Rollout profile (bench.py --config configs4, HIP-graph replay): rocprofv3 kernel trace + stats and the FETCH_SIZE / WRITE_SIZE passes
# -> gpurun_out/prof/rNN_rollout_* (copy the summaries to profiles/).  usage: bash tools/profile_rollout.sh r02
R=${1:-r02}
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rtrace -- python3 $ROOT/bench.py --config configs4 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/${R}_rollout_bench_under_rocprof.json 2> $OUT/rtrace.log || { tail -5 $OUT/rtrace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rpmc_fetch -- python3 $ROOT/bench.py --config configs4 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/rpmc_fetch.log || { tail -5 $OUT/rpmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/rpmc_write -- python3 $ROOT/bench.py --config configs4 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/rpmc_write.log || { tail -5 $OUT/rpmc_write.log; exit 1; }
cd $ROOT
python3 tools/rocprof_summary.py $OUT/rtrace $OUT/rpmc_fetch $OUT/rpmc_write --steps 50 --out $OUT/${R}_rollout > /dev/null
rm -rf $OUT/rtrace $OUT/rpmc_fetch $OUT/rpmc_write
head -16 $OUT/${R}_rollout_kernel_stats.md
