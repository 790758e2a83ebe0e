#!/bin/bash
# Per-kernel utilisation counters of the training step (rocprofv3 derived metrics, ONE metric per pass; with --pmc every dispatch runs alone on
# the chip, so these are the kernels' own figures, not the two-queue schedule's) -> gpurun_out/prof/rNN_pmc_utilisation.md
# usage (through gpurun): bash tools/pmc_utilisation.sh r04
R=${1:-r04}
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in MfmaUtil VALUBusy LdsUtil LdsBankConflict MemUnitStalled MeanOccupancyPerCU; do
  rocprofv3 --pmc $M --kernel-trace --output-format csv -d $OUT/pmc_$M -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc_$M.log || { tail -5 $OUT/pmc_$M.log; exit 1; }
  echo "pass $M done"
done
cd $ROOT
python3 tools/pmc_table.py $OUT $R MfmaUtil VALUBusy LdsUtil LdsBankConflict MemUnitStalled MeanOccupancyPerCU > $OUT/${R}_pmc_utilisation.md
for M in MfmaUtil VALUBusy LdsUtil LdsBankConflict MemUnitStalled MeanOccupancyPerCU; do rm -rf $OUT/pmc_$M; done
head -30 $OUT/${R}_pmc_utilisation.md
