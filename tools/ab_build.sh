# usage: bash tools/ab_build.sh "<EXTRA flags A>" "<EXTRA flags B>"   (same-box A/B of two builds; run through gpurun)
set -e
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for v in "$1" "$2"; do
    make -C bubbleformer_amd/csrc clean > /dev/null
    make -C bubbleformer_amd/csrc EXTRA="$v" -j12 > gpurun_out/ab_build.log 2>&1
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel_avg_us'].get('gemm_pair<inbwd>'))"
  done
done
