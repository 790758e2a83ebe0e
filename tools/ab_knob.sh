# usage: bash tools/ab_knob.sh KNOB v1 v2 ...   (same-box A/B on an experiments build; run through gpurun)
set -e
cd ${GRAFT_REPO_ROOT:-.}
K=$1; shift
make -C bubbleformer_amd/csrc clean > /dev/null
make -C bubbleformer_amd/csrc EXTRA=-DBF_EXPERIMENTS -j12 > gpurun_out/ab_build.log 2>&1
for rep in 1 2 3; do
  for v in "$@"; do
    env $K=$v timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$K=$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
