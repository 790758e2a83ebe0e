set -e
cd $GRAFT_REPO_ROOT
make -C bubbleformer_amd/csrc clean > /dev/null
make -C bubbleformer_amd/csrc EXTRA=-DBF_EXPERIMENTS -j12 > gpurun_out/ab_build.log 2>&1
for i in 1 2; do
  for cfg in "A" "B BF_GATHER_GEMM=0 BF_SCATTER_GEMM=0" "C BF_GATHER_GEMM=0 BF_SCATTER_GEMM=0 BF_DEBED_LAST_NORM=0 BF_EMBED_TAIL=0"; do
    set -- $cfg; name=$1; shift
    env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/ab_${name}_$i.json 2> gpurun_out/ab_${name}_$i.err
    python3 -c "
import json,sys; d=json.loads(open('gpurun_out/ab_${name}_$i.json').read().strip().splitlines()[-1]); print('$name', $i, round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
