# usage: bash tools/ab_lib.sh tools/_scratch/lib_old.so   (same-box A/B of a prebuilt library against the in-tree one; run through gpurun)
set -e
cd ${GRAFT_REPO_ROOT:-.}
L=bubbleformer_amd/libbubbleformer_hip.so
cp $L /tmp/lib_new.so; cp $1 /tmp/lib_old.so
for rep in 1 2 3; do
  for v in old new; do
    cp /tmp/lib_$v.so $L
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_avg_us']; print('$v', round(d['value'],1), round(d['ms_per_step'],3), {n: k[n] for n in k if 'attn' in n})"
  done
done
cp /tmp/lib_new.so $L
