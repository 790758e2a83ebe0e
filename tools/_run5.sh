mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "stream" > gpurun_out/r2_t4.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t4.log; [ $rc -eq 0 ] || exit 1
for v in 1 0 1 0; do echo "stagger=$v"; BF_STREAM_STAGGER=$v timeout -k 10 200 python tools/stream_scale.py 2>&1 | grep -v amdgpu | grep "M= 18432\|M= 73728"; done
for v in 1 0; do BF_STREAM_STAGGER=$v timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e.json 2> gpurun_out/r2_e.log; python3 -c "
import json; d=json.load(open('gpurun_out/r2_e.json')); print('stagger=$v', round(d['value'],1), {k:v for k,v in d['roofline']['kernel_avg_us'].items() if 'stream' in k})"; done
