mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r2_gpu_all.log 2>&1; rc=$?; tail -3 gpurun_out/r2_gpu_all.log; [ $rc -eq 0 ] || exit $rc
for v in "A=1" "BF_SIDE_FRAME_SCALE=1" "A=2"; do env $v timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e.json 2> gpurun_out/r2_e.log; python3 -c "
import json; d=json.load(open('gpurun_out/r2_e.json')); print('$v', round(d['value'],1))"; done
