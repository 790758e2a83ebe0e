mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stream or gemm" > gpurun_out/r2_k.log 2>&1; rc=$?; tail -3 gpurun_out/r2_k.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/stream_scale.py > gpurun_out/r2_ss.log 2>&1 && cat gpurun_out/r2_ss.log &&
for v in 1 2; do timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e.json 2> gpurun_out/r2_e.log && python3 -c "
import json; d=json.load(open('gpurun_out/r2_e.json')); print('run$v', round(d['value'],1))"; done
