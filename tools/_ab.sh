for v in 100 0 200 400; do BF_GEMM_FEW_TILES=$v timeout -k 10 300 python bench.py --config configs4 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e.json 2> gpurun_out/r2_e.log && python3 -c "
import json; d=json.load(open('gpurun_out/r2_e.json')); print('few=$v configs4', round(d['value'],1), round(d['ms_per_step'],4))"; done
