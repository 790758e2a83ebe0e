mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "stream or tokred or gemm" > gpurun_out/r2_t3.log 2>&1; rc=$?; tail -15 gpurun_out/r2_t3.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/gemm_bench.py 20 > gpurun_out/r2_gemm_new.log 2>&1; cat gpurun_out/r2_gemm_new.log
BF_GEMM_STREAM=0 timeout -k 10 200 python tools/gemm_bench.py 20 > gpurun_out/r2_gemm_old.log 2>&1; cat gpurun_out/r2_gemm_old.log
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2_d_$name.json 2> gpurun_out/r2_d_$name.log || { tail -5 gpurun_out/r2_d_$name.log; return 1; }; python3 - <<PY
import json
d=json.load(open("gpurun_out/r2_d_$name.json")); r=d["roofline"]
print("$name", round(d["value"],1), "loss", d["loss"], "kernel ms/step", round(r["gpu_kernel_ms_per_step"],2))
print("   ", r["kernel_avg_us"])
PY
}
run new A=1 && run nostream BF_GEMM_STREAM=0
