mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "frame_linear" > gpurun_out/r2_k.log 2>&1; rc=$?; tail -5 gpurun_out/r2_k.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "config4 or bench_size" > gpurun_out/r2_c4.log 2>&1; rc=$?; tail -5 gpurun_out/r2_c4.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config configs4 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e.json 2> gpurun_out/r2_e.log && python3 -c "
import json; d=json.load(open('gpurun_out/r2_e.json')); print('configs4', round(d['value'],1), d['ms_per_step'])"
export TMPDIR=/tmp; ROOT=$PWD; cd /tmp; rm -rf /tmp/tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $ROOT/tools/frame_fwd_bench.py > /tmp/o.log 2>&1 || { tail -3 /tmp/o.log; exit 1; }
python3 - "$(find /tmp/tr -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "frame_" in n: print("   %-60s calls %6s avg %8.2f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
