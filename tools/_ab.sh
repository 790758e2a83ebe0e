mkdir -p gpurun_out/final
timeout -k 10 900 python bench.py > gpurun_out/final/r02_bench_n1.json 2> gpurun_out/final/n1.log || { tail -5 gpurun_out/final/n1.log; exit 1; }
timeout -k 10 900 python bench.py --config configs3 > gpurun_out/final/r02_bench_configs3.json 2> gpurun_out/final/c3.log || { tail -5 gpurun_out/final/c3.log; exit 1; }
timeout -k 10 900 python bench.py --config configs4 > gpurun_out/final/r02_bench_configs4.json 2> gpurun_out/final/c4.log || { tail -5 gpurun_out/final/c4.log; exit 1; }
cp profiles/r02_rollout_record.json gpurun_out/final/ 2>/dev/null
python3 - <<'PY'
import json
for n in ("n1", "configs3", "configs4"):
    d = json.load(open(f"gpurun_out/final/r02_bench_{n}.json"))
    print(n, round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d["roofline"].get("traffic"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
