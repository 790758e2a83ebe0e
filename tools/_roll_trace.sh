ROOT=$PWD; OUT=$ROOT/gpurun_out/roll; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config configs4 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
cd $ROOT
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/roll/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# take the middle third
n = len(rows); seg = rows[n//3: 2*n//3]
busy = sum(e-s for s,e,_ in seg); wall = seg[-1][1]-seg[0][0]
gaps = [seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
import statistics
print("kernels", len(seg), "wall us", wall/1e3, "busy us", busy/1e3, "median gap ns", statistics.median(gaps), "mean gap", sum(gaps)/len(gaps))
c = collections.defaultdict(lambda:[0,0])
for s,e,k in seg: c[k][0]+=1; c[k][1]+=e-s
for k,(cnt,t) in sorted(c.items(), key=lambda kv:-kv[1][1])[:25]: print("%-62s %5d %8.2f us avg %8.1f us total" % (k, cnt, t/cnt/1e3, t/1e3))
PY
rm -rf $OUT/trace
