#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 derived metrics (one --pmc pass per metric; tools/pmc_utilisation.sh).

  python tools/pmc_table.py <dir with pmc_<Metric>/ sub-directories> <round tag> Metric [Metric ...]

Each pass's counter_collection.csv has one row per dispatch and counter; values are averaged over the dispatches of a kernel (names
normalised as in tools/rocprof_summary.py).  With --pmc the profiler runs every dispatch alone, so the figures describe the kernel by itself."""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import normalise      # noqa: E402


def read(d, metric):
    vals, durs = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != metric:
                continue
            k = normalise(row["Kernel_Name"])
            vals[k].append(float(row["Counter_Value"]))
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                durs[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return vals, durs


def main():
    root, tag, metrics = sys.argv[1], sys.argv[2], sys.argv[3:]
    table, dur, calls = {}, {}, {}
    for m in metrics:
        vals, durs = read(os.path.join(root, "pmc_" + m), m)
        for k, v in vals.items():
            table.setdefault(k, {})[m] = sum(v) / len(v)
            calls[k] = len(v)
            if durs.get(k):
                dur[k] = sum(durs[k]) / len(durs[k])
    keys = sorted(table, key=lambda k: -(dur.get(k, 0.0) * calls.get(k, 0)))
    print("# Per-kernel utilisation counters, training step (%s; rocprofv3 --pmc, one derived metric per pass, every dispatch alone on the chip)\n" % tag)
    print("| kernel | dispatches seen | avg us (serialised) | " + " | ".join(metrics) + " |")
    print("|---|---|---|" + "---|" * len(metrics))
    for k in keys[:40]:
        print("| `%s` | %d | %s | " % (k, calls[k], ("%.1f" % dur[k]) if k in dur else "") + " | ".join(("%.1f" % table[k][m]) if m in table[k] else "" for m in metrics) + " |")


if __name__ == "__main__":
    main()
