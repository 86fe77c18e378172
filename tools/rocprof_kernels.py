"""Per-kernel durations from a rocprofv3 run (``--kernel-trace``; the default rocpd database or ``--output-format csv``):
count, average / min / max microseconds per kernel name, largest total first.  Usage: python tools/rocprof_kernels.py <dir> [top]"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    return [(r[0], r[1]) for r in cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id")]


def rows_from_csv(path):
    with open(path) as f:
        return [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(f)]


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*results.db"), recursive=True):
        rows += rows_from_db(p)
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += rows_from_csv(p)
    agg = defaultdict(list)
    for name, ns in rows:
        agg[name].append(ns / 1e3)
    print("%-100s %6s %10s %10s %10s" % ("kernel", "n", "avg_us", "min_us", "max_us"))
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print("%-100s %6d %10.1f %10.1f %10.1f" % (name[:100], len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
