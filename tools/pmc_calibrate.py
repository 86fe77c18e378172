"""Counter / true-byte ratios of the calibration kernels (tools/pmc_calibrate.sh).  Prints one row per kernel and counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
true = {}
for ln in open(os.path.join(out, "cal_bytes.txt")):
    p = ln.split()
    if len(p) == 3 and p[0] == "bytes":
        true[p[1]] = int(p[2])
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "cal_*", "**", "p_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in true:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per dispatch, median of 3; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them")
print(f"{'kernel':18s} {'true MB':>9s}  counters")
for k in true:
    row = []
    for c, v in sorted(vals[k].items()):
        m = sorted(v)[len(v) // 2]
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            row.append(f"{c}={m * 1024 / 1e6:.1f}MB ({m * 1024 / true[k]:.3f}x)")
        else:
            row.append(f"{c}={m:.4g} ({true[k] / max(m, 1):.1f} B/req)")
    print(f"{k:18s} {true[k] / 1e6:9.1f}  " + "  ".join(row))
