"""Instruction counts per kernel (SQ_INSTS_*: wave-instructions) from a rocprofv3 --pmc run: mean per dispatch.
    python3 tools/pmc_insts.py <p_counter_collection.csv> [substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void hp::(anonymous namespace)::", "").replace("hp::(anonymous namespace)::", "").split("(")[0]
    if sub and sub not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
print(f"{'kernel':52s} {'disp':>5s} " + " ".join(f"{c.replace('SQ_INSTS_', ''):>11s}" for c in names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    d = max(len(n[k]), 1)
    print(f"{k[:52]:52s} {d:5d} " + " ".join(f"{v.get(c, 0) / d:11.0f}" for c in names))
