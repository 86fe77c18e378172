"""Mean of each PMC counter per kernel from a rocprofv3 --pmc run (counter_collection.csv)."""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void hp::", "")
    name = re.sub(r"\(.*", "", name)
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if pat and pat not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k, f"(n={len(next(iter(acc[k].values())))})")
    print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        print("   of wave cycles: " + "  ".join(f"{n[3:]}={c[n] / w:.3f}" for n in sorted(c) if n != "SQ_WAVE_CYCLES" and n.startswith("SQ_") and "BUSY" not in n))
