"""Per-kernel summary of a rocprofv3 ``--kernel-trace --pmc`` pass (counter_collection.csv): calls, mean duration, and --
when the counters are there -- the matrix-pipe busy fraction MFMA_BUSY / (SIMDs x GRBM_GUI_ACTIVE), the effective clock
GRBM_GUI_ACTIVE / duration, and the SQ wave-cycle shares.  Kernels run one at a time under --pmc (no lane overlap)."""
import collections, csv, re, sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|void |hp::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
SIMDS = 1024
tot = sum(sum(d.values()) for d in dur.values())
print(f"{'kernel':70s} {'calls':>6s} {'avg us':>8s} {'share':>6s} {'MHz':>6s} {'mfma_busy':>9s} {'valu':>6s} {'wait_any':>8s} {'wait_inst':>9s} {'lds':>6s}")
for name in sorted(dur, key=lambda n: -sum(dur[n].values())):
    c = {k: sum(v) / len(v) for k, v in rows[name].items()}
    d = sum(dur[name].values()) / len(dur[name])
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    f = lambda k: f"{c[k] / w:.3f}" if w and k in c else "-"
    busy = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * gui):.3f}" if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c else "-"
    mhz = f"{gui / d:.0f}" if gui else "-"
    print(f"{name:70s} {len(dur[name]):6d} {d:8.1f} {sum(dur[name].values()) / tot:6.3f} {mhz:>6s} {busy:>9s} {f('SQ_ACTIVE_INST_VALU'):>6s} {f('SQ_WAIT_ANY'):>8s} {f('SQ_WAIT_INST_ANY'):>9s} {f('SQ_ACTIVE_INST_LDS'):>6s}")
