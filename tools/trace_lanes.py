"""Lane concurrency of the C2 step from a rocprofv3 kernel trace (round 6).

    rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads
    python tools/trace_lanes.py out/.../p_kernel_trace.csv[.gz]  > profiles/rNN_step_trace_analysis.txt

Per timed step (a step = the launches between two consecutive groups of ten head_kernel launches: 2 lanes x 5 iterations): its
duration, the time during which 0 / 1 / 2 of the lanes' streams had a kernel in flight, and per lane the summed kernel time by
family and the summed GAPS between the end of a kernel and the start of the next one on the same stream."""
import collections
import csv
import gzip
import sys


def main(path):
    op = gzip.open if path.endswith(".gz") else open
    rows = list(csv.DictReader(op(path, "rt")))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    lanes = [sid for sid, _ in collections.Counter(r["Stream_Id"] for r in rows if "conv3x3_pp" in r["Kernel_Name"]).most_common(2)]
    heads = [r for r in rows if "head_kernel" in r["Kernel_Name"] and r["Stream_Id"] in lanes]
    n_steps = len(heads) // 10
    print(f"{len(rows)} kernel rows, lanes = streams {lanes}, {n_steps} steps of 2 lanes x 5 iterations (the tracer inflates every launch-to-launch gap)")
    fam = lambda n: ("conv3x3_pp" if "conv3x3_pp" in n else "conv3x3s2_pp" if "conv3x3s2" in n else "stem" if "stem" in n else
                     "raster" if "raster" in n else "crop" if "crop" in n else "other")
    for g in range(2, n_steps):
        prev_end = heads[10 * g - 1]["e"]
        last = heads[10 * g + 9]["e"] + 200_000
        ks = [r for r in rows if prev_end < r["s"] <= last and r["Stream_Id"] in lanes]
        start, end = min(r["s"] for r in ks), max(r["e"] for r in ks)
        ev = sorted([(r["s"], 1, r["Stream_Id"]) for r in ks] + [(r["e"], -1, r["Stream_Id"]) for r in ks])
        act, tot, t_last = collections.Counter(), collections.Counter(), start
        for t, d, sid in ev:
            tot[sum(1 for v in act.values() if v > 0)] += t - t_last
            t_last = t
            act[sid] += d
        line = f"step {g}: {(end - start) / 1e6:6.2f} ms; lanes with a kernel in flight: none {tot[0] / 1e6:5.2f} ms, one {tot[1] / 1e6:5.2f} ms, both {tot[2] / 1e6:5.2f} ms"
        for sid in lanes:
            lane = [r for r in ks if r["Stream_Id"] == sid]
            c = collections.Counter()
            for r in lane:
                c[fam(r["Kernel_Name"])] += r["e"] - r["s"]
            gaps = sum(max(0, lane[i + 1]["s"] - lane[i]["e"]) for i in range(len(lane) - 1))
            line += (f"\n    lane {sid}: {len(lane)} kernels, kernel time {sum(c.values()) / 1e6:5.2f} ms ("
                     + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in c.most_common()) + f"), gaps between its kernels {gaps / 1e6:5.2f} ms")
        print(line)


if __name__ == "__main__":
    main(sys.argv[1])
