"""Timeline summary of a rocprofv3 ``--kernel-trace`` CSV of ``bench.py`` (run on the GPU box, the trace stays there).

python3 tools/trace_timeline.py <p_kernel_trace.csv> <out.json> [n_steps]

Takes the LAST n_steps refiner steps of the C2 run (delimited by ``pose_prep_kernel`` launches: 2 lanes x 5 iterations per
step), and reports for that window: wall time, per-queue busy time and idle gaps, the time during which 0 / 1 / 2+ kernels
ran, per-kernel totals, and the window's kernel list (short name, queue, start us, duration us) so that the overlap of the
two lanes can be looked at offline.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"hp::", "", name)
    m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    path, out = sys.argv[1], sys.argv[2]
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    preps = [i for i, r in enumerate(rows) if "pose_prep_kernel" in r[2]]
    per_step = 10  # 2 lanes x 5 iterations
    assert len(preps) >= per_step * (n_steps + 1), len(preps)
    first = preps[-per_step * (n_steps + 1)]  # window: from the first prep of the n-th step from the end ...
    last = preps[-per_step]                   # ... to the first prep of the last step (whole steps, the final one is cut off)
    win = rows[first:last]
    t0, t1 = win[0][0], max(r[1] for r in win)
    t1 = rows[last][0]
    wall = (t1 - t0) * 1e-3
    by_q = defaultdict(list)
    for s, e, n, q, st in win:
        by_q[(q, st)].append((s, e, n))
    queues = {}
    for q, lst in by_q.items():
        busy = sum(e - s for s, e, _ in lst) * 1e-3
        gaps = [(lst[i + 1][0] - lst[i][1]) * 1e-3 for i in range(len(lst) - 1)]
        pos = [g for g in gaps if g > 0]
        queues[f"q{q[0]}/s{q[1]}"] = {"launches": len(lst), "busy_us": busy, "gap_sum_us": sum(pos), "gap_mean_us": sum(pos) / max(1, len(pos)),
                                      "gaps_over_10us": sum(1 for g in pos if g > 10), "gap_over_10us_sum": sum(g for g in pos if g > 10)}
    ev = []
    for s, e, *_ in win:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, prev, hist = 0, t0, defaultdict(float)
    for t, d in ev:
        hist[min(depth, 3)] += (min(t, t1) - prev) * 1e-3 if t > prev else 0.0
        prev = max(prev, min(t, t1))
        depth += d
    per_kernel = defaultdict(lambda: [0, 0.0])
    for s, e, n, *_ in win:
        k = per_kernel[short(n)]
        k[0] += 1; k[1] += (e - s) * 1e-3
    summary = {"steps": n_steps, "wall_us": wall, "us_per_step": wall / n_steps, "queues": queues,
               "concurrency_us": {str(k): v for k, v in sorted(hist.items())},
               "kernels": {k: {"n": v[0], "total_us": v[1], "avg_us": v[1] / v[0]} for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
               "sum_kernel_us": sum(v[1] for v in per_kernel.values())}
    # the kernel list of the LAST whole step in the window
    step_first = preps[-2 * per_step]
    lst = rows[step_first:last]
    qid = {q: i for i, q in enumerate(sorted({(r[3], r[4]) for r in lst}))}
    summary["last_step_kernels"] = [[short(n), qid[(q, st)], round((s - lst[0][0]) * 1e-3, 2), round((e - s) * 1e-3, 2)] for s, e, n, q, st in lst]
    with open(out, "w") as fh:
        json.dump(summary, fh)
    print(json.dumps({k: summary[k] for k in ("us_per_step", "queues", "concurrency_us", "sum_kernel_us")}, indent=1))


if __name__ == "__main__":
    main()
