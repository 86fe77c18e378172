#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06g; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for g in off on; do
    timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --graphs $g > $O/bench_graphs_${g}_$rep.json 2> $O/bench_graphs_${g}_$rep.err
    python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_graphs_${g}_$rep.json").read().strip().splitlines()[-1])
    e=d["estimator"]
    print("graphs $g rep $rep", round(d["value"],1), "poses/s frac", round(d["roofline"]["frac"],4), "predictor", round(e["predictor_value"],1), "overhead", round(e["overhead_vs_predictor"],4), "eager ms", d.get("eager_ms_per_step"))
except Exception as e: print("bench graphs=$g $rep failed", e); print(open("$O/bench_graphs_${g}_$rep.err").read()[-1500:])
P
  done
done
