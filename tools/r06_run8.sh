#!/bin/bash
# graph replay of the refiner step against eager launches, through the entry point: five alternating pairs on one box
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 10 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 > /dev/null 2>&1   # warm the box
for rep in 1 2 3 4 5; do
  for g in off on; do
    timeout 400 python bench.py --steps 30 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --graphs $g > $O/bench_graphs_${g}_$rep.json 2> $O/bench_graphs_${g}_$rep.err
    python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_graphs_${g}_$rep.json").read().strip().splitlines()[-1])
    e=d["estimator"]
    print("graphs $g rep $rep", round(d["value"],1), "poses/s; predictor", round(e["predictor_value"],1))
except Exception as e: print("bench graphs=$g $rep failed", e); print(open("$O/bench_graphs_${g}_$rep.err").read()[-800:])
P
  done
done
