#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06e; mkdir -p $O
export PYTHONUNBUFFERED=1
for b in 64; do
  for v in 0 1; do
    HP_CONV_PP1=$v B=$b timeout 300 python tools/conv_bench.py > $O/convbench_b${b}_pp1_$v.txt 2>&1
    echo "== B=$b PP1=$v"; grep -E "x" $O/convbench_b${b}_pp1_$v.txt | grep -E "30x| 15x|  8x" 
  done
done
for v in 0 1; do
  HP_CONV_PP1=$v timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --entry predictor > $O/bench_pp1_${v}.json 2> $O/bench_pp1_${v}.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_pp1_${v}.json").read().strip().splitlines()[-1])
    print("C2 pp1=$v", round(d["value"],1), "poses/s frac", round(d["roofline"]["frac"],4), "scratch", d["scratch_launches"])
except Exception as e: print("bench pp1=$v failed", e); print(open("$O/bench_pp1_${v}.err").read()[-1500:])
P
done
