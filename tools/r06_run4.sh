#!/bin/bash
# Round 6, GPU call 4: the one-wave-per-SIMD 3x3 kernel (conv3x3_pp1, HP_CONV_PP1=1) against the ping-pong kernel: parity, per-layer
# times, the C2 step, C5.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06d; mkdir -p $O
export PYTHONUNBUFFERED=1
HP_CONV_PP1=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or backbone" > $O/tests_pp1.log 2>&1
echo "pp1 conv tests rc=$?"; tail -4 $O/tests_pp1.log
for b in 64 128; do
  for v in 0 1; do
    HP_CONV_PP1=$v B=$b timeout 300 python tools/conv_bench.py > $O/convbench_b${b}_pp1_$v.txt 2>&1
    echo "== B=$b PP1=$v"; grep -E "x" $O/convbench_b${b}_pp1_$v.txt | grep -E "30x| 15x|  8x" 
  done
done
for rep in 1 2; do
  for v in 0 1; do
    HP_CONV_PP1=$v timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --entry predictor > $O/bench_pp1_${v}_$rep.json 2> $O/bench_pp1_${v}_$rep.err
    python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_pp1_${v}_$rep.json").read().strip().splitlines()[-1])
    print("C2 pp1=$v rep $rep", round(d["value"],1), "poses/s frac", round(d["roofline"]["frac"],4), "scratch", d["scratch_launches"], "parity?", d.get("parity"))
except Exception as e: print("bench pp1=$v $rep failed", e); print(open("$O/bench_pp1_${v}_$rep.err").read()[-1500:])
P
  done
done
for v in 0 1; do
  HP_CONV_PP1=$v timeout 400 python bench.py --workload C5 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c5_pp1_$v.json 2> $O/bench_c5_pp1_$v.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_c5_pp1_$v.json").read().strip().splitlines()[-1])
    print("C5 pp1=$v", round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],4))
except Exception as e: print("C5 pp1=$v failed", e); print(open("$O/bench_c5_pp1_$v.err").read()[-1500:])
P
done
