#!/bin/bash
# MFMA operand order inside the C segment of conv3x3_pp (split mode): does sharing an operand fragment between consecutive MFMAs move
# the power-limited clock?  ord0 = as shipped, ord1 = pixel fragment shared for runs of 4 / 2, ord2 = weight fragment shared.  Stamps build.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06i; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in 0 1 2; do
  export HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib_ord$v/libhappypose_amd.so
  B=128 timeout 300 python tools/conv_bench.py > $O/ord${v}_b128_$rep.txt 2>&1
  echo "== order $v rep $rep (batch 128)"; grep -E "30x| 15x|  8x" $O/ord${v}_b128_$rep.txt | grep "pre=0"
done
done
for rep in 1 2; do
for v in 0 1 2; do
  export HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib_ord$v/libhappypose_amd.so
  timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 --entry predictor > $O/bench_ord${v}_$rep.json 2> $O/bench_ord${v}_$rep.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_ord${v}_$rep.json").read().strip().splitlines()[-1])
    print("C2 order $v rep $rep", round(d["value"],1), "poses/s frac", round(d["roofline"]["frac"],4))
except Exception as e: print("bench failed", e)
P
done
done
