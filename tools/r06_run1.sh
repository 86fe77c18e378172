#!/bin/bash
# (build the packed variant first: HP_BUILD_VARIANT=pk HP_BUILD_DROP_FLAGS="-packed-fp32-ops" python -m happypose_amd.build)
# Round 6, GPU call 1: the C5 / fp16 parity probe, the GPU suite on the new build (no packed fp32, hidden visibility, debug table),
# A/B of the packed-fp32 build (lib_pk: the round-5 flags) against it, A/B of tail K-slicing inside forward_chunks.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/probes/c5_parity_probe.py > $O/c5_probe.json 2> $O/c5_probe.err
echo "probe rc=$?"; tail -c 3000 $O/c5_probe.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/gpu_tests.log
for rep in 1 2; do
  for v in new pk; do
    if [ $v = pk ]; then export HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib_pk/libhappypose_amd.so; else unset HAPPYPOSE_AMD_LIB; fi
    timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<P
import json
try:
    d=json.loads(open("$O/bench_${v}_$rep.json").read().strip().splitlines()[-1])
    print("$v $rep", round(d["value"],1), "poses/s  frac", round(d["roofline"]["frac"],4), "raster us", d.get("stages",{}).get("rasterize_reference_state",{}).get("us"))
except Exception as e: print("$v $rep failed", e)
P
  done
done
unset HAPPYPOSE_AMD_LIB
for ts in True False; do
  timeout 600 python - > $O/e2e_chunks_tailsplit_$ts.json 2> $O/e2e_chunks_tailsplit_$ts.err <<P
import sys, runpy
import happypose_amd.pose_predictor as PP
PP.TwoLanePredictor.CHUNKS_TAIL_SPLIT = $ts
sys.argv = ["bench.py", "--workload", "E2E", "--precision", "f16", "--steps", "6", "--warmup", "3"]
runpy.run_path("bench.py", run_name="__main__")
P
  python - <<P
import json
try:
    d=json.loads(open("$O/e2e_chunks_tailsplit_$ts.json").read().strip().splitlines()[-1])
    print("chunks tail split $ts:", round(d["value"],3), "frames/s", d["stage_ms_per_frame"])
except Exception as e: print("e2e $ts failed", e)
P
done
