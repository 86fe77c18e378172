#!/bin/bash
# (build the SLP variant first, in the build container: HP_BUILD_VARIANT=slp HP_BUILD_DROP_FLAGS="-packed-fp32-ops -fno-slp-vectorize" python -m happypose_amd.build)
# Round 6, GPU call 3: (a) the packed-fp32 finding, verdict item 5(b): the SLP build (lib_slp: the vectoriser on, packed fp32 allowed --
# the build that differed in 1 - 13 % of its two-lane steps in round 5) alone and with HP_RASTER_CANARY=1 (set-up records and
# transformed vertices filled with NaN before every rasteriser call, scratch addresses logged): do differing steps read something
# the call did not write?  (b) the bench line through the entry point after the estimator's bookkeeping moved under the kernels.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r06c; mkdir -p $O
export PYTHONUNBUFFERED=1
for mode in plain canary; do
  for i in 1 2 3 4 5 6; do
    if [ $mode = canary ]; then export HP_RASTER_CANARY=1; else unset HP_RASTER_CANARY; fi
    HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib_slp/libhappypose_amd.so timeout 300 python3 tools/probes/two_lane_repro.py 150 2 0 > $O/slp_${mode}_$i.json 2> $O/slp_${mode}_$i.err
    python3 - <<P
import json
try:
    d=json.loads(open("$O/slp_${mode}_$i.json").read().strip().splitlines()[-1]); t=d["first_difference_tally"]; f=d["first"] or {}
    print("slp $mode $i: differing runs", sum(t.values()), "of", d["runs"], sorted(t), "nan_in_cur", f.get("nan_in_cur"), "nan in reference snapshot", d.get("nan_in_reference_snapshot"), "channels", f.get("channels"))
except Exception as e: print("slp $mode $i failed", e)
P
    grep "hp raster scratch" $O/slp_${mode}_$i.err | tail -2
  done
done
unset HP_RASTER_CANARY
# the shipped library under the same canary: every step must stay bit-equal and NaN-free
HP_RASTER_CANARY=1 timeout 300 python3 tools/probes/two_lane_repro.py 150 2 0 > $O/shipped_canary.json 2> $O/shipped_canary.err
tail -c 600 $O/shipped_canary.json; echo
for rep in 1 2; do
  timeout 400 python bench.py --steps 20 --warmup 3 --no-extra-workloads --no-cpu-baseline --no-exact-fp32 > $O/bench_entry_$rep.json 2> $O/bench_entry_$rep.err
  python3 - <<P
import json
try:
    d=json.loads(open("$O/bench_entry_$rep.json").read().strip().splitlines()[-1])
    print("entry $rep", round(d["value"],1), "poses/s", d["entry"][:40], {k: (round(v,4) if isinstance(v,float) else v) for k,v in d["estimator"].items() if k != "entry"})
except Exception as e: print("bench $rep failed", e)
P
done
du -sh $O
