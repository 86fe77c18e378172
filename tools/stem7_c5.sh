#!/bin/bash
# C5 with the persistent fp16 stem and with the tile kernel, alternating on one box, + the kernel's duration under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S="--workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads"
python tools/stem_ab.py stem7f16 2>&1 | tail -1
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then export HP_STEM7_F16_OLD=1; else unset HP_STEM7_F16_OLD; fi
  python bench.py $S | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$v\", d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"])"
done; done
unset HP_STEM7_F16_OLD
rocprofv3 --kernel-trace --stats -d gpurun_out/ks_s7 -o p --output-format csv -- python3 bench.py $S --lanes 1 > gpurun_out/ks_s7.log 2>&1
grep -E "stem7" $(find gpurun_out/ks_s7 -name p_kernel_stats.csv) | cut -c1-160; rm -rf gpurun_out/ks_s7
