#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE (and the raw TCC_EA0 request counters) on known byte counts: tools/probes/pmc_calibrate.hip.
# Run on the GPU box: gpurun -- bash tools/pmc_calibrate.sh <tag>
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate $GRAFT_REPO_ROOT/tools/probes/pmc_calibrate.hip || exit 1
/tmp/pmc_calibrate > $OUT/cal_bytes.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_32B" "TCC_EA0_WRREQ_WRITE_DRAM_32B" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/cal_$N -o p --output-format csv -- /tmp/pmc_calibrate > $OUT/cal_$N.log 2>&1
done
rocprofv3 --list-avail 2>/dev/null | grep -i -E "TCC_EA0|FETCH_SIZE|WRITE_SIZE|TCC_BUBBLE|TCC_REQ|TCC_HIT|TCC_MISS" | head -80 > $OUT/cal_avail.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_calibrate.py $OUT > $OUT/pmc_calibration.txt
cat $OUT/pmc_calibration.txt
find $OUT -name "p_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
