#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun).  $1 = tag, $2 = what: "stage" (rasteriser / crop counters),
# "bench" (kernel stats + conv traffic of the bench command), "all".
set -x
TAG=${1:-r03}
WHAT=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$WHAT" = "stage" ] || [ "$WHAT" = "all" ]; then
  timeout 300 python3 tools/stage_workload.py > $OUT/stage_workload.json 2> $OUT/stage_workload.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/st_stats -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/st_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/st_pmc1 -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/st_pmc1.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $OUT/st_pmc2 -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/st_pmc2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/st_fetch -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/st_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/st_write -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/st_write.log 2>&1
  python3 tools/pmc_summary.py $(find $OUT/st_pmc1 -name p_counter_collection.csv | head -1) > $OUT/raster_pmc.txt
  python3 tools/pmc_summary.py $(find $OUT/st_pmc2 -name p_counter_collection.csv | head -1) >> $OUT/raster_pmc.txt
  python3 tools/pmc_traffic_stage.py $(dirname $(find $OUT/st_fetch -name p_counter_collection.csv | head -1)) $(dirname $(find $OUT/st_write -name p_counter_collection.csv | head -1)) $OUT/stage_workload.json $OUT/raster_hbm_traffic.json
  cp $(find $OUT/st_stats -name p_kernel_stats.csv | head -1) $OUT/stage_kernel_stats.csv
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $OUT/prof.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/trf_fetch -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $OUT/trf_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/trf_write -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $OUT/trf_write.log 2>&1
  python3 tools/pmc_traffic.py $(dirname $(find $OUT/trf_fetch -name p_counter_collection.csv | head -1)) $(dirname $(find $OUT/trf_write -name p_counter_collection.csv | head -1)) $OUT/conv_hbm_traffic.json 1800
  cp $(find $OUT/prof -name p_kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
fi
find $OUT -name "p_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
