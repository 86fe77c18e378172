#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun).  $1 = tag (r04a ...), $2 = what: bench | pmc | traffic | raster | all
set -x
TAG=${1:-r06}
WHAT=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHORT="--steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads"
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $OUT/prof.log 2>&1
  cp $(find $OUT/prof -name p_kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  # SQ counters of every kernel of the C2 step and of the C5 step (kernels run one at a time under --pmc)
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_c2 -o p --output-format csv -- python3 bench.py $SHORT > $OUT/pmc_c2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_c5 -o p --output-format csv -- python3 bench.py --workload C5 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-extra-workloads > $OUT/pmc_c5.log 2>&1
  { echo "# C2 (bench.py $SHORT): SQ counters per kernel; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE), where GRBM_GUI_ACTIVE";
    echo "# is summed over the 8 XCDs on this stack (the MHz column is 8x the shader clock): multiply mfma_busy by 8 for the busy fraction of the matrix pipe";
    python3 tools/pmc_kernels.py $(find $OUT/pmc_c2 -name p_counter_collection.csv | head -1) | head -16;
    echo; echo "# C5 (bench.py --workload C5): the fp16 plan";
    python3 tools/pmc_kernels.py $(find $OUT/pmc_c5 -name p_counter_collection.csv | head -1) | head -12; } > $OUT/conv_pmc_summary.txt
fi
if [ "$WHAT" = "traffic" ] || [ "$WHAT" = "all" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/trf_fetch -o p --output-format csv -- python3 bench.py $SHORT > $OUT/trf_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/trf_write -o p --output-format csv -- python3 bench.py $SHORT > $OUT/trf_write.log 2>&1
  python3 tools/pmc_traffic.py $(dirname $(find $OUT/trf_fetch -name p_counter_collection.csv | head -1)) $(dirname $(find $OUT/trf_write -name p_counter_collection.csv | head -1)) $OUT/conv_hbm_traffic.json
fi
if [ "$WHAT" = "raster" ] || [ "$WHAT" = "all" ]; then
  export HP_STAGE_MSAA=1 HP_STAGE_ANISO=1 HP_STAGE_ONLY=raster
  SPECS=""
  for WL in C2 C3; do
    export HP_STAGE_WORKLOADS=$WL
    timeout 300 python3 tools/stage_workload.py > $OUT/stage_$WL.json 2> $OUT/stage_$WL.err
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/rs_fetch_$WL -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/rs_fetch_$WL.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/rs_write_$WL -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/rs_write_$WL.log 2>&1
    SPECS="$SPECS $WL:$(dirname $(find $OUT/rs_fetch_$WL -name p_counter_collection.csv | head -1)):$(dirname $(find $OUT/rs_write_$WL -name p_counter_collection.csv | head -1)):$OUT/stage_$WL.json"
  done
  python3 tools/pmc_traffic_raster.py $OUT/raster_hbm_traffic.json $SPECS
  unset HP_STAGE_ONLY HP_STAGE_WORKLOADS
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/rs_pmc -o p --output-format csv -- python3 tools/stage_workload.py > $OUT/rs_pmc.log 2>&1
  python3 tools/pmc_kernels.py $(find $OUT/rs_pmc -name p_counter_collection.csv | head -1) | head -10 > $OUT/raster_pmc.txt
fi
if [ "$WHAT" = "layers" ] || [ "$WHAT" = "all" ]; then
  # per-layer tables at the lanes' batch sizes (HP_PROFILE_LAYERS: each launch alone) -> the recomputable roofline file
  python3 tools/backbone_layers.py resnet34 6 f32 64 > $OUT/layers_C2.txt 2>&1
  python3 tools/backbone_layers.py vanilla_resnet34 32 f32 22 > $OUT/layers_C3.txt 2>&1
  python3 tools/backbone_layers.py vanilla_resnet34 9 f16 576 > $OUT/layers_C5.txt 2>&1
  python3 tools/backbone_layers.py efficientnet-b3 6 f32 64 > $OUT/layers_C2_efficientnet.txt 2>&1
  python3 tools/kernel_roofline.py $OUT/kernel_roofline.json --layers C2:$OUT/layers_C2.txt:64:4:3 C3:$OUT/layers_C3.txt:22:4:3 C5:$OUT/layers_C5.txt:576:2:1 \
      C2_efficientnet_b3:$OUT/layers_C2_efficientnet.txt:64:4:3 --kernel-stats $OUT/bench_kernel_stats.csv --conv-traffic $OUT/conv_hbm_traffic.json \
      --raster-traffic $OUT/raster_hbm_traffic.json --bench-line $OUT/bench_line.json
fi
find $OUT -name "p_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
