#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): bench lines of every configuration, rocprofv3 kernel stats of
# the C2 bench command, SQ counters of the conv kernels on the layer shapes, HBM traffic (separate PMC passes).
# Outputs land in gpurun_out/r02/; the summaries are copied into profiles/ by hand.
set -x
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
timeout 300 python bench.py --workload C3 --no-cpu-baseline > $OUT/bench_line_C3.json 2>/dev/null
timeout 300 python bench.py --workload C5 --no-cpu-baseline > $OUT/bench_line_C5.json 2>/dev/null
timeout 300 python bench.py --workload E2E --steps 5 --warmup 2 > $OUT/bench_line_E2E.json 2>/dev/null
timeout 300 python bench.py --arch efficientnet-b3 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 > $OUT/bench_line_efficientnet.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 > $OUT/prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_C3 -o p --output-format csv -- python3 bench.py --workload C3 --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 > $OUT/prof_C3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_C5 -o p --output-format csv -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_C5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_eff -o p --output-format csv -- python3 bench.py --arch efficientnet-b3 --steps 4 --warmup 2 --no-cpu-baseline --no-exact-fp32 > $OUT/prof_eff.log 2>&1
timeout 300 python bench.py --workload C3 --graphs on --no-cpu-baseline --no-exact-fp32 > $OUT/bench_line_C3_graphs.json 2>/dev/null
timeout 300 python bench.py --graphs on --no-cpu-baseline --no-exact-fp32 > $OUT/bench_line_graphs.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/trf_fetch -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > $OUT/trf_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/trf_write -o p --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 > $OUT/trf_write.log 2>&1
python3 tools/pmc_traffic.py $(dirname $(find $OUT/trf_fetch -name p_counter_collection.csv | head -1)) $(dirname $(find $OUT/trf_write -name p_counter_collection.csv | head -1)) $OUT/conv_hbm_traffic.json 1800
export HP_CONV_SPLIT=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/spmc1 -o p --output-format csv -- python3 tools/conv_bench.py > $OUT/spmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/spmc2 -o p --output-format csv -- python3 tools/conv_bench.py > $OUT/spmc2.log 2>&1
unset HP_CONV_SPLIT
python3 tools/pmc_summary.py $(find $OUT/spmc1 -name p_counter_collection.csv | head -1) conv3x3 > $OUT/conv_pmc_summary.txt
python3 tools/pmc_summary.py $(find $OUT/spmc2 -name p_counter_collection.csv | head -1) conv3x3 >> $OUT/conv_pmc_summary.txt
for d in prof prof_C3 prof_C5 prof_eff; do cp $(find $OUT/$d -name p_kernel_stats.csv | head -1) $OUT/${d}_kernel_stats.csv; done
# keep the merge small: drop the raw traces
find $OUT -name "p_kernel_trace.csv" -delete; find $OUT -name "p_counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT; ls $OUT
